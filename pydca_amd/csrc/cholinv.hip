// Inverse of a symmetric positive definite float64 matrix on MI355X, used for
// couplings = -inv(C)  (reference: compute_couplings, pydca/meanfield_dca/msa_numerics.py:321-342,
// which goes through LAPACK getrf/getri; C is SPD for pseudocount > 0, so the Cholesky route
// inv(A) = L^-T L^-1 is used: ~n^3 flop instead of 2 n^3).
//
// Everything that is a GEMM runs on the f64 matrix cores (v_mfma_f64_16x16x4_f64):
//   recursive block step on A = [[A11, .], [A21, A22]]  (sizes multiples of 64)
//     (X11)        <- cholinv(A11)                 X = L^-1, stored lower + mirrored upper
//     L21          <- A21 * X11^T                  (TRSM as a GEMM with the inverse)
//     A22          <- A22 - L21 * L21^T            (SYRK, lower tiles only)
//     (X22)        <- cholinv(A22)
//     T^T          <- X11^T * L21^T
//     X21          <- -X22 * T                     (written to (2,1) and mirrored to (1,2))
//   finally inv(A) = X^T X as one triangular-aware GEMM.
// All products are of the form C[i][j] = sum_k A[i][k] * B[j][k] ("NT", both operands
// K-contiguous), which is why X is kept mirrored: X^T rows are then plain rows.
// The 64x64 diagonal leaves (Cholesky + triangular inverse) run in one workgroup in LDS.
#include <climits>
#include <mutex>
#include <atomic>

#include <string>

#include "dca_internal.h"

namespace {

typedef double double4_t __attribute__((ext_vector_type(4)));

enum { MASK_NONE = 0, MASK_LOWER = 1 /* k <= row */, MASK_UPPER = 2 /* k >= row */ };

enum { WALK_ROWS = 0, WALK_ROWS_REVERSED = 1 /* lower triangular A */, WALK_COLUMNS_REVERSED = 2 /* lower triangular B; grid transposed */ };

constexpr int BM = 64, BN = 64;

// The kernels of the factorisation's serial chain (leaves, few-tile products) raise their waves' issue priority: when the
// look-ahead bulk products of another stream share a CU with them, the chain's instructions go first on the shared SIMDs
// and matrix pipes (s_setprio arbitrates between the waves of a CU; it is not a queue priority -- those made the chain slower, round 3).
#ifndef DCA_CHAIN_PRIO_LEVEL
#define DCA_CHAIN_PRIO_LEVEL 3
#endif
#define DCA_CHAIN_PRIO() __builtin_amdgcn_s_setprio(DCA_CHAIN_PRIO_LEVEL)

struct GemmArgs {
    const double* A; int lda; int maskA;
    const double* B; int ldb; int maskB;
    double* C; int ldc;
    double* Cm; int ldcm;          // optional mirror: Cm[j][i] = C[i][j]
    int M, N, K;
    double alpha, beta;
    int lowerOnly;                 // square output: only tiles with tj <= ti; strict mirror rule on the diagonal
    int walk;                      // order in which the tiles are started, so that with a triangular operand the tiles with the
                                   // longest k range are not the ones that start last (WALK_*)
    int row0 = 0;                  // first tile row of this launch (WALK_ROWS only): a product issued as several row bands
    // split-k form (gemm_nt_f64_dma_kernel only): blockIdx.z = slice of kChunk columns of the k range; slice z writes its partial
    // product (alpha, beta = 0) to C + z * sliceStride -- summed in slice order by gemm_slices_reduce_kernel
    int kSlices = 1, kChunk = 0;
    size_t sliceStride = 0;
    const double* Cin = nullptr; int ldcin = 0;   // beta != 0: the addend is read from here instead of from C (C is then written only)
};

// BK = 16: the throughput form (35 KB of LDS, three workgroups per CU).  BK = 64: for the many products of
// the recursion that are a handful of tiles on an otherwise empty GPU -- their time is a chain of global
// round trips, one per k-tile, so a four times deeper tile means four times fewer of them (K = 64: one).
template <int BK>
constexpr size_t gemm_lds_bytes() { return (size_t)2 * (BM + BN) * (BK + 2) * sizeof(double); }

template <int BK, bool AHEAD2 = false>
__global__ __launch_bounds__(256)
void gemm_nt_f64_kernel(GemmArgs g)
{
    constexpr int LDS_STRIDE = BK + 2;   // doubles; even: rows stay 16-byte aligned, so a lane's two neighbouring k values are ONE
                                         // ds_read_b128 / ds_write_b128 (operands of two MFMAs; the k order inside a tile is free as long as A and B agree)
    constexpr int PG = BK / 16;          // 16-byte pieces per thread and row: the row's BK/2 pieces over 8 threads
    extern __shared__ __attribute__((aligned(16))) unsigned char dca_gemm_smem[];
    double* const As = reinterpret_cast<double*>(dca_gemm_smem);      // [2][BM * LDS_STRIDE]
    double* const Bs = As + 2 * BM * LDS_STRIDE;                      // [2][BN * LDS_STRIDE]
    const int ti = g.walk == WALK_COLUMNS_REVERSED ? (int)blockIdx.x : g.walk == WALK_ROWS_REVERSED ? (int)(gridDim.y - 1 - blockIdx.y) : (int)blockIdx.y + g.row0;
    const int tj = g.walk == WALK_COLUMNS_REVERSED ? (int)(gridDim.y - 1 - blockIdx.y) : (int)blockIdx.x;
    if (g.lowerOnly && tj > ti) return;
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;

    // K range implied by the triangular operands
    int kLo = 0, kHi = g.K;
    if (g.maskA == MASK_LOWER) kHi = min(kHi, (ti + 1) * BM);
    if (g.maskA == MASK_UPPER) kLo = max(kLo, ti * BM);
    if (g.maskB == MASK_LOWER) kHi = min(kHi, (tj + 1) * BN);
    if (g.maskB == MASK_UPPER) kLo = max(kLo, tj * BN);
    kLo = kLo / BK * BK;

    double4_t acc[2][2];
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int n = 0; n < 2; ++n) acc[m][n] = (double4_t){0.0, 0.0, 0.0, 0.0};

    // staging role: 16-byte pieces, thread -> rows sr and sr + 32, doubles sp + 16 * pg, +1 of the k-tile, so that
    // one wave-wide load covers 8 full 128-byte lines.  global -> registers (raw) -> LDS; the triangular masks are
    // applied when a tile is WRITTEN to LDS, after the MFMAs of the current tile, so the loads of the next tile
    // are in flight during those MFMAs instead of being waited for right away.
    typedef double double2_t __attribute__((ext_vector_type(2)));
    const int sr = tid >> 3, sp = (tid & 7) * 2;
    const double* Ap = g.A + (size_t)(ti * BM + sr) * g.lda + sp;
    const double* Bp = g.B + (size_t)(tj * BN + sr) * g.ldb + sp;
    const size_t aStep = (size_t)32 * g.lda, bStep = (size_t)32 * g.ldb;
    double2_t ra[2][PG], rb[2][PG];
    double2_t ra2[2][PG], rb2[2][PG];      // second register set (AHEAD2: two k-tiles of global loads in flight)
    auto load_tile_into = [&](double2_t (&xa)[2][PG], double2_t (&xb)[2][PG], int k0) {
#pragma unroll
        for (int ps = 0; ps < 2; ++ps)
#pragma unroll
            for (int pg = 0; pg < PG; ++pg) {
                const int kp = k0 + 16 * pg;
                xa[ps][pg] = *reinterpret_cast<const double2_t*>(Ap + ps * aStep + kp);
                xb[ps][pg] = *reinterpret_cast<const double2_t*>(Bp + ps * bStep + kp);
            }
    };
    auto load_tile = [&](int k0) {
#pragma unroll
        for (int ps = 0; ps < 2; ++ps)
#pragma unroll
            for (int pg = 0; pg < PG; ++pg) {
                const int kp = k0 + 16 * pg;
                ra[ps][pg] = *reinterpret_cast<const double2_t*>(Ap + ps * aStep + kp);
                rb[ps][pg] = *reinterpret_cast<const double2_t*>(Bp + ps * bStep + kp);
            }
    };
    auto store_tile_from = [&](double2_t (&ra)[2][PG], double2_t (&rb)[2][PG], int buf, int k0) {
        const bool diagA = g.maskA != MASK_NONE && k0 + BK > ti * BM && k0 < (ti + 1) * BM;
        const bool diagB = g.maskB != MASK_NONE && k0 + BK > tj * BN && k0 < (tj + 1) * BN;
        if (diagA || diagB) {
#pragma unroll
            for (int ps = 0; ps < 2; ++ps) {
                const int aRow = ti * BM + sr + 32 * ps, bRow = tj * BN + sr + 32 * ps;
                const int aLo = g.maskA == MASK_UPPER ? aRow : INT_MIN, aHi = g.maskA == MASK_LOWER ? aRow : INT_MAX;
                const int bLo = g.maskB == MASK_UPPER ? bRow : INT_MIN, bHi = g.maskB == MASK_LOWER ? bRow : INT_MAX;
#pragma unroll
                for (int pg = 0; pg < PG; ++pg)
#pragma unroll
                    for (int u = 0; u < 2; ++u) {
                        const int k = k0 + 16 * pg + sp + u;
                        ra[ps][pg][u] = (k < aLo || k > aHi) ? 0.0 : ra[ps][pg][u];
                        rb[ps][pg][u] = (k < bLo || k > bHi) ? 0.0 : rb[ps][pg][u];
                    }
            }
        }
#pragma unroll
        for (int ps = 0; ps < 2; ++ps)
#pragma unroll
            for (int pg = 0; pg < PG; ++pg) {
                double* ad = As + buf * BM * LDS_STRIDE + (sr + 32 * ps) * LDS_STRIDE + 16 * pg + sp;
                double* bd = Bs + buf * BN * LDS_STRIDE + (sr + 32 * ps) * LDS_STRIDE + 16 * pg + sp;
                *reinterpret_cast<double2_t*>(ad) = ra[ps][pg];
                *reinterpret_cast<double2_t*>(bd) = rb[ps][pg];
            }
    };
    auto store_tile = [&](int buf, int k0) {
        // only k-tiles that cross the diagonal of a triangular operand's row block need masking (wave-uniform test)
        const bool diagA = g.maskA != MASK_NONE && k0 + BK > ti * BM && k0 < (ti + 1) * BM;
        const bool diagB = g.maskB != MASK_NONE && k0 + BK > tj * BN && k0 < (tj + 1) * BN;
        if (diagA || diagB) {
#pragma unroll
            for (int ps = 0; ps < 2; ++ps) {
                // k range in which the operand row is non-zero: lower k <= row, upper k >= row
                const int aRow = ti * BM + sr + 32 * ps, bRow = tj * BN + sr + 32 * ps;
                const int aLo = g.maskA == MASK_UPPER ? aRow : INT_MIN, aHi = g.maskA == MASK_LOWER ? aRow : INT_MAX;
                const int bLo = g.maskB == MASK_UPPER ? bRow : INT_MIN, bHi = g.maskB == MASK_LOWER ? bRow : INT_MAX;
#pragma unroll
                for (int pg = 0; pg < PG; ++pg)
#pragma unroll
                    for (int u = 0; u < 2; ++u) {
                        const int k = k0 + 16 * pg + sp + u;
                        ra[ps][pg][u] = (k < aLo || k > aHi) ? 0.0 : ra[ps][pg][u];
                        rb[ps][pg][u] = (k < bLo || k > bHi) ? 0.0 : rb[ps][pg][u];
                    }
            }
        }
#pragma unroll
        for (int ps = 0; ps < 2; ++ps)
#pragma unroll
            for (int pg = 0; pg < PG; ++pg) {
                double* ad = As + buf * BM * LDS_STRIDE + (sr + 32 * ps) * LDS_STRIDE + 16 * pg + sp;
                double* bd = Bs + buf * BN * LDS_STRIDE + (sr + 32 * ps) * LDS_STRIDE + 16 * pg + sp;
                *reinterpret_cast<double2_t*>(ad) = ra[ps][pg];
                *reinterpret_cast<double2_t*>(bd) = rb[ps][pg];
            }
    };

    const int nk = (kHi - kLo + BK - 1) / BK;
    auto mma_tile = [&](int buf) {
        const double* as = As + buf * BM * LDS_STRIDE;
        const double* bs = Bs + buf * BN * LDS_STRIDE;
#pragma unroll
        for (int kk = 0; kk < BK / 8; ++kk) {
            // lane group g = lane >> 4 holds k = 8 kk + 2 g, 8 kk + 2 g + 1: the first MFMA sums the even, the second the odd ones
            const int kcol = kk * 8 + 2 * (lane >> 4);
            const double2_t a0 = *reinterpret_cast<const double2_t*>(&as[(wm * 32 + (lane & 15)) * LDS_STRIDE + kcol]);
            const double2_t a1 = *reinterpret_cast<const double2_t*>(&as[(wm * 32 + 16 + (lane & 15)) * LDS_STRIDE + kcol]);
            const double2_t b0 = *reinterpret_cast<const double2_t*>(&bs[(wn * 32 + (lane & 15)) * LDS_STRIDE + kcol]);
            const double2_t b1 = *reinterpret_cast<const double2_t*>(&bs[(wn * 32 + 16 + (lane & 15)) * LDS_STRIDE + kcol]);
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                acc[0][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0[h], b0[h], acc[0][0], 0, 0, 0);
                acc[0][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0[h], b1[h], acc[0][1], 0, 0, 0);
                acc[1][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1[h], b0[h], acc[1][0], 0, 0, 0);
                acc[1][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1[h], b1[h], acc[1][1], 0, 0, 0);
            }
        }
    };
    if constexpr (AHEAD2) {
        // two k-tiles of global loads in flight (register sets ra/rb and ra2/rb2 alternate): a tile's data has two
        // MFMA phases to arrive before it is written to LDS
        if (nk > 0) { load_tile_into(ra, rb, kLo); store_tile_from(ra, rb, 0, kLo); }
        if (nk > 1) load_tile_into(ra2, rb2, kLo + BK);
        __syncthreads();
        for (int t = 0; t < nk; t += 2) {
            if (t + 2 < nk) load_tile_into(ra, rb, kLo + (t + 2) * BK);
            mma_tile(0);
            if (t + 1 < nk) store_tile_from(ra2, rb2, 1, kLo + (t + 1) * BK);
            __syncthreads();
            if (t + 1 >= nk) break;
            if (t + 3 < nk) load_tile_into(ra2, rb2, kLo + (t + 3) * BK);
            mma_tile(1);
            if (t + 2 < nk) store_tile_from(ra, rb, 0, kLo + (t + 2) * BK);
            __syncthreads();
        }
    } else {
#ifndef DCA_GEMM_ABLATE
#define DCA_GEMM_ABLATE 0          // tools/experiments/gemm_bench.hip, timing only: 1 no global loads in the loop, 2 no LDS stores, 4 no barrier
#endif
        if (nk > 0) {
            load_tile(kLo);
            store_tile(0, kLo);
        }
        __syncthreads();
        for (int t = 0; t < nk; ++t) {
            const int buf = t & 1;
            if (t + 1 < nk && !(DCA_GEMM_ABLATE & 1)) load_tile(kLo + (t + 1) * BK);
            mma_tile(buf);
            if (t + 1 < nk && !(DCA_GEMM_ABLATE & 2)) store_tile(buf ^ 1, kLo + (t + 1) * BK);
            if (!(DCA_GEMM_ABLATE & 4)) __syncthreads();
        }
    }

    // epilogue.  f64 16x16x4 accumulator layout: col = lane & 15, row = (lane >> 4) + 4 * reg
    // (the mirror stores are 8 bytes at stride ldcm; transposing the block through LDS first made no
    // measurable difference to the inverse)
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int n = 0; n < 2; ++n)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int i = ti * BM + wm * 32 + m * 16 + (lane >> 4) + 4 * r;
                const int j = tj * BN + wn * 32 + n * 16 + (lane & 15);
                if (g.lowerOnly && j > i) continue;
                double v = g.alpha * acc[m][n][r];
                double* cp = g.C + (size_t)i * g.ldc + j;
                if (g.beta != 0.0) v += g.beta * (g.Cin ? g.Cin[(size_t)i * g.ldcin + j] : *cp);
                *cp = v;
                if (g.Cm && !(g.lowerOnly && i == j)) g.Cm[(size_t)j * g.ldcm + i] = v;
            }
}

// The deep-k product on 32 x 32 output tiles, for launches that are far fewer than one 64 x 64 tile per CU: a tile's
// time is its k loop on the matrix pipe (64 cycles per 16 x 16 x 4 MFMA, one wave per SIMD) and 256 CUs are there, so a
// quarter of the work per workgroup on four times as many CUs is what shortens the chain of small products in the
// recursion (plain product of n = 640: 34 -> 21 us, n = 256: 16 -> 8, the 2 x 2-tile products of the 256-nodes 10.6 -> 5.2 us;
// inverse at n = 10 048: 27.1 -> 24.2 ms, at n = 4000: 4.96 -> 3.95 ms).  Same staging as gemm_nt_f64_kernel<64> (one pass of 32 rows per operand), each wave
// one 16 x 16 accumulator.
__device__ __forceinline__ void gemm_nt_f64_small_tile(const GemmArgs& g, int bx, int by, int gy)
{
    constexpr int TM = 32, BK = 64;
    constexpr int LDS_STRIDE = BK + 2;
    constexpr int PG = BK / 16;
    typedef double double2_t __attribute__((ext_vector_type(2)));
    DCA_CHAIN_PRIO();
    extern __shared__ __attribute__((aligned(16))) unsigned char dca_gemm_smem[];
    double* const As = reinterpret_cast<double*>(dca_gemm_smem);      // [2][TM * LDS_STRIDE]
    double* const Bs = As + 2 * TM * LDS_STRIDE;
    const int ti = g.walk == WALK_COLUMNS_REVERSED ? bx : g.walk == WALK_ROWS_REVERSED ? gy - 1 - by : by + g.row0;
    const int tj = g.walk == WALK_COLUMNS_REVERSED ? gy - 1 - by : bx;
    if (g.lowerOnly && tj > ti) return;
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;

    int kLo = 0, kHi = g.K;
    if (g.maskA == MASK_LOWER) kHi = min(kHi, (ti + 1) * TM);
    if (g.maskA == MASK_UPPER) kLo = max(kLo, ti * TM);
    if (g.maskB == MASK_LOWER) kHi = min(kHi, (tj + 1) * TM);
    if (g.maskB == MASK_UPPER) kLo = max(kLo, tj * TM);
    kLo = kLo / BK * BK;
    const int nk = (kHi - kLo + BK - 1) / BK;

    double4_t acc = (double4_t){0.0, 0.0, 0.0, 0.0};
    const int sr = tid >> 3, sp = (tid & 7) * 2;                      // row 0 .. 31, doubles sp + 16 pg, + 1 of the k-tile
    const double* Ap = g.A + (size_t)(ti * TM + sr) * g.lda + sp;
    const double* Bp = g.B + (size_t)(tj * TM + sr) * g.ldb + sp;
    double2_t ra[PG], rb[PG];
    auto load_tile = [&](int k0) {
#pragma unroll
        for (int pg = 0; pg < PG; ++pg) {
            ra[pg] = *reinterpret_cast<const double2_t*>(Ap + k0 + 16 * pg);
            rb[pg] = *reinterpret_cast<const double2_t*>(Bp + k0 + 16 * pg);
        }
    };
    auto store_tile = [&](int buf, int k0) {
        const bool diagA = g.maskA != MASK_NONE && k0 + BK > ti * TM && k0 < (ti + 1) * TM;
        const bool diagB = g.maskB != MASK_NONE && k0 + BK > tj * TM && k0 < (tj + 1) * TM;
        if (diagA || diagB) {
            const int aRow = ti * TM + sr, bRow = tj * TM + sr;
            const int aLo = g.maskA == MASK_UPPER ? aRow : INT_MIN, aHi = g.maskA == MASK_LOWER ? aRow : INT_MAX;
            const int bLo = g.maskB == MASK_UPPER ? bRow : INT_MIN, bHi = g.maskB == MASK_LOWER ? bRow : INT_MAX;
#pragma unroll
            for (int pg = 0; pg < PG; ++pg)
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    const int k = k0 + 16 * pg + sp + u;
                    ra[pg][u] = (k < aLo || k > aHi) ? 0.0 : ra[pg][u];
                    rb[pg][u] = (k < bLo || k > bHi) ? 0.0 : rb[pg][u];
                }
        }
#pragma unroll
        for (int pg = 0; pg < PG; ++pg) {
            *reinterpret_cast<double2_t*>(As + buf * TM * LDS_STRIDE + sr * LDS_STRIDE + 16 * pg + sp) = ra[pg];
            *reinterpret_cast<double2_t*>(Bs + buf * TM * LDS_STRIDE + sr * LDS_STRIDE + 16 * pg + sp) = rb[pg];
        }
    };
    auto mma_tile = [&](int buf) {
        const double* as = As + buf * TM * LDS_STRIDE;
        const double* bs = Bs + buf * TM * LDS_STRIDE;
#pragma unroll
        for (int kk = 0; kk < BK / 8; ++kk) {
            const int kcol = kk * 8 + 2 * (lane >> 4);
            const double2_t a = *reinterpret_cast<const double2_t*>(&as[(wm * 16 + (lane & 15)) * LDS_STRIDE + kcol]);
            const double2_t b = *reinterpret_cast<const double2_t*>(&bs[(wn * 16 + (lane & 15)) * LDS_STRIDE + kcol]);
            acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a[0], b[0], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a[1], b[1], acc, 0, 0, 0);
        }
    };
    if (nk > 0) { load_tile(kLo); store_tile(0, kLo); }
    __syncthreads();
    for (int t = 0; t < nk; ++t) {
        const int buf = t & 1;
        if (t + 1 < nk) load_tile(kLo + (t + 1) * BK);
        mma_tile(buf);
        if (t + 1 < nk) store_tile(buf ^ 1, kLo + (t + 1) * BK);
        __syncthreads();
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int i = ti * TM + wm * 16 + (lane >> 4) + 4 * r;
        const int j = tj * TM + wn * 16 + (lane & 15);
        if (g.lowerOnly && j > i) continue;
        double v = g.alpha * acc[r];
        double* cp = g.C + (size_t)i * g.ldc + j;
        if (g.beta != 0.0) v += g.beta * (g.Cin ? g.Cin[(size_t)i * g.ldcin + j] : *cp);
        *cp = v;
        if (g.Cm && !(g.lowerOnly && i == j)) g.Cm[(size_t)j * g.ldcm + i] = v;
    }
}

__global__ __launch_bounds__(256)
void gemm_nt_f64_small_kernel(GemmArgs g)
{
    gemm_nt_f64_small_tile(g, (int)blockIdx.x, (int)blockIdx.y, (int)gridDim.y);
}

// Two INDEPENDENT few-tile products in one launch (blockIdx.z picks; the grid is the larger of the two): the recursion's
// A22 -= L21 L21^T and T^T = X11^T L21^T both need L21 and nothing of each other, and at the lower levels a launch costs
// what its handful of tiles cost.
__global__ __launch_bounds__(256)
void gemm_nt_f64_small_pair_kernel(GemmArgs g0, int gx0, int gy0, GemmArgs g1, int gx1, int gy1)
{
    const bool second = blockIdx.z != 0;
    const int gx = second ? gx1 : gx0, gy = second ? gy1 : gy0;
    if ((int)blockIdx.x >= gx || (int)blockIdx.y >= gy) return;
    gemm_nt_f64_small_tile(second ? g1 : g0, (int)blockIdx.x, (int)blockIdx.y, gy);
}

// The same product with the operand tiles brought into LDS by LDS-DMA (global_load_lds_dwordx4: 16 bytes per lane straight
// from global memory into LDS, no staging registers and no ds_write).  In the kernel above a quarter of the matrix-core
// time is lost around the register -> LDS stores (tools/experiments/gemm_bench.hip -DDCA_GEMM_ABLATE=2: 59.6 -> 66.2 TF
// without them, 67.7 with MFMAs and fragment reads alone, which is the f64 MFMA rate at the sustained clock).
// A DMA instruction writes the 64 lanes' 16-byte pieces to CONTIGUOUS LDS (1 KiB), so a tile row is exactly its 16
// doubles (128 bytes, no padding) and bank conflicts are avoided by a swizzle instead: the piece that lies in 16-byte
// slot s of row R is k-piece s ^ ((R >> 1) & 7) -- chosen on the global side, where every lane may fetch what it likes.
// A fragment read (16 rows x 4 lane groups, ds_read_b128) then touches 16 different slots of the 256-byte bank window in
// each of its four lane groups.  Triangular operands: masked on the fragments of the k-tiles that cross the diagonal.
// MFMA tiles per wave along M and N: <2,2> -> 64 x 64 output tile per workgroup, <4,4> -> 128 x 128 (twice the flop per
// operand byte), <4,2> -> 128 x 64 (10.7 instead of 8 flop per operand byte at three workgroups per CU: for the products
// with two triangular operands, where the 128 x 128 form loses to its tails)
// KW = 2 (round 5): TWO groups of four waves in one workgroup work on the SAME output tile, group w on the k-tiles
// t = w (mod 2), each with its own double-buffered operand tiles; group 1 hands its accumulators over through LDS at the end.
// For the look-ahead bulk products, which run as ONE workgroup per CU so that whole CUs stay free for the chain: with four
// waves a SIMD's matrix pipe idles whenever its only wave waits (48 TF with 240 workgroups), with eight it has a second one.
template <int TWM, int TWN = TWM, int KW = 1>
__global__ __launch_bounds__(256 * KW, KW == 2 ? 1 : (TWM == 4 && TWN == 4) ? 2 : (TWM == 4 || TWN == 4) ? 3 : 4)
void gemm_nt_f64_dma_kernel(GemmArgs g)
{
    constexpr int BK = 16;
    constexpr int BM = 32 * TWM, BN = 32 * TWN;               // shadow the file-level 64 x 64
    constexpr int PWA = BM / 8 / 4, PWB = BN / 8 / 4;         // 1 KiB DMA pieces (8 tile rows) per wave and operand
    constexpr int OPA = BM * BK * (int)sizeof(double), OPB = BN * BK * (int)sizeof(double);       // bytes of the operand tiles
    typedef double double2_t __attribute__((ext_vector_type(2)));
    extern __shared__ __attribute__((aligned(16))) unsigned char dca_gemm_smem[];   // [2][A | B]
    const int ti = g.walk == WALK_COLUMNS_REVERSED ? (int)blockIdx.x : g.walk == WALK_ROWS_REVERSED ? (int)(gridDim.y - 1 - blockIdx.y) : (int)blockIdx.y + g.row0;
    const int tj = g.walk == WALK_COLUMNS_REVERSED ? (int)(gridDim.y - 1 - blockIdx.y) : (int)blockIdx.x;
    if (g.lowerOnly && tj * BN >= (ti + 1) * BM) return;       // the tile lies entirely above the diagonal
    const int tid = threadIdx.x;
    const int lane = tid & 63, waveAll = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wave = waveAll & 3, grp = waveAll >> 2;           // grp: the k-tile group of this wave (KW == 2)
    const int wm = wave >> 1, wn = wave & 1;
    unsigned char* const smem = dca_gemm_smem + grp * 2 * (OPA + OPB);

    int kLo = 0, kHi = g.K;
    if (g.maskA == MASK_LOWER) kHi = min(kHi, (ti + 1) * BM);
    if (g.maskA == MASK_UPPER) kLo = max(kLo, ti * BM);
    if (g.maskB == MASK_LOWER) kHi = min(kHi, (tj + 1) * BN);
    if (g.maskB == MASK_UPPER) kLo = max(kLo, tj * BN);
    double* Cout = g.C;
    if (g.kSlices > 1) {
        kLo = max(kLo, (int)blockIdx.z * g.kChunk);
        kHi = min(kHi, ((int)blockIdx.z + 1) * g.kChunk);
        Cout += (size_t)blockIdx.z * g.sliceStride;
    }
    kLo = kLo / BK * BK;
    const int nk = (kHi - kLo + BK - 1) / BK;

    double4_t acc[TWM][TWN];
#pragma unroll
    for (int m = 0; m < TWM; ++m)
#pragma unroll
        for (int n = 0; n < TWN; ++n) acc[m][n] = (double4_t){0.0, 0.0, 0.0, 0.0};

    // DMA role of the lane: pieces PW wave .. PW wave + PW - 1 of each operand (piece = 8 tile rows); row-in-piece lane >> 3,
    // slot lane & 7.  Rows past the end of the operand (M, N multiples of 64, not of 128) are fetched from its last row:
    // their results are not stored.
    const double* srcA[PWA];
    const double* srcB[PWB];
#pragma unroll
    for (int i = 0; i < PWA; ++i) {
        const int R = 8 * (PWA * wave + i) + (lane >> 3);
        const int piece = (lane & 7) ^ ((R >> 1) & 7);
        srcA[i] = g.A + (size_t)min(ti * BM + R, g.M - 1) * g.lda + 2 * piece;
    }
#pragma unroll
    for (int i = 0; i < PWB; ++i) {
        const int R = 8 * (PWB * wave + i) + (lane >> 3);
        const int piece = (lane & 7) ^ ((R >> 1) & 7);
        srcB[i] = g.B + (size_t)min(tj * BN + R, g.N - 1) * g.ldb + 2 * piece;
    }
    auto issue = [&](int buf, int k0) {
#pragma unroll
        for (int i = 0; i < PWA; ++i)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(srcA[i] + k0),
                                             (__attribute__((address_space(3))) void*)(smem + buf * (OPA + OPB) + (PWA * wave + i) * 1024), 16, 0, 0);
#pragma unroll
        for (int i = 0; i < PWB; ++i)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(srcB[i] + k0),
                                             (__attribute__((address_space(3))) void*)(smem + buf * (OPA + OPB) + OPA + (PWB * wave + i) * 1024), 16, 0, 0);
    };
    const int fr = lane & 15, fg = lane >> 4, swz = (fr >> 1) & 7;
    auto mma_tile = [&](int buf, int k0) {
        const unsigned char* as = smem + buf * (OPA + OPB);
        const unsigned char* bs = as + OPA;
        const bool diagA = g.maskA != MASK_NONE && k0 + BK > ti * BM && k0 < (ti + 1) * BM;    // wave-uniform
        const bool diagB = g.maskB != MASK_NONE && k0 + BK > tj * BN && k0 < (tj + 1) * BN;
#pragma unroll
        for (int kk = 0; kk < BK / 8; ++kk) {
            const int slot = (4 * kk + fg) ^ swz;             // lane group fg holds k = 8 kk + 2 fg, + 1
            double2_t a[TWM], b[TWN];
#pragma unroll
            for (int m = 0; m < TWM; ++m) a[m] = *reinterpret_cast<const double2_t*>(as + (wm * 16 * TWM + 16 * m + fr) * 128 + slot * 16);
#pragma unroll
            for (int m = 0; m < TWN; ++m) b[m] = *reinterpret_cast<const double2_t*>(bs + (wn * 16 * TWN + 16 * m + fr) * 128 + slot * 16);
            if (diagA || diagB) {
                // only the 16-row fragments whose rows the 8 k values of this slab actually cross need the per-element test
                // (wave-uniform per fragment): the masking is vector-ALU work in front of the MFMAs
                const int kmin = k0 + 8 * kk, kmax = kmin + 7;
#pragma unroll
                for (int m = 0; m < TWM; ++m) {
                    const int aBase = ti * BM + wm * 16 * TWM + 16 * m;
                    const bool needA = diagA && ((g.maskA == MASK_UPPER && kmin < aBase + 15) || (g.maskA == MASK_LOWER && kmax > aBase));
                    if (needA) {
                        const int aRow = aBase + fr;
                        const int aLo = g.maskA == MASK_UPPER ? aRow : INT_MIN, aHi = g.maskA == MASK_LOWER ? aRow : INT_MAX;
#pragma unroll
                        for (int h = 0; h < 2; ++h) {
                            const int k = kmin + 2 * fg + h;
                            a[m][h] = (k < aLo || k > aHi) ? 0.0 : a[m][h];
                        }
                    }
                }
#pragma unroll
                for (int m = 0; m < TWN; ++m) {
                    const int bBase = tj * BN + wn * 16 * TWN + 16 * m;
                    const bool needB = diagB && ((g.maskB == MASK_UPPER && kmin < bBase + 15) || (g.maskB == MASK_LOWER && kmax > bBase));
                    if (needB) {
                        const int bRow = bBase + fr;
                        const int bLo = g.maskB == MASK_UPPER ? bRow : INT_MIN, bHi = g.maskB == MASK_LOWER ? bRow : INT_MAX;
#pragma unroll
                        for (int h = 0; h < 2; ++h) {
                            const int k = kmin + 2 * fg + h;
                            b[m][h] = (k < bLo || k > bHi) ? 0.0 : b[m][h];
                        }
                    }
                }
            }
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int m = 0; m < TWM; ++m)
#pragma unroll
                    for (int n = 0; n < TWN; ++n) acc[m][n] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[m][h], b[n][h], acc[m][n], 0, 0, 0);
        }
    };

    if constexpr (KW == 1) {
        if (nk > 0) issue(0, kLo);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        for (int t = 0; t < nk; ++t) {
            const int buf = t & 1;
            if (t + 1 < nk) issue(buf ^ 1, kLo + (t + 1) * BK);
            mma_tile(buf, kLo + t * BK);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // this wave's pieces of the next tile have landed
            __syncthreads();                                       // ... everyone's; and the tile just used may be overwritten
        }
    } else {
        // group grp takes the k-tiles grp, grp + KW, ...; both groups make the same number of trips (the barriers are the workgroup's)
        const int trips = (nk + KW - 1) / KW;
        if (grp < nk) issue(0, kLo + grp * BK);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        for (int u = 0; u < trips; ++u) {
            const int buf = u & 1, t = u * KW + grp;
            if (t + KW < nk) issue(buf ^ 1, kLo + (t + KW) * BK);
            if (t < nk) mma_tile(buf, kLo + t * BK);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
        }
        // group 1 -> LDS -> group 0 (one double4 per lane and MFMA tile: 16 x 4 x 64 lanes x 32 B = 128 KiB, the operand buffers' space)
        double4_t* const xfer = reinterpret_cast<double4_t*>(dca_gemm_smem);
        if (grp == 1) {
#pragma unroll
            for (int m = 0; m < TWM; ++m)
#pragma unroll
                for (int n = 0; n < TWN; ++n) xfer[((m * TWN + n) * 4 + wave) * 64 + lane] = acc[m][n];
        }
        __syncthreads();
        if (grp == 1) return;
#pragma unroll
        for (int m = 0; m < TWM; ++m)
#pragma unroll
            for (int n = 0; n < TWN; ++n) acc[m][n] += xfer[((m * TWN + n) * 4 + wave) * 64 + lane];
    }

#pragma unroll
    for (int m = 0; m < TWM; ++m)
#pragma unroll
        for (int n = 0; n < TWN; ++n)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int i = ti * BM + wm * 16 * TWM + m * 16 + (lane >> 4) + 4 * r;
                const int j = tj * BN + wn * 16 * TWN + n * 16 + (lane & 15);
                if (i >= g.M || j >= g.N) continue;
                if (g.lowerOnly && j > i) continue;
                double v = g.alpha * acc[m][n][r];
                double* cp = Cout + (size_t)i * g.ldc + j;
                if (g.beta != 0.0) v += g.beta * (g.Cin ? g.Cin[(size_t)i * g.ldcin + j] : *cp);
                *cp = v;
                if (g.Cm && !(g.lowerOnly && i == j)) g.Cm[(size_t)j * g.ldcm + i] = v;
            }
}

// ------------------------------------------------------------------ stream-K form of the bulk products (round 5)
// The look-ahead factorisation's bulk products (C -= A B^T, no triangular operand, 128 x 128 tiles) have FEW output tiles and
// a DEEP k range -- the deep update of a 512-column panel is (rows / 128) x 4 tiles, e.g. 172 at the middle of n = 10 048 --
// and ran as one workgroup per tile: 172 of 256 CUs busy for the whole k walk (41 TF), or, split along k into equal
// slices, tile x slice counts that again do not divide by the CU count.  Here the ITERATION space (tiles x k-tiles of 16)
// is cut into equal contiguous pieces: every workgroup does the same number of k-tiles (+-1), a piece that covers a whole
// tile is finished in place, the at most two partial pieces of a workgroup (the tail of one tile at its start, the head of
// another at its end) go to two slots of a scratch array, and gemm_streamk_fixup_kernel adds the pieces of every split tile
// in ascending k order before it applies alpha / beta -- a fixed order, so the result does not depend on timing.
// Operand reuse: the gx column tiles of one tile row read the same rows of A.  The iteration space is therefore
// (tile rows x k-tiles), cut into W pieces, and piece q is walked by a GROUP of gx workgroups, one per column tile, with
// ids that put them on the same XCD (id % 8) -- they start together and run the same loop, so A's k-tiles are fetched
// once per group and hit that XCD's L2 for the others.  (The first version cut tiles x k-tiles per workgroup: neighbouring
// workgroups then stand at different k of the same rows, every operand tile comes from the fabric -- 1.4 GB per deep
// update instead of 0.2 -- and the deep updates got slower, 8.8 against 6.9 ms at n = 10 048.)
struct StreamKArgs {
    const double* A; int lda;
    const double* B; int ldb; int maskB;     // MASK_LOWER: B is lower triangular (k <= row); the k range is NOT shortened
    double* C; int ldc;
    int M, N, K;
    double alpha, beta;
    double* P;                               // 2 W gx slots of 128 x 128 doubles: slot ((2 q + s) gx + tj)
    int gx, KT, W;                           // tile columns, k-tiles per tile, groups (pieces)
    long long I;                             // iterations = tile rows x KT
};
__host__ __device__ __forceinline__ long long streamk_first(const StreamKArgs& g, int q) { return g.I * q / g.W; }
__host__ __device__ __forceinline__ int streamk_owner(const StreamKArgs& g, long long it)      // the q with first(q) <= it < first(q + 1)
{
    return (int)(((it + 1) * g.W + g.I - 1) / g.I) - 1;
}

__global__ __launch_bounds__(256, 1)
void gemm_nt_f64_streamk_kernel(StreamKArgs g)
{
    constexpr int BK = 16, TW = 4;
    constexpr int BM = 128, BN = 128;
    constexpr int PW = BM / 8 / 4;                                  // 1 KiB DMA pieces (8 tile rows) per wave and operand
    constexpr int OP = BM * BK * (int)sizeof(double);
    typedef double double2_t __attribute__((ext_vector_type(2)));
    extern __shared__ __attribute__((aligned(16))) unsigned char dca_gemm_smem[];   // [3][A | B]
    unsigned char* const smem = dca_gemm_smem;
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int fr = lane & 15, fg = lane >> 4, swz = (fr >> 1) & 7;
    // id -> (group q, column tile tj): the gx workgroups of a group have the same id % 8, i.e. the same XCD
    const int per = 8 * g.gx;
    const int w = ((int)blockIdx.x / per) * 8 + ((int)blockIdx.x % 8);
    const int tj = ((int)blockIdx.x % per) / 8;
    if (w >= g.W) return;
    const long long itBeg = streamk_first(g, w), itEnd = streamk_first(g, w + 1);

    for (long long it = itBeg; it < itEnd;) {
        const int ti = (int)(it / g.KT), kt0 = (int)(it % g.KT);
        const int kt1 = (int)min((long long)g.KT, kt0 + (itEnd - it));
        const int nk = kt1 - kt0, kLo = kt0 * BK;
        double4_t acc[TW][TW];
#pragma unroll
        for (int m = 0; m < TW; ++m)
#pragma unroll
            for (int n = 0; n < TW; ++n) acc[m][n] = (double4_t){0.0, 0.0, 0.0, 0.0};
        const double* srcA[PW];
        const double* srcB[PW];
#pragma unroll
        for (int i = 0; i < PW; ++i) {
            const int R = 8 * (PW * wave + i) + (lane >> 3);
            const int piece = (lane & 7) ^ ((R >> 1) & 7);
            srcA[i] = g.A + (size_t)min(ti * BM + R, g.M - 1) * g.lda + 2 * piece;
            srcB[i] = g.B + (size_t)min(tj * BN + R, g.N - 1) * g.ldb + 2 * piece;
        }
        auto issue = [&](int buf, int k0) {
#pragma unroll
            for (int i = 0; i < PW; ++i)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(srcA[i] + k0),
                                                 (__attribute__((address_space(3))) void*)(smem + buf * 2 * OP + (PW * wave + i) * 1024), 16, 0, 0);
#pragma unroll
            for (int i = 0; i < PW; ++i)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(srcB[i] + k0),
                                                 (__attribute__((address_space(3))) void*)(smem + buf * 2 * OP + OP + (PW * wave + i) * 1024), 16, 0, 0);
        };
        auto mma_tile = [&](int buf, int k0) {
            const unsigned char* as = smem + buf * 2 * OP;
            const unsigned char* bs = as + OP;
            const bool diagB = g.maskB != MASK_NONE && k0 + BK > tj * BN && k0 < (tj + 1) * BN;    // wave-uniform
            const bool zeroB = g.maskB == MASK_LOWER && k0 >= (tj + 1) * BN;                         // the whole k-tile lies above B's diagonal
#pragma unroll
            for (int kk = 0; kk < BK / 8; ++kk) {
                const int slot = (4 * kk + fg) ^ swz;
                double2_t a[TW], b[TW];
#pragma unroll
                for (int m = 0; m < TW; ++m) a[m] = *reinterpret_cast<const double2_t*>(as + (wm * 16 * TW + 16 * m + fr) * 128 + slot * 16);
#pragma unroll
                for (int m = 0; m < TW; ++m) b[m] = *reinterpret_cast<const double2_t*>(bs + (wn * 16 * TW + 16 * m + fr) * 128 + slot * 16);
                if (diagB || zeroB) {
                    const int kmin = k0 + 8 * kk;
#pragma unroll
                    for (int m = 0; m < TW; ++m) {
                        const int bRow = tj * BN + wn * 16 * TW + 16 * m + fr;
#pragma unroll
                        for (int h = 0; h < 2; ++h) b[m][h] = (kmin + 2 * fg + h > bRow) ? 0.0 : b[m][h];
                    }
                }
#pragma unroll
                for (int h = 0; h < 2; ++h)
#pragma unroll
                    for (int m = 0; m < TW; ++m)
#pragma unroll
                        for (int n = 0; n < TW; ++n) acc[m][n] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[m][h], b[n][h], acc[m][n], 0, 0, 0);
            }
        };
        // three operand buffers, two k-tiles of loads in flight: one workgroup per CU has nothing else to cover a fetch
        // that misses the L2 (the operands of a piece come from the fabric once per group), and a k-tile of MFMAs is 1.95 us
        issue(0, kLo);
        if (nk > 1) issue(1, kLo + BK);
        if (nk > 1) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(2 * PW) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        for (int t = 0, buf = 0; t < nk; ++t) {
            const int nxt2 = buf == 0 ? 2 : buf - 1;                  // the buffer of tile t + 2 = the one tile t - 1 used
            if (t + 2 < nk) issue(nxt2, kLo + (t + 2) * BK);
            mma_tile(buf, kLo + t * BK);
            if (t + 2 < nk) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(2 * PW) : "memory");     // tile t + 1 has landed (t + 2 may be in flight)
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            buf = buf == 2 ? 0 : buf + 1;
        }
        const bool whole = kt0 == 0 && kt1 == g.KT;
        double* const slot = g.P + ((size_t)(2 * w + (it == itBeg ? 0 : 1)) * g.gx + tj) * (BM * BN);
#pragma unroll
        for (int m = 0; m < TW; ++m)
#pragma unroll
            for (int n = 0; n < TW; ++n)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int ii = wm * 16 * TW + m * 16 + (lane >> 4) + 4 * r, jj = wn * 16 * TW + n * 16 + (lane & 15);
                    if (!whole) { slot[ii * BN + jj] = acc[m][n][r]; continue; }
                    const int i = ti * BM + ii, j = tj * BN + jj;
                    if (i >= g.M || j >= g.N) continue;
                    double* cp = g.C + (size_t)i * g.ldc + j;
                    double v = g.alpha * acc[m][n][r];
                    if (g.beta != 0.0) v += g.beta * (*cp);
                    *cp = v;
                }
        it += nk;
    }
}

// one workgroup per output tile: nothing to do where one stream-K workgroup walked the whole tile; otherwise the pieces in ascending w
__global__ __launch_bounds__(256)
void gemm_streamk_fixup_kernel(StreamKArgs g)
{
    constexpr int BM = 128, BN = 128;
    const int tile = (int)blockIdx.x;
    const int ti = tile / g.gx, tj = tile % g.gx;
    const long long t0 = (long long)ti * g.KT, t1 = t0 + g.KT;
    const int wa = streamk_owner(g, t0), wb = streamk_owner(g, t1 - 1);
    if (wa == wb) return;
    for (int e = threadIdx.x; e < BM * BN; e += 256) {
        const int ii = e / BN, jj = e % BN;
        const int i = ti * BM + ii, j = tj * BN + jj;
        if (i >= g.M || j >= g.N) continue;
        double sum = 0.0;
        for (int w = wa; w <= wb; ++w) {
            const long long first = streamk_first(g, w);
            sum += g.P[((size_t)(2 * w + (first >= t0 ? 0 : 1)) * g.gx + tj) * (BM * BN) + e];      // the piece starts with w's range, or behind a piece of an earlier tile row
        }
        double* cp = g.C + (size_t)i * g.ldc + j;
        double v = g.alpha * sum;
        if (g.beta != 0.0) v += g.beta * (*cp);
        *cp = v;
    }
}

// 64x64 leaf: A (lower triangle valid) -> X = inv(chol(A)), written lower + mirrored upper.
// One workgroup of 16 x 16 threads; thread (ty, tx) keeps ONE 4 x 4 register block c.  The factorisation and the
// triangular inverse advance together, one 4-wide block column per step, 16 steps of two barriers: the time of a
// leaf is its chain of barriers, LDS round trips and the serial factorisation of the 4 x 4 diagonal block, not its
// flops.  With L = chol(A) and D_k the inverse of L's k-th diagonal triangle, step k does
//   1. thread (k,k): factors its block (LDL^T steps with refined reciprocals on the dependent chain, the reciprocal
//      square roots beside it), publishes D_k;
//   2. the threads of block column k below the diagonal: L(i,k) = C(i,k) D_k^T, publish the panel L(:,k) and the
//      panel L(:,k) D_k; their block then starts over as the zero block of B.  The threads of block row k
//      publish their blocks of B (unit diagonal block for (k,k));
//   3. every thread (i,j) with i > k: j > k: C(i,j) -= L(i,k) L(j,k)^T (Cholesky trailing update);
//      j <= k: B(i,j) -= [L(i,k) D_k] B(k,j)   (forward substitution on B = I with the scaling by D deferred:
//      X(k,:) = D_k B(k,:), so the rows below subtract L(i,k) X(k,:) = [L(i,k) D_k] B(k,:)).
// A block holds C until its own column has been the panel and B afterwards, so one register block does for both;
// at the end X(i,j) = D_i B(i,j).
#ifndef DCA_RSQ_NEWTON
#define DCA_RSQ_NEWTON 1        // + the residual step: 7e-16 against LAPACK with 1 as with 2 (tools/experiments/inv_err.py)
#endif
__device__ __forceinline__ double rsqrt_refined(double a)
{
    double y = __builtin_amdgcn_rsq(a);
#pragma unroll
    for (int it = 0; it < DCA_RSQ_NEWTON; ++it) {
        const double h = 0.5 * y;
        const double e = __builtin_fma(-a * y, y, 1.0);     // 1 - a y^2
        y = __builtin_fma(h, e, y);
    }
    // last step on the residual of s = a y: one more correct digit than another Newton step on y
    const double sq = a * y;
    const double res = __builtin_fma(-sq, sq, a);
    return __builtin_fma(0.5 * y * y, res * y, y);          // y + y^3 (a - s^2) / 2
}

__device__ __forceinline__ double rcp_refined(double a)
{
    double y = __builtin_amdgcn_rcp(a);
#pragma unroll
    for (int it = 0; it < 2; ++it) {
        const double e = __builtin_fma(-a, y, 1.0);
        y = __builtin_fma(y, e, y);
    }
    return y;
}

#ifdef DCA_LEAF_TRACE
__device__ unsigned long long g_leaf_trace[32 * 8];
#endif
// NB = 64: 16 x 16 threads; NB = 128: 32 x 32 threads (one workgroup of 16 waves) -- the same steps on twice as many
// block columns.  A 128-leaf replaces two 64-leaves AND the four one-tile products between them (2 x 29 + 4 x 7.5 us
// of serial launches), which is where the recursion of the inverse spends a quarter of its time.
template <int NB>
__global__ __launch_bounds__((NB / 4) * (NB / 4))
void cholinv_leaf_kernel(double* __restrict__ M, int ld, int pivotBase, int* __restrict__ info)
{
    constexpr int n = NB;
    constexpr int TB = NB / 4;           // threads per side = 4-wide block columns
    // all three [m][64]: a thread's four values per m are one 32-byte run and neighbouring threads' runs are 32 bytes
    // apart ([row][4] put the 16 column-side reads of a wave's lanes 128 bytes apart: two banks, 8-way conflicts)
    __shared__ __attribute__((aligned(16))) double panel[2][4][n];    // L(:,k) transposed, rows of block rows > k valid
    __shared__ __attribute__((aligned(16))) double panelx[2][4][n];   // L(:,k) D_k transposed
    __shared__ __attribute__((aligned(16))) double brow[2][4][n];     // block row k of B, columns of block columns <= k valid
    __shared__ __attribute__((aligned(16))) double dinv[TB][16];      // D_k, 4 x 4 row-major (upper part zero)
    // block column major: the threads of one block column (the panel of a step) are neighbouring lanes of ONE wave,
    // so the serial panel phase occupies one wave instead of a few lanes in every wave of the workgroup
    const int tid = threadIdx.x;
#ifndef DCA_LEAF_ROW_MAJOR
    const int tx = tid / TB, ty = tid % TB;
#else
    const int ty = tid / TB, tx = tid % TB;
#endif
    const int r0 = 4 * ty, c0 = 4 * tx;

    double c[4][4];
    // diagonal threads need their full symmetric block, the others their block as stored (lower part of A valid)
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int cc = 0; cc < 4; ++cc) {
            const int i = r0 + r, j = c0 + cc;
            c[r][cc] = (tx < ty) ? M[(size_t)i * ld + j] : (tx == ty) ? M[(size_t)max(i, j) * ld + min(i, j)] : 0.0;
        }

#ifdef DCA_LEAF_TRACE
#define LEAF_T(slot) do { if (tid == DCA_LEAF_TRACE) g_leaf_trace[kb * 8 + (slot)] = wall_clock64(); } while (0)
#else
#define LEAF_T(slot) do { } while (0)
#endif
    for (int kb = 0; kb < TB; ++kb) {
        const int cur = kb & 1;
        LEAF_T(0);
#ifndef DCA_LEAF_ABLATE
#define DCA_LEAF_ABLATE 0       // tools/experiments/leaf_bench.hip: 1 no diagonal factorisation, 2 no panel products, 3 no updates
#endif
        if (ty == kb && tx == kb && DCA_LEAF_ABLATE != 1) {
            double l[4][4], xi[4][4], rs[4];
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int cc = 0; cc < 4; ++cc) { l[r][cc] = 0.0; xi[r][cc] = 0.0; }
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const double dk = c[k][k];
                if (!(dk > 0.0)) atomicCAS(info, 0, pivotBase + 4 * kb + k + 1);
                const double ik = k < 3 ? rcp_refined(dk) : 0.0;   // on the chain to the next pivot (the last has none)
                rs[k] = rsqrt_refined(dk);                      // beside it
                double t[4];
#pragma unroll
                for (int r = k + 1; r < 4; ++r) t[r] = c[r][k] * ik;
#pragma unroll
                for (int r = k + 1; r < 4; ++r)
#pragma unroll
                    for (int cc = k + 1; cc <= r; ++cc) c[r][cc] = __builtin_fma(-t[r], c[cc][k], c[r][cc]);
                l[k][k] = dk * rs[k];
#pragma unroll
                for (int r = k + 1; r < 4; ++r) l[r][k] = c[r][k] * rs[k];
            }
            // D = inverse of the lower triangle l
#pragma unroll
            for (int cc = 0; cc < 4; ++cc) {
                xi[cc][cc] = rs[cc];
#pragma unroll
                for (int r = cc + 1; r < 4; ++r) {
                    double acc = 0.0;
#pragma unroll
                    for (int m = cc; m < r; ++m) acc = __builtin_fma(l[r][m], xi[m][cc], acc);
                    xi[r][cc] = -rs[r] * acc;
                }
            }
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int cc = 0; cc < 4; ++cc) {
                    dinv[kb][r * 4 + cc] = xi[r][cc];
                    c[r][cc] = (r == cc) ? 1.0 : 0.0;            // B(k,k) = I
                    brow[cur][r][c0 + cc] = c[r][cc];
                }
        }
        LEAF_T(1);
        __syncthreads();
        LEAF_T(2);
        if (tx == kb && ty > kb && DCA_LEAF_ABLATE != 2) {
            double xi[4][4], lb[4][4];
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int cc = 0; cc < 4; ++cc) xi[r][cc] = dinv[kb][r * 4 + cc];
            // L(ty,kb) = C(ty,kb) * D^T.  Summation index outermost in all the small products of this kernel: a
            // dependent f64 FMA issues ~30 clocks after its predecessor, 16 independent ones go back to back.
            double lx[4][4];
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int m = 0; m < 4; ++m) lb[r][m] = c[r][0] * xi[m][0];
#pragma unroll
            for (int cc = 1; cc < 4; ++cc)
#pragma unroll
                for (int r = 0; r < 4; ++r)
#pragma unroll
                    for (int m = cc; m < 4; ++m) lb[r][m] = __builtin_fma(c[r][cc], xi[m][cc], lb[r][m]);
            // (L D)[r][m] = sum_{k >= m} L[r][k] D[k][m]
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int m = 0; m < 4; ++m) lx[r][m] = lb[r][m] * xi[m][m];
#pragma unroll
            for (int k = 1; k < 4; ++k)
#pragma unroll
                for (int r = 0; r < 4; ++r)
#pragma unroll
                    for (int m = 0; m < k; ++m) lx[r][m] = __builtin_fma(lb[r][k], xi[k][m], lx[r][m]);
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int m = 0; m < 4; ++m) {
                    panel[cur][m][r0 + r] = lb[r][m];
                    panelx[cur][m][r0 + r] = lx[r][m];
                    c[r][m] = 0.0;                               // the block starts over as B(ty,kb) = 0
                }
        } else if (ty == kb && tx < kb) {
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int cc = 0; cc < 4; ++cc) brow[cur][r][c0 + cc] = c[r][cc];
        }
        LEAF_T(3);
        __syncthreads();
        LEAF_T(4);
        if (ty > kb && DCA_LEAF_ABLATE != 3) {
            // one update for both roles: rows from panel / panelx, the other factor from panel (as columns) / brow
            const bool chol = tx > kb;
            const double* prow = chol ? &panel[cur][0][r0] : &panelx[cur][0][r0];      // pr(r, m) = prow[m * n + r]
            const double* qbase = chol ? &panel[cur][0][c0] : &brow[cur][0][c0];       // q(m, cc) = qbase[m * n + cc]
            double pr[4][4], q[4][4];
#pragma unroll
            for (int m = 0; m < 4; ++m)
#pragma unroll
                for (int r = 0; r < 4; ++r) { pr[r][m] = prow[m * n + r]; q[m][r] = qbase[m * n + r]; }
#pragma unroll
            for (int m = 0; m < 4; ++m)
#pragma unroll
                for (int r = 0; r < 4; ++r)
#pragma unroll
                    for (int cc = 0; cc < 4; ++cc) c[r][cc] = __builtin_fma(-pr[r][m], q[m][cc], c[r][cc]);
        }
        LEAF_T(5);
    }
    __syncthreads();
    if (tx <= ty) {
        // X(ty,tx) = D_ty * B(ty,tx)
        double xi[4][4], x[4][4];
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int cc = 0; cc < 4; ++cc) xi[r][cc] = dinv[ty][r * 4 + cc];
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int cc = 0; cc < 4; ++cc) x[r][cc] = xi[r][0] * c[0][cc];
#pragma unroll
        for (int m = 1; m < 4; ++m)
#pragma unroll
            for (int r = m; r < 4; ++r)
#pragma unroll
                for (int cc = 0; cc < 4; ++cc) x[r][cc] = __builtin_fma(xi[r][m], c[m][cc], x[r][cc]);
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int cc = 0; cc < 4; ++cc)
                if (c0 + cc <= r0 + r) {
                    M[(size_t)(r0 + r) * ld + c0 + cc] = x[r][cc];
                    M[(size_t)(c0 + cc) * ld + r0 + r] = x[r][cc];
                }
    }
}


// ---------------------------------------------------------------------------------------------------------------
// The same leaf on the f64 matrix cores.  The register-block leaf above spends half of its time in the rank-4 updates
// (64 FMAs per thread and step on the vector ALU); a rank-4 update of a 16 x 16 tile is exactly ONE
// v_mfma_f64_16x16x4_f64.  Here the block lives in MFMA accumulator layout (tile (ti, tj), ti >= tj: lane holds column
// lane & 15, rows (lane >> 4) + 4 r), the lower tiles dealt round-robin to the WORKER waves 1 .. WAVES-1, and wave 0
// does nothing but the serial part: the 4 x 4 diagonal blocks.  A step (4 columns, index kb):
//   a. the owners publish the 4 columns of C on and below the diagonal (colbuf), the 4 rows of B of block row kb
//      (rowbuf) and the NEXT diagonal block as it stands (nextdiag), and restart B(i, kb) = 0 below the block;
//   c. [needs D_kb] one lane per (m, index): below the block  L(i, kb) = C(i, kb) D^T  -> left[m][i] = right[m][i];
//      up to the block  X(kb, j) = D B(kb, j)  (D itself inside the block) -> right[m][j], left = 0, and X's rows --
//      final now -- go straight to memory (lower part + mirrored upper);
//   d. every tile with rows below the block:  acc -= left^T right  (one MFMA per tile): for columns right of the
//      block this is the Cholesky update C(i, j) -= L(i, kb) L(j, kb)^T, for the others the forward substitution
//      B(i, j) -= L(i, kb) X(kb, j); rows up to the block have left = 0 and do not change;
//   b. meanwhile wave 0 (look-ahead): the next diagonal block  C(kb+1, kb+1) - L(kb+1, kb) L(kb+1, kb)^T  from
//      nextdiag and colbuf, its Cholesky triangle and D_{kb+1} = the triangle's inverse -- the dependent chain that
//      bounds the leaf -- overlapped with d. and with a. of the next step.
// Two barriers per step.  Same arithmetic as the register-block leaf up to the order of the FMAs.
__device__ __forceinline__ void factor_diag_block(double (&c)[4][4], double* __restrict__ dinvOut, int* __restrict__ info, int pivot0)
{
    double l[4][4], xi[4][4], rs[4];
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int cc = 0; cc < 4; ++cc) { l[r][cc] = 0.0; xi[r][cc] = 0.0; }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const double dk = c[k][k];
        if (!(dk > 0.0)) atomicCAS(info, 0, pivot0 + k + 1);
        const double ik = k < 3 ? rcp_refined(dk) : 0.0;       // on the chain to the next pivot (the last has none)
        rs[k] = rsqrt_refined(dk);                             // beside it
        double t[4];
#pragma unroll
        for (int r = k + 1; r < 4; ++r) t[r] = c[r][k] * ik;
#pragma unroll
        for (int r = k + 1; r < 4; ++r)
#pragma unroll
            for (int cc = k + 1; cc <= r; ++cc) c[r][cc] = __builtin_fma(-t[r], c[cc][k], c[r][cc]);
        l[k][k] = dk * rs[k];
#pragma unroll
        for (int r = k + 1; r < 4; ++r) l[r][k] = c[r][k] * rs[k];
    }
#pragma unroll
    for (int cc = 0; cc < 4; ++cc) {
        xi[cc][cc] = rs[cc];
#pragma unroll
        for (int r = cc + 1; r < 4; ++r) {
            double a2 = 0.0;
#pragma unroll
            for (int m = cc; m < r; ++m) a2 = __builtin_fma(l[r][m], xi[m][cc], a2);
            xi[r][cc] = -rs[r] * a2;
        }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int cc = 0; cc < 4; ++cc) dinvOut[r * 4 + cc] = xi[r][cc];
}

// Barrier of the leaf's steps: what the waves exchange goes through LDS, so only the LDS counter is drained.  __syncthreads()
// also waits for the global stores of the finished X rows (step c.), which nobody in the kernel reads: ~120 ns of the 1.4 us step.
__device__ __forceinline__ void leaf_step_barrier()
{
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

#ifndef DCA_LEAF_WAVES128
#define DCA_LEAF_WAVES128 8          // waves of the 128-leaf (experiments: 12, 16 -- fewer tiles per worker wave, fewer registers for the chain)
#endif
template <int NB>
__global__ __launch_bounds__(NB == 128 ? DCA_LEAF_WAVES128 * 64 : 320)
void cholinv_leaf_mfma_kernel(double* __restrict__ M, int ld, int pivotBase, int* __restrict__ info)
{
    DCA_CHAIN_PRIO();
    constexpr int NT = NB / 16;                       // tiles per side
    constexpr int NTILES = NT * (NT + 1) / 2;         // lower tiles
    constexpr int WAVES = NB == 128 ? DCA_LEAF_WAVES128 : 5;           // 8 x 64 = 4 NB lanes for step c.; 256 VGPRs for the unrolled 4 x 4 chain
    constexpr int WORKERS = WAVES - 1;
    constexpr int SLOTS = (NTILES + WORKERS - 1) / WORKERS;
    constexpr int STEPS = NB / 4;
    __shared__ __attribute__((aligned(16))) double colbuf[2][4][NB];
    __shared__ __attribute__((aligned(16))) double rowbuf[4][NB];
    __shared__ __attribute__((aligned(16))) double leftP[4][NB];
    __shared__ __attribute__((aligned(16))) double rightP[4][NB];
    __shared__ __attribute__((aligned(16))) double nextdiag[2][16];
    __shared__ __attribute__((aligned(16))) double dinv[2][16];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lc = lane & 15, lr = lane >> 4;

    // tile t of the row-major enumeration of the lower triangle -> (ti, tj); slot sl of worker wave w is tile (w-1) + sl * WORKERS
    int tI[SLOTS], tJ[SLOTS];
    double4_t acc[SLOTS];
#pragma unroll
    for (int sl = 0; sl < SLOTS; ++sl) {
        const int t = wave - 1 + sl * WORKERS;
        int ti = 0;
        while ((ti + 1) * (ti + 2) / 2 <= t) ++ti;
        const int tj = t - ti * (ti + 1) / 2;
        tI[sl] = (wave > 0 && t < NTILES) ? ti : -1;
        tJ[sl] = tj;
        acc[sl] = (double4_t){0.0, 0.0, 0.0, 0.0};
        if (tI[sl] >= 0) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int i = 16 * ti + lr + 4 * r, j = 16 * tj + lc;
                acc[sl][r] = M[(size_t)max(i, j) * ld + min(i, j)];          // diagonal tiles: the full symmetric tile
            }
            if (ti == 0 && tj == 0 && (lc >> 2) == 0) nextdiag[0][lr * 4 + (lc & 3)] = acc[sl][0];   // block (0, 0)
        }
    }
    __syncthreads();
    if (tid == 0) {
        double c[4][4];
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int cc = 0; cc < 4; ++cc) c[r][cc] = nextdiag[0][max(r, cc) * 4 + min(r, cc)];
        factor_diag_block(c, dinv[0], info, pivotBase);
    }

    for (int kb = 0; kb < STEPS; ++kb) {
        const int T = kb >> 2, s = kb & 3, k0 = 4 * kb, cur = kb & 1;
        LEAF_T(0);
        // ---- a. publish the step's columns of C, the rows of B of block row kb and the next diagonal block
#pragma unroll
        for (int sl = 0; sl < SLOTS; ++sl) {
            if (tI[sl] < T) continue;                                           // wave-uniform (also skips empty slots: -1)
            if (tJ[sl] == T && (lc >> 2) == s) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int row = 16 * tI[sl] + lr + 4 * r;
                    colbuf[cur][lc & 3][row] = acc[sl][r];
                    if (row > k0 + 3) acc[sl][r] = 0.0;                         // B(i, kb) starts over as the zero block
                }
            }
            if (tI[sl] == T) {
                const double v = s == 0 ? acc[sl][0] : s == 1 ? acc[sl][1] : s == 2 ? acc[sl][2] : acc[sl][3];
                rowbuf[lr][16 * tJ[sl] + lc] = v;                               // rows k0 + lr of B (columns left of the block)
            }
            const int Tn = (kb + 1) >> 2, sn = (kb + 1) & 3;
            if (kb + 1 < STEPS && tI[sl] == Tn && tJ[sl] == Tn && (lc >> 2) == sn) {
                const double v = sn == 0 ? acc[sl][0] : sn == 1 ? acc[sl][1] : sn == 2 ? acc[sl][2] : acc[sl][3];
                nextdiag[cur][lr * 4 + (lc & 3)] = v;                           // C(kb+1, kb+1) before this step's update
            }
        }
        LEAF_T(1);
        leaf_step_barrier();
        LEAF_T(2);
        // ---- c. panel below the block, X rows up to it (one lane per (m, index))
        if (tid < 4 * NB) {
            // indices dealt from the top: wave 0 -- whose serial chain bounds the step -- then holds the rows BELOW the block
            // for most of the leaf (four FMAs and two LDS stores) instead of the X rows with their global stores
            // (phase stamps: 280 -> 160 ns of its 1.44 us step)
            const int m = tid / NB, idx = NB - 1 - tid % NB;
            double d[4];
#pragma unroll
            for (int p = 0; p < 4; ++p) d[p] = dinv[cur][m * 4 + p];           // row m of D (lower: d[p] = 0 for p > m)
            if (idx > k0 + 3) {
                double v = colbuf[cur][0][idx] * d[0];
#pragma unroll
                for (int p = 1; p < 4; ++p) v = __builtin_fma(colbuf[cur][p][idx], d[p], v);   // L(idx, kb)[m] = sum_p C[idx][p] D[m][p]
                leftP[m][idx] = v;
                rightP[m][idx] = v;
            } else {
                double x;
                if (idx >= k0) x = d[idx - k0];                                 // X(kb, kb) = D
                else {
                    x = d[0] * rowbuf[0][idx];
#pragma unroll
                    for (int p = 1; p < 4; ++p) x = __builtin_fma(d[p], rowbuf[p][idx], x);     // X(kb, j)[m] = sum_p D[m][p] B[p][j]
                }
                leftP[m][idx] = 0.0;
                rightP[m][idx] = x;
                const int row = k0 + m;
                if (idx <= row) {
                    M[(size_t)row * ld + idx] = x;
                    if (idx < row) M[(size_t)idx * ld + row] = x;
                }
            }
        }
        LEAF_T(3);
        leaf_step_barrier();
        LEAF_T(4);
        // ---- b. (wave 0, look-ahead) the next diagonal block and its D
        if (tid == 0 && kb + 1 < STEPS) {
            double d[4][4], l1[4][4], c[4][4];
#pragma unroll
            for (int m = 0; m < 4; ++m)
#pragma unroll
                for (int p = 0; p < 4; ++p) d[m][p] = dinv[cur][m * 4 + p];
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int m = 0; m < 4; ++m) {                                   // L(kb+1, kb)[r][m] = sum_{p <= m} C[r][p] D[m][p]
                    double v = colbuf[cur][0][k0 + 4 + r] * d[m][0];
#pragma unroll
                    for (int p = 1; p < 4; ++p)
                        if (p <= m) v = __builtin_fma(colbuf[cur][p][k0 + 4 + r], d[m][p], v);
                    l1[r][m] = v;
                }
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int cc = 0; cc <= r; ++cc) {
                    double v = nextdiag[cur][r * 4 + cc];
#pragma unroll
                    for (int m = 0; m < 4; ++m) v = __builtin_fma(-l1[r][m], l1[cc][m], v);
                    c[r][cc] = v;
                    c[cc][r] = v;
                }
            factor_diag_block(c, dinv[cur ^ 1], info, pivotBase + k0 + 4);
        }
        LEAF_T(5);
        // ---- d. rank-4 update of every tile that has rows below the block.  (Issuing the next step's tiles first, or all
        // operand fragments before the MFMAs, measured slower: 52 / 49 against 45 us per 128-leaf.)
#pragma unroll
        for (int sl = 0; sl < SLOTS; ++sl) {
            if (tI[sl] < 0 || 16 * tI[sl] + 15 <= k0 + 3) continue;             // wave-uniform
            const double a = -leftP[lr][16 * tI[sl] + lc];
            const double b = rightP[lr][16 * tJ[sl] + lc];
            acc[sl] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[sl], 0, 0, 0);
        }
        LEAF_T(6);
    }
}

// ---------------------------------------------------------------------------------------------------------------
// Round 6: the leaf in steps of SIXTEEN columns (NB = 64 ... 256).
// The leaf above advances four columns per step: two workgroup barriers, three LDS round trips and the 4 x 4 diagonal block
// factored by ONE lane (600 ns) for every four columns -- 1.44 us per step, 46 us per 128 columns, and with 8 x 189 registers
// it needs a CU of its own.  Here a step is one 16-column tile column T:
//   1. [D16 = inv(chol(C(T,T))) is in LDS]  L(i,T) = C(i,T) D16^T for the tile rows below (4 MFMAs per tile, operands through
//      LDS), X(T,j) = D16 B(T,j) for the tiles left of the diagonal (4 MFMAs on the owner's own accumulators -- the
//      accumulator layout IS the B-operand layout), X(T,T) = D16; X rows go to memory, all of it to the `panel` buffer;
//   2. the tiles of column T+1 take their rank-16 update first and publish themselves (next diagonal tile, next column);
//   3. wave 0 factors and inverts the next diagonal tile WHILE the other waves update the remaining tiles.
// The 16 x 16 diagonal tile is done by one WAVE with one matrix row per lane (registers, v_readlane broadcasts of the pivot
// row, no LDS, no barrier): Gaussian elimination of [S | I] with the multipliers s_rk / p_k (reciprocal: one seed + one
// cubic correction, 4 dependent operations) leaves the inverse of the unit factor in the right half, scaled at the end by
// 1 / sqrt(p_r).  Per column the dependent chain is 7 operations instead of ~11 per column of the one-lane 4 x 4 form.
// Seven worker waves + the chain wave; NB <= 128 fits in 128 registers per lane (a CU that runs one of the sweep's update
// workgroups still has room for it), NB = 256 (136 tiles) needs 256.
__device__ __forceinline__ double lane_bcast(double v, int src)
{
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), src);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(v), src);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double rcp_cubic(double a)
{
    // v_rcp_f64 is good to ~2^-23; y (1 + e + e^2), e = 1 - a y, leaves e^3 ~ 2^-69
    const double y = __builtin_amdgcn_rcp(a);
    const double e = __builtin_fma(-a, y, 1.0);
    const double t = __builtin_fma(e, e, e);
    return __builtin_fma(y, t, y);
}
// S: 16 x 16 symmetric tile in LDS (row-major, both halves); D <- inv(chol(S)) (lower, zeros above).  One whole wave.
// One COLUMN of [S | I] per lane (lanes 0 .. 15 the columns of S, 16 .. 31 those of the right half): eliminating column k
// is  v[r] -= m_r v[k]  in every lane, with the multipliers m_r = S[r][k] / p_k formed in lane k and broadcast -- 15 - k
// broadcasts per column serve both halves (one matrix ROW per lane needed 16: the pivot row of both halves).
__device__ __forceinline__ void factor16(const double* __restrict__ S, double* __restrict__ D, int lane, int* __restrict__ info, int pivot0)
{
    const int c = lane & 15;
    const bool right = (lane & 16) != 0;
    double v[16], rs[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const double sv = S[c * 16 + r];                    // column c = row c
        v[r] = right ? (r == c ? 1.0 : 0.0) : sv;
    }
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        const double p = lane_bcast(v[k], k);
        if (!(p > 0.0) && lane == 0) atomicCAS(info, 0, pivot0 + k + 1);
        const double ik = rcp_cubic(p);
        rs[k] = lane_bcast(rsqrt_refined(p), 0);            // beside the chain; uniform: kept in scalar registers
#pragma unroll
        for (int r = k + 1; r < 16; ++r) {
            const double m = lane_bcast(v[r] * ik, k);
            v[r] = __builtin_fma(-m, v[k], v[r]);
        }
    }
    if (lane >= 16 && lane < 32) {
#pragma unroll
        for (int r = 0; r < 16; ++r) D[r * 16 + c] = (r >= c) ? rs[r] * v[r] : 0.0;
    }
}

#ifndef DCA_LEAF16_ABLATE
#define DCA_LEAF16_ABLATE 0      // tools/experiments/leaf16_bench.hip, timing only: 1 no diagonal tiles, 2 no panel, 4 no updates, 8 no X stores
#endif
template <int NB>
__device__ __forceinline__ void cholinv_leaf16_body(double* __restrict__ M, int ld, int pivotBase, int* __restrict__ info)
{
    DCA_CHAIN_PRIO();
    constexpr int NT = NB / 16;
    constexpr int NTILES = NT * (NT + 1) / 2;
    constexpr int WORKERS = 7;
    constexpr int SLOTS = (NTILES + WORKERS - 1) / WORKERS;
    constexpr int PS = NB + 2;                        // row stride of the k-major buffers (doubles)
    extern __shared__ __attribute__((aligned(16))) unsigned char dca_gemm_smem[];
    double* const colbuf = reinterpret_cast<double*>(dca_gemm_smem);        // [16][PS]: C(i, T) of the tile rows below T, [column][row]
    double* const panel = colbuf + 16 * PS;                                 // [16][PS]: [m][idx] = X(T, idx)[m] up to the tile, L(idx, T)[m] below
    double* const diagbuf = panel + 16 * PS;                                // [16][16]: the next diagonal tile
    double* const dinv = diagbuf + 256;                                     // [16][16]: D16
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lc = lane & 15, lr = lane >> 4;

    if (wave == 0) {
        // the chain wave: nothing but the diagonal tiles (same barriers as the workers below)
        __syncthreads();
        factor16(diagbuf, dinv, lane, info, pivotBase);
        leaf_step_barrier();
#pragma unroll 1
        for (int T = 0; T < NT; ++T) {
            leaf_step_barrier();
            if (T + 1 == NT) break;
            if (!(DCA_LEAF16_ABLATE & 1)) factor16(diagbuf, dinv, lane, info, pivotBase + 16 * (T + 1));
            leaf_step_barrier();
        }
        return;
    }

    int tI[SLOTS], tJ[SLOTS];
    double4_t acc[SLOTS];
#pragma unroll
    for (int sl = 0; sl < SLOTS; ++sl) {
        const int t = wave - 1 + sl * WORKERS;
        int ti = 0;
        while ((ti + 1) * (ti + 2) / 2 <= t) ++ti;
        tI[sl] = t < NTILES ? ti : -1;
        tJ[sl] = t - ti * (ti + 1) / 2;
        acc[sl] = (double4_t){0.0, 0.0, 0.0, 0.0};
        if (tI[sl] >= 0) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int i = 16 * ti + lr + 4 * r, j = 16 * tJ[sl] + lc;
                acc[sl][r] = M[(size_t)max(i, j) * ld + min(i, j)];          // diagonal tiles: the full symmetric tile
            }
        }
    }
    // the tiles of column T1 BELOW the diagonal publish themselves as C(i, T1) and start over as B(i, T1) = 0
    auto publish_column = [&](int T1) {
#pragma unroll
        for (int sl = 0; sl < SLOTS; ++sl) {
            if (tI[sl] <= T1 || tJ[sl] != T1) continue;                      // wave-uniform (also passes over empty slots: -1)
#pragma unroll
            for (int r = 0; r < 4; ++r) colbuf[lc * PS + 16 * tI[sl] + lr + 4 * r] = acc[sl][r];
            acc[sl] = (double4_t){0.0, 0.0, 0.0, 0.0};
        }
    };
    auto publish_diag = [&](int sl) {
#pragma unroll
        for (int r = 0; r < 4; ++r) diagbuf[(lr + 4 * r) * 16 + lc] = acc[sl][r];
    };
    auto update = [&](int sl) {
        const int i0 = 16 * tI[sl], j0 = 16 * tJ[sl];
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            const double a = -panel[(4 * kk + lr) * PS + i0 + lc];
            const double b = panel[(4 * kk + lr) * PS + j0 + lc];
            acc[sl] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[sl], 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);          // one tile's operands at a time: hoisting every slot's LDS reads costs more registers than a wave has
    };
    auto panel_row = [&](int i) {                   // L(i, T) = C(i, T) D16^T -> panel
        double4_t l = (double4_t){0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int kk = 0; kk < 4; ++kk)
            l = __builtin_amdgcn_mfma_f64_16x16x4f64(colbuf[(4 * kk + lr) * PS + 16 * i + lc], dinv[lc * 16 + 4 * kk + lr], l, 0, 0, 0);
#pragma unroll
        for (int r = 0; r < 4; ++r) panel[lc * PS + 16 * i + lr + 4 * r] = l[r];
        __builtin_amdgcn_sched_barrier(0);
    };
    // rows 16 T .. 16 T + 15 of X (final: they stand in `panel`) go to memory, lower part and mirror; all worker lanes, no per-tile addresses
    auto store_x_rows = [&](int T) {
        if (DCA_LEAF16_ABLATE & 8) return;
        const int W = 16 * (T + 1);
#pragma unroll 1
        for (int e = tid - 64; e < 16 * W; e += 64 * WORKERS) {
            const int m = e / W, col = e - m * W, row = 16 * T + m;
            const double x = panel[m * PS + col];
            if (col <= row) {
                M[(size_t)row * ld + col] = x;
                if (col < row) M[(size_t)col * ld + row] = x;
            }
        }
    };
#pragma unroll
    for (int sl = 0; sl < SLOTS; ++sl)
        if (tI[sl] == 0) publish_diag(sl);
    publish_column(0);
    __syncthreads();
    leaf_step_barrier();

#pragma unroll 1
    for (int T = 0; T < NT; ++T) {
        // ---- A. the panel of the step.  The owner of the NEXT diagonal tile makes that tile's row of the panel itself, updates the
        // tile and publishes it: the chain wave waits for nothing else
#pragma unroll
        for (int sl = 0; sl < SLOTS; ++sl) {
            if (tI[sl] != T + 1 || tJ[sl] != T + 1) continue;
            if (!(DCA_LEAF16_ABLATE & 2)) panel_row(T + 1);
            update(sl);                                                      // reads what this wave has just written (LDS keeps a wave's order)
            publish_diag(sl);
        }
#pragma unroll 1
        for (int i = T + 1 + wave; i < NT && !(DCA_LEAF16_ABLATE & 2); i += WORKERS) panel_row(i);
#pragma unroll
        for (int sl = 0; sl < SLOTS; ++sl) {
            if (tI[sl] != T) continue;
            double4_t x = (double4_t){0.0, 0.0, 0.0, 0.0};
            if (tJ[sl] < T) {
#pragma unroll
                for (int r = 0; r < 4; ++r) x = __builtin_amdgcn_mfma_f64_16x16x4f64(dinv[lc * 16 + lr + 4 * r], acc[sl][r], x, 0, 0, 0);
            } else {
#pragma unroll
                for (int r = 0; r < 4; ++r) x[r] = dinv[(lr + 4 * r) * 16 + lc];
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) panel[(lr + 4 * r) * PS + 16 * tJ[sl] + lc] = x[r];
            __builtin_amdgcn_sched_barrier(0);
        }
        leaf_step_barrier();
        if (T + 1 == NT) break;
        // ---- B. everybody's updates, column T + 1 first (wave 0 factors the next diagonal tile meanwhile)
#pragma unroll
        for (int sl = 0; sl < SLOTS; ++sl)
            if (tI[sl] > T + 1 && tJ[sl] == T + 1) update(sl);
        publish_column(T + 1);
#pragma unroll
        for (int sl = 0; sl < SLOTS; ++sl)
            if (tI[sl] > T && tJ[sl] != T + 1 && !(DCA_LEAF16_ABLATE & 4)) update(sl);      // (T + 1, T + 1) is in column T + 1: done in A
        store_x_rows(T);
        leaf_step_barrier();
    }
    store_x_rows(NT - 1);
}
// 144 registers: two of its waves per SIMD next to one wave of an update workgroup (224)
template <int NB>
__global__ __launch_bounds__(512) __attribute__((amdgpu_num_vgpr(144)))
void cholinv_leaf16_small_kernel(double* __restrict__ M, int ld, int pivotBase, int* __restrict__ info) { cholinv_leaf16_body<NB>(M, ld, pivotBase, info); }
template <int NB>
__global__ __launch_bounds__(512)
void cholinv_leaf16_kernel(double* __restrict__ M, int ld, int pivotBase, int* __restrict__ info) { cholinv_leaf16_body<NB>(M, ld, pivotBase, info); }
template <int NB> constexpr size_t leaf16_lds_bytes() { return (size_t)(2 * 16 * (NB + 2) + 512) * sizeof(double); }

constexpr size_t kLeafLds = 0;

struct Arena {
    double* base; size_t cap, top = 0;
    double* alloc(size_t n) { if (top + n > cap) return nullptr; double* p = base + top; top += n; return p; }
};

// Dynamic LDS sizes above the default limit need an attribute per kernel AND PER DEVICE; set for all of them the first time an
// inverse runs on a device (contexts may live on several host threads and several devices).
int gemm_kernels_prepare(int device)
{
    static std::mutex mu;
    static std::vector<int> done;
    std::lock_guard<std::mutex> lk(mu);
    for (int d : done) if (d == device) return DCA_OK;
    const int small = (int)((size_t)2 * (32 + 32) * (64 + 2) * sizeof(double));
    HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_nt_f64_dma_kernel<4>), hipFuncAttributeMaxDynamicSharedMemorySize, 4 * 128 * 16 * 8));
    HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_nt_f64_dma_kernel<4, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, 2 * (128 + 64) * 16 * 8));
    HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_nt_f64_dma_kernel<4, 4, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, 8 * 128 * 16 * 8));
    HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_nt_f64_streamk_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 6 * 128 * 16 * 8));
    HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_nt_f64_small_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, small));
    HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_nt_f64_small_pair_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, small));
    HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_nt_f64_kernel<64>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)gemm_lds_bytes<64>()));
    HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(cholinv_leaf16_kernel<256>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)leaf16_lds_bytes<256>()));
    done.push_back(device);
    return DCA_OK;
}

// Background form of a product (side stream of the recursion): 128 x 128 tiles, issued as bands of tile rows of at most
// `maxWGs` workgroups each.  MI355X places one such workgroup per CU before it doubles up, so a launch of fewer
// workgroups than CUs leaves whole CUs to the kernels of another stream: the chain of single-workgroup leaves and
// few-tile products of a subtree runs next to it at its own pace (tools/experiments/overlap_bench.hip: 200 leaves
// 45.6 us each alone, 45.7 next to 255-workgroup launches that sustain 54 TF, 49 -- and the product at 25 TF -- next to
// 256).  WALK_ROWS only.
int launch_gemm_banded(dca_ctx* ctx, hipStream_t stream, GemmArgs g, int maxWGs)
{
    const int gx = (g.N + 127) / 128, gy = (g.M + 127) / 128;
    const int band = std::max(1, maxWGs / gx);
    for (int r = 0; r < gy; r += band) {
        g.row0 = r;
        hipLaunchKernelGGL(gemm_nt_f64_dma_kernel<4>, dim3(gx, std::min(band, gy - r)), dim3(256), (size_t)6 * 128 * 16 * sizeof(double), stream, g);
    }
    HIP_TRY(hipGetLastError());
    return DCA_OK;
}

// The block sweep's chain runs NEXT TO its update launches: there the few-tile products do better as fewer, larger workgroups
// (n = 10 048: 21.4 -> 20.6 ms with both bounds at 16), so the sweep lowers the two bounds for the launches it makes
thread_local int tl_small32_max = -1, tl_deep_max_tiles = -1;
static int small32_max()
{
    static const int v = getenv("DCA_GEMM_SMALL32_MAX") ? atoi(getenv("DCA_GEMM_SMALL32_MAX")) : 400;     // 0: the 64 x 64 deep kernel
    static const bool fixed = getenv("DCA_GEMM_SMALL32_MAX") != nullptr;
    return (!fixed && tl_small32_max >= 0) ? tl_small32_max : v;
}
static dim3 grid64(const GemmArgs& g) { return g.walk == WALK_COLUMNS_REVERSED ? dim3(g.M / BM, g.N / BN) : dim3(g.N / BN, g.M / BM); }

// both products on 32 x 32 tiles in ONE launch when each is small enough for that kernel; false: launch them one by one
bool launch_gemm_small_pair(dca_ctx* ctx, const GemmArgs& a, const GemmArgs& b)
{
    static const bool on = !(getenv("DCA_GEMM_PAIR") && atoi(getenv("DCA_GEMM_PAIR")) == 0);
    const dim3 ga = grid64(a), gb = grid64(b);
    if (!on || (long long)ga.x * ga.y > small32_max() || (long long)gb.x * gb.y > small32_max()) return false;
    const size_t lds = (size_t)2 * (32 + 32) * (64 + 2) * sizeof(double);
    const dim3 grid(std::max(ga.x, gb.x) * 2, std::max(ga.y, gb.y) * 2, 2);
    hipLaunchKernelGGL(gemm_nt_f64_small_pair_kernel, grid, dim3(256), lds, ctx->stream, a, (int)ga.x * 2, (int)ga.y * 2, b, (int)gb.x * 2, (int)gb.y * 2);
    return true;
}

int launch_gemm_on(hipStream_t stream, const GemmArgs& g)
{
    dim3 grid(g.N / BN, g.M / BM);
    if (g.walk == WALK_COLUMNS_REVERSED) grid = dim3(g.M / BM, g.N / BN);
    static const int deepMaxTilesEnv = getenv("DCA_GEMM_DEEP_MAX_TILES") ? atoi(getenv("DCA_GEMM_DEEP_MAX_TILES")) : 400;
    static const bool deepFixed = getenv("DCA_GEMM_DEEP_MAX_TILES") != nullptr;
    const int deepMaxTiles = (!deepFixed && tl_deep_max_tiles >= 0) ? tl_deep_max_tiles : deepMaxTilesEnv;
    const int small32Max = small32_max();
    // inverse at n = 10 048 / 4000 by this bound: 0 -> 27.1 / 4.96 ms, 16 -> 25.8 / 4.36, 128 -> 24.7 / 4.08, 400 -> 24.2 / 3.95, 1200 -> 23.9 / 4.01, 1600 -> 24.8 / 3.99
    if ((long long)grid.x * grid.y <= small32Max) {          // far fewer tiles than CUs: 32 x 32 tiles on four times as many CUs
        dim3 g32(grid.x * 2, grid.y * 2);
        const size_t lds = (size_t)2 * (32 + 32) * (64 + 2) * sizeof(double);
        hipLaunchKernelGGL(gemm_nt_f64_small_kernel, g32, dim3(256), lds, stream, g);
        return DCA_OK;
    }
    if ((long long)grid.x * grid.y <= deepMaxTiles) {       // under two workgroups per CU: latency bound (measured: 0 -> 37.7, 128 -> 37.0, 400 -> 36.4, 1600 -> 36.9 ms)
        hipLaunchKernelGGL(gemm_nt_f64_kernel<64>, grid, dim3(256), gemm_lds_bytes<64>(), stream, g);
    } else {
        static const bool ahead2 = getenv("DCA_GEMM_AHEAD2") && atoi(getenv("DCA_GEMM_AHEAD2")) != 0;
        static const bool dma = !(getenv("DCA_GEMM_DMA") && atoi(getenv("DCA_GEMM_DMA")) == 0);
        static const int dma128 = getenv("DCA_GEMM_DMA128") ? atoi(getenv("DCA_GEMM_DMA128")) : 1;     // 0 never, 1 by the rule below, 2 always
        if (dma) {
            // 128 x 128 tiles: 16 instead of 8 flop per operand byte (the 64 x 64 kernel sits at the L2 -> LDS fill limit), but a
            // quarter of the tiles on half of the slots: taken when the rounds of equal tiles still fill up
            const long long t64 = g.lowerOnly ? (long long)grid.x * (grid.x + 1) / 2 : (long long)grid.x * grid.y;
            const int gx = (int)(grid.x + 1) / 2, gy = (int)(grid.y + 1) / 2;
            const long long t128 = g.lowerOnly ? (long long)gx * (gx + 1) / 2 : (long long)gx * gy;
            auto fill = [](long long t, long long slots) { return (double)t / (double)(((t + slots - 1) / slots) * slots); };
            // measured inside the inverse at n = 10 048 (kernel trace, 64 vs 128): SYRK 2.84 -> 2.61 ms, L21 / T^T 2.26 -> 2.19,
            // X21 2.40 -> 2.29, but X^T X (both operands triangular: the 128-tiles waste work along two diagonals) 6.10 -> 7.05
            const int masks = (g.maskA != MASK_NONE) + (g.maskB != MASK_NONE);
            bool use128 = false;
            if (masks == 0) use128 = t128 >= 512 && fill(t128, 512) * 66.0 > fill(t64, 1024) * 59.0;
            else if (masks == 1) use128 = t128 >= 1024;
            if (dma128 == 0) use128 = false;
            if (dma128 == 2) use128 = true;
            static const int rect = getenv("DCA_GEMM_RECT") ? atoi(getenv("DCA_GEMM_RECT")) : 1;     // 0 never, 1 two triangular operands, 2 also one
            const bool walkOk = g.walk != WALK_COLUMNS_REVERSED;
            if (!use128 && walkOk && rect && (masks == 2 || (rect == 2 && masks == 1)) && (long long)grid.x * grid.y >= 2048) {
                // 128 x 64 tiles: grid.y counts 128-row tiles
                hipLaunchKernelGGL((gemm_nt_f64_dma_kernel<4, 2>), dim3(grid.x, gy), dim3(256), (size_t)2 * (128 + 64) * 16 * sizeof(double), stream, g);
            } else if (use128) {
                dim3 grid128(gx, gy);
                if (g.walk == WALK_COLUMNS_REVERSED) grid128 = dim3((g.M + 127) / 128, (g.N + 127) / 128);
                hipLaunchKernelGGL(gemm_nt_f64_dma_kernel<4>, grid128, dim3(256), (size_t)4 * 128 * 16 * sizeof(double), stream, g);
            } else {
                hipLaunchKernelGGL(gemm_nt_f64_dma_kernel<2>, grid, dim3(256), (size_t)4 * 64 * 16 * sizeof(double), stream, g);
            }
        }
        else if (ahead2) hipLaunchKernelGGL((gemm_nt_f64_kernel<16, true>), grid, dim3(256), gemm_lds_bytes<16>(), stream, g);
        else hipLaunchKernelGGL(gemm_nt_f64_kernel<16>, grid, dim3(256), gemm_lds_bytes<16>(), stream, g);
    }
    return DCA_OK;
}
int launch_gemm(dca_ctx* ctx, const GemmArgs& g) { return launch_gemm_on(ctx->stream, g); }

// Side streams of the recursion, one per depth (a node's background product must not queue behind its ancestors'); the
// context's own stream carries the critical path.
constexpr int kSideDepths = DCA_SIDE_DEPTHS;
// Stream and event creation costs about a millisecond each -- more than the whole inverse of a small matrix -- so the sets are
// made once per process and device and lent to one inverse at a time (contexts of several host threads get a set each).
struct SideSet {
    hipStream_t s[kSideDepths] = {};
    hipStream_t masked[2] = {};       // round 6 (block sweep): two streams that may use only `maskedCus` of the CUs (the rest stay the chain's)
    int maskedCus = 0;
    // round 6: the sweep's two side streams, PROBED to run concurrently with the context's stream and with each other (see sweep_streams)
    std::vector<hipStream_t> extra;
    hipStream_t selFor = nullptr, selRest = nullptr, selSide = nullptr;
    int* probe = nullptr;             // device: flag, result
    unsigned* reserved = nullptr;     // device: bit per CU (cu_key()), the CUs the update workgroups leave to the chain's kernels
    int reservedPerXcd = 0;
    hipEvent_t fork[kSideDepths] = {}, join[kSideDepths] = {}, mid[kSideDepths] = {};
    int device = -1;
    bool busy = false;
};
std::mutex g_sideMu;
std::vector<SideSet*> g_sideSets;          // never destroyed: the HIP runtime may be gone at exit

SideSet* side_set_acquire(int device)
{
    std::lock_guard<std::mutex> lk(g_sideMu);
    for (SideSet* S : g_sideSets)
        if (S->device == device && !S->busy) { S->busy = true; return S; }
    SideSet* S = new SideSet();
    S->device = device;
    // plain streams: with stream priorities (context stream highest, these lowest) the chain's few-tile products ran 3 - 10
    // times slower whenever a side stream had work (10-workgroup launches at depth 2 cost the inverse 10 ms)
    bool good = true;
    for (int d = 0; d < kSideDepths && good; ++d) {
        good = hipStreamCreateWithFlags(&S->s[d], hipStreamNonBlocking) == hipSuccess &&
               hipEventCreateWithFlags(&S->fork[d], hipEventDisableTiming) == hipSuccess &&
               hipEventCreateWithFlags(&S->join[d], hipEventDisableTiming) == hipSuccess &&
               hipEventCreateWithFlags(&S->mid[d], hipEventDisableTiming) == hipSuccess;
    }
    if (!good) { delete S; return nullptr; }      // the products then run in line
    S->busy = true;
    g_sideSets.push_back(S);
    return S;
}
// the set's two CU-masked streams, made the first time a sweep asks for them (`cus` CUs, striped over the XCDs: bit i of the
// mask is a CU of XCD i % 8, tools/experiments/cumask_probe.hip); false: no such streams, the caller uses the plain ones
bool side_set_masked(SideSet* S, int cus)
{
    if (S->maskedCus == cus && S->masked[0] && S->masked[1]) return true;
    for (hipStream_t& m : S->masked) if (m) { hipStreamDestroy(m); m = nullptr; }
    S->maskedCus = 0;
    uint32_t mask[8] = {};
    for (int i = 0; i < cus && i < 256; ++i) mask[i / 32] |= 1u << (i % 32);
    for (hipStream_t& m : S->masked)
        if (hipExtStreamCreateWithCUMask(&m, 8, mask) != hipSuccess) { m = nullptr; (void)hipGetLastError(); return false; }
    S->maskedCus = cus;
    return true;
}

void side_set_release(SideSet* S)
{
    if (!S) return;
    std::lock_guard<std::mutex> lk(g_sideMu);
    S->busy = false;
}

// pending: an event on a side stream after which the blocks (2,1) and (2,2) of M have received their update from the
// PARENT's panel (see the deferred SYRK below); the main stream waits for it only when it first touches those blocks.
int cholinv_rec(dca_ctx* ctx, double* M, int ld, int n, int pivotBase, Arena& ws, int* dInfo, SideSet* side, int depth = 0, hipEvent_t pending = nullptr, int leafMaxOverride = 0)
{
    static const bool leaf128 = !(getenv("DCA_CHOLINV_LEAF128") && atoi(getenv("DCA_CHOLINV_LEAF128")) == 0);
    static const bool leafMfma = !(getenv("DCA_CHOLINV_LEAF_MFMA") && atoi(getenv("DCA_CHOLINV_LEAF_MFMA")) == 0);
    static const int leaf16Max = getenv("DCA_CHOLINV_LEAF16") ? atoi(getenv("DCA_CHOLINV_LEAF16")) : 128;      // 0: the four-column leaves
    const int l16 = leafMaxOverride > 0 ? leafMaxOverride : leaf16Max;
    if (n <= l16 && leaf16Max > 0) {
        switch (n) {
        case 64: hipLaunchKernelGGL(cholinv_leaf16_small_kernel<64>, dim3(1), dim3(512), leaf16_lds_bytes<64>(), ctx->stream, M, ld, pivotBase, dInfo); return DCA_OK;
        case 128: hipLaunchKernelGGL(cholinv_leaf16_small_kernel<128>, dim3(1), dim3(512), leaf16_lds_bytes<128>(), ctx->stream, M, ld, pivotBase, dInfo); return DCA_OK;
        case 192: hipLaunchKernelGGL(cholinv_leaf16_kernel<192>, dim3(1), dim3(512), leaf16_lds_bytes<192>(), ctx->stream, M, ld, pivotBase, dInfo); return DCA_OK;
        case 256: hipLaunchKernelGGL(cholinv_leaf16_kernel<256>, dim3(1), dim3(512), leaf16_lds_bytes<256>(), ctx->stream, M, ld, pivotBase, dInfo); return DCA_OK;
        default: break;
        }
    }
    if (n == 64) {
        if (leafMfma) hipLaunchKernelGGL(cholinv_leaf_mfma_kernel<64>, dim3(1), dim3(320), kLeafLds, ctx->stream, M, ld, pivotBase, dInfo);
        else hipLaunchKernelGGL(cholinv_leaf_kernel<64>, dim3(1), dim3(256), kLeafLds, ctx->stream, M, ld, pivotBase, dInfo);
        return DCA_OK;
    }
    if (n == 128 && leaf128) {
        if (leafMfma) hipLaunchKernelGGL(cholinv_leaf_mfma_kernel<128>, dim3(1), dim3(DCA_LEAF_WAVES128 * 64), kLeafLds, ctx->stream, M, ld, pivotBase, dInfo);
        else hipLaunchKernelGGL(cholinv_leaf_kernel<128>, dim3(1), dim3(1024), kLeafLds, ctx->stream, M, ld, pivotBase, dInfo);
        return DCA_OK;
    }
    // halves in multiples of 128 where possible, so that the recursion ends in 128-leaves (a 64-leaf only where n is an
    // odd multiple of 64)
    const int n1 = (leaf128 && n >= 256) ? (n / 128 / 2) * 128 : (n / 64 / 2) * 64, n2 = n - n1;
    double* M11 = M;
    double* M12 = M + n1;
    double* M21 = M + (size_t)n1 * ld;
    double* M22 = M + (size_t)n1 * ld + n1;
    DCA_TRY(cholinv_rec(ctx, M11, ld, n1, pivotBase, ws, dInfo, side, depth + 1, nullptr, leafMaxOverride));
    if (pending) HIP_TRY(hipStreamWaitEvent(ctx->stream, pending, 0));        // M21 and M22 are read / updated from here on
    const size_t mark = ws.top;
    double* L21 = ws.alloc((size_t)n2 * n1);
    double* Tt = ws.alloc((size_t)n1 * n2);
    if (!L21 || !Tt) { dca_set_error("cholinv workspace exhausted"); return DCA_ERR_NOMEM; }
    // L21 = A21 * X11^T : C[i][j] = sum_k A21[i][k] * X11[j][k],  X11 lower (k <= j)
    DCA_TRY(launch_gemm(ctx, GemmArgs{M21, ld, MASK_NONE, M11, ld, MASK_LOWER, L21, n1, nullptr, 0, n2, n1, n1, 1.0, 0.0, 0, WALK_COLUMNS_REVERSED}));
    // A22 -= L21 * L21^T (lower tiles); at the lower levels together with T^T in one launch
    const GemmArgs syrkArgs{L21, n1, MASK_NONE, L21, n1, MASK_NONE, M22, ld, nullptr, 0, n2, n2, n1, -1.0, 1.0, 1};
    const GemmArgs ttArgs{M11, ld, MASK_UPPER, L21, n1, MASK_NONE, Tt, n2, nullptr, 0, n1, n2, n1, 1.0, 0.0, 0};
    const bool paired = launch_gemm_small_pair(ctx, syrkArgs, ttArgs);
    // T^T[j][i] = sum_k X11^T[j][k] * L21[i][k];  X11^T rows are the mirrored upper part of M11 (k >= j).  It needs X11 and
    // L21 only, so at the upper levels it runs on a side stream NEXT TO the A22 subtree, whose chain of leaves and few-tile
    // products leaves the chip idle -- as launches of fewer workgroups than CUs (launch_gemm_banded), which is what makes the
    // overlap work: full-grid launches on a side stream (round 1) and CU-masked streams (round 2) had gained nothing.
    static const int sideMode = getenv("DCA_CHOLINV_SIDE") ? atoi(getenv("DCA_CHOLINV_SIDE")) : 1;
    // workgroups per background launch at depth 0, 1, 2 (0: that depth runs its product in line).  Measured at n = 10 048
    // (inverse, ms; 24.4 - 24.6 without): depth 0 alone 40 / 80 / 120 / 240 workgroups 32.3 / 26.1 / 24.6 / 23.7 (a slow
    // background product is waited for at the join); any background at depths 1 or 2 loses -- their subtrees are chains of
    // few-tile products that share CUs with it (0,40,10: 36.0, 0,0,10: 25.5 without stream priorities) -- so only the top
    // level overlaps: n = 8000: 14.7 -> 14.5, n = 6000: 7.6 -> 7.5, n = 4000: unchanged
    struct SideBudget { int v[kSideDepths]; };
    static const SideBudget sideBudgetInit = [] {            // parsed once, thread-safe (contexts may live on several host threads)
        SideBudget b{{240, 0, 0}};
        if (const char* e = getenv("DCA_CHOLINV_SIDE_WGS")) sscanf(e, "%d,%d,%d", &b.v[0], &b.v[1], &b.v[2]);
        return b;
    }();
    const int* sideBudget = sideBudgetInit.v;
    bool onSide = false;
    static const int sideMinN1 = getenv("DCA_CHOLINV_SIDE_MIN") ? atoi(getenv("DCA_CHOLINV_SIDE_MIN")) : 1024;
    // the fork comes BEFORE the SYRK: that product's 820 lower 128 x 128 tiles fill 512 slots 1.6 times (47 TF), and the
    // background bands take what its ragged second round leaves idle
    static const bool forkBeforeSyrk = !(getenv("DCA_CHOLINV_SIDE_FORK") && atoi(getenv("DCA_CHOLINV_SIDE_FORK")) == 0);
    const bool useSide = !paired && side && sideMode && depth < kSideDepths && n1 >= sideMinN1 && sideBudget[depth] > 0;
    // Deferred SYRK (round 4).  The A22 subtree starts with the recursion on A22's OWN first diagonal block (h x h): only that
    // quadrant of the update A22 -= L21 L21^T is needed at once.  The other three quarters of its flop -- the blocks (2,1)
    // and (2,2) of A22 -- go to the side stream as banded launches IN FRONT of T^T and run next to that first, chain-heavy
    // half of the subtree; the child waits for them (`pending`) where it first touches those blocks.
    // MEASURED (round 4, opt-in DCA_CHOLINV_DEFER_SYRK=1, NOT adopted): n = 10 048 23.5 - 23.8 ms against 23.5, n = 8000 14.8
    // against 14.5, n = 6000 7.8 against 8.0, n = 4000 3.9 against 3.8 -- the background bands take from the foreground's
    // mid-size products what they give to the leaf chain, as every other background product tried at this depth did.
    static const bool deferSyrk = getenv("DCA_CHOLINV_DEFER_SYRK") && atoi(getenv("DCA_CHOLINV_DEFER_SYRK")) == 1;
    const int h = (leaf128 && n2 >= 256) ? (n2 / 128 / 2) * 128 : (n2 / 64 / 2) * 64;       // the child's split of n2
    bool deferred = false;
    auto fork_side = [&]() -> int {
        HIP_TRY(hipEventRecord(side->fork[depth], ctx->stream));                  // L21 (and X11) are complete
        HIP_TRY(hipStreamWaitEvent(side->s[depth], side->fork[depth], 0));
        if (deferSyrk && !paired && n2 >= 2048 && h >= 128 && n2 - h >= 128) {
            const double* Lh = L21 + (size_t)h * n1;
            DCA_TRY(launch_gemm_banded(ctx, side->s[depth], GemmArgs{Lh, n1, MASK_NONE, L21, n1, MASK_NONE, M22 + (size_t)h * ld, ld, nullptr, 0, n2 - h, h, n1, -1.0, 1.0, 0}, sideBudget[depth]));
            DCA_TRY(launch_gemm_banded(ctx, side->s[depth], GemmArgs{Lh, n1, MASK_NONE, Lh, n1, MASK_NONE, M22 + (size_t)h * ld + h, ld, nullptr, 0, n2 - h, n2 - h, n1, -1.0, 1.0, 1}, sideBudget[depth]));
            HIP_TRY(hipEventRecord(side->mid[depth], side->s[depth]));
            deferred = true;
        }
        DCA_TRY(launch_gemm_banded(ctx, side->s[depth], ttArgs, sideBudget[depth]));
        HIP_TRY(hipEventRecord(side->join[depth], side->s[depth]));
        onSide = true;
        return DCA_OK;
    };
    // Once the background product is in flight, NO path may leave this frame without joining it: it writes Tt in the
    // arena, and the caller hands the side set back and may reuse or free the workspace as soon as this returns.  On
    // an error below, the side stream is drained on the host before the error is passed on.
    auto bail = [&](int rc) -> int {
        if (onSide) hipStreamSynchronize(side->s[depth]);
        ws.top = mark;
        return rc;
    };
    int rc = DCA_OK;
    if (useSide && forkBeforeSyrk && (rc = fork_side()) != DCA_OK) return bail(rc);
    if (!paired) {
        // with the deferred form only the first diagonal quadrant runs here
        const GemmArgs q00{L21, n1, MASK_NONE, L21, n1, MASK_NONE, M22, ld, nullptr, 0, h, h, n1, -1.0, 1.0, 1};
        if ((rc = launch_gemm(ctx, deferred ? q00 : syrkArgs)) != DCA_OK) return bail(rc);
    }
    if (useSide && !forkBeforeSyrk && (rc = fork_side()) != DCA_OK) return bail(rc);
    if ((rc = cholinv_rec(ctx, M22, ld, n2, pivotBase + n1, ws, dInfo, side, depth + 1, deferred ? side->mid[depth] : nullptr, leafMaxOverride)) != DCA_OK) return bail(rc);
    if (onSide) {
        if (hipStreamWaitEvent(ctx->stream, side->join[depth], 0) != hipSuccess) { dca_set_error("cholinv: join of the side stream failed"); return bail(DCA_ERR_HIP); }
    } else if (!paired) DCA_TRY(launch_gemm(ctx, ttArgs));
    // X21[i][j] = -sum_k X22[i][k] * T^T[j][k];  X22 lower (k <= i); mirrored into the (1,2) block
    DCA_TRY(launch_gemm(ctx, GemmArgs{M22, ld, MASK_LOWER, Tt, n2, MASK_NONE, M21, ld, M12, ld, n2, n1, n2, -1.0, 0.0, 0, WALK_ROWS_REVERSED}));
    ws.top = mark;
    return DCA_OK;
}

// ---------------------------------------------------------------------------------------------------------------
// Round 5: factorisation with look-ahead, triangular inverse as its own tree.
//
// cholinv_rec fuses factorisation and inversion in ONE serial walk: every leaf and every few-tile product of every level
// lies on the critical path, and while they run (one workgroup, a handful of workgroups) the chip idles -- 9 of the 23 ms
// at n = 10 048 for 17 % of the flop.  Here the walk is cut in two:
//
//  (a) Left-looking blocked Cholesky over panels of W columns.  The CHAIN (context stream) of panel j is
//        X_jj = inv(chol(D_j))                 the fused recursion above on the W x W diagonal block only
//        L_j  = A[below, j] X_jj^T             ONE product: the panel of the factor below the block (kept in `Lm`)
//        D_j+1 -= L_j[top] L_j[top]^T          the next diagonal block's update from this panel (a few tiles)
//      and everything else is BULK on a side stream, one panel ahead of the chain:
//        A[below j+1, j+1] -= L_j[below] L_j[top]^T        rows of the next panel under its diagonal block (K = W)
//        A[from j+2, j+2]  -= Lm[from j+2, 0 .. j] Lm[j+2, 0 .. j]^T     the panel after that, all earlier panels at once (deep K)
//      issued as launches of at most 240 workgroups of 128 x 128 tiles: MI355X places one such workgroup per CU before it
//      doubles up, so whole CUs stay free and the chain's single-workgroup leaves run next to the bulk at their own pace
//      (tools/experiments/overlap_bench.hip).  Events order the two streams: the chain's panel product waits for the rows
//      under its block, the diagonal update for the deep-K update of the same columns; the bulk waits for L_j.
//  (b) X = L^-1 above the diagonal blocks as a tree of products X21 = -X22 (L21 X11) over Lm: no leaf, no chain -- every
//      launch is a GEMM of at least 2 W rows.
//  (c) inv(A) = X^T X as before.
// Same arithmetic per product as the fused recursion; the sums are split at other places, so results differ from it
// in the last places (both are ~1e-15 from LAPACK relative to the norm).
struct EventPool {
    std::vector<hipEvent_t> ev;
    hipEvent_t get(size_t i)
    {
        while (ev.size() <= i) {
            hipEvent_t e = nullptr;
            if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) return nullptr;
            ev.push_back(e);
        }
        return ev[i];
    }
};
std::mutex g_poolMu;
std::vector<std::pair<SideSet*, EventPool*>> g_eventPools;     // one pool per side set, never destroyed (as the sets)
EventPool* event_pool_of(SideSet* S)
{
    std::lock_guard<std::mutex> lk(g_poolMu);
    for (auto& p : g_eventPools) if (p.first == S) return p.second;
    g_eventPools.emplace_back(S, new EventPool());
    return g_eventPools.back().second;
}

// A product on `stream` as 128 x 128 tiles in launches of at most maxWGs ACTIVE workgroups each (bands of tile rows of about
// equal tile count; with lowerOnly the tiles right of the diagonal are not launched).  WALK_ROWS only.
// the bulk kernel: 128 x 128 tiles; DCA_CHOLINV_BULK_KW=2 selects the eight-wave form (two k-tile groups per workgroup)
static int bulk_kw()
{
    // measured at n = 10 048: 1 -> 21.6 ms, 2 -> 22.3 ms.  The eight-wave form is faster per product but takes the whole register file of
    // its CU, so the chain's few-tile products find no room beside it (their average goes from 15 to 21 us, single ones wait 280 us)
    static const int v = (getenv("DCA_CHOLINV_BULK_KW") && atoi(getenv("DCA_CHOLINV_BULK_KW")) == 2) ? 2 : 1;
    return v;
}
void bulk_kernel_launch(hipStream_t stream, dim3 grid, const GemmArgs& g)
{
    if (bulk_kw() == 2) hipLaunchKernelGGL((gemm_nt_f64_dma_kernel<4, 4, 2>), grid, dim3(512), (size_t)8 * 128 * 16 * sizeof(double), stream, g);
    else hipLaunchKernelGGL(gemm_nt_f64_dma_kernel<4>, grid, dim3(256), (size_t)6 * 128 * 16 * sizeof(double), stream, g);
}

int launch_gemm_capped(hipStream_t stream, GemmArgs g, int maxWGs)
{
    const int gx = (g.N + 127) / 128, gy = (g.M + 127) / 128;
    auto active = [&](int r) { return g.lowerOnly ? std::min(gx, r + 1) : gx; };
    long long total = 0;
    for (int r = 0; r < gy; ++r) total += active(r);
    const int bands = (int)((total + maxWGs - 1) / maxWGs);
    const long long target = (total + bands - 1) / std::max(1, bands);
    for (int r = 0; r < gy;) {
        int r1 = r;
        long long wgs = 0;
        while (r1 < gy && (r1 == r || (wgs + active(r1) <= maxWGs && wgs < target))) wgs += active(r1++);
        g.row0 = r;
        bulk_kernel_launch(stream, dim3(active(r1 - 1), r1 - r), g);
        r = r1;
    }
    HIP_TRY(hipGetLastError());
    return DCA_OK;
}

// C[i][j] -= sum_z P[z][i][j] in slice order (the partial products of a split-k launch; P rows are N long)
__global__ __launch_bounds__(256)
void gemm_slices_reduce_kernel(double* __restrict__ C, int ldc, const double* __restrict__ P, size_t sliceStride, int slices, int M, int N)
{
    typedef double double2_t __attribute__((ext_vector_type(2)));
    const size_t pairs = (size_t)M * N / 2;
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < pairs; e += (size_t)gridDim.x * blockDim.x) {
        const size_t i = 2 * e / N, j = 2 * e % N;
        double2_t s = *reinterpret_cast<const double2_t*>(P + 2 * e);
        for (int z = 1; z < slices; ++z) s += *reinterpret_cast<const double2_t*>(P + (size_t)z * sliceStride + 2 * e);
        double2_t* cp = reinterpret_cast<double2_t*>(C + i * ldc + j);
        *cp -= s;
    }
}

// tile rows per launch of a split-k product: as many as fit under the cap, evened out over the launches
static int splitk_band_rows(int gx, int gy, int slices, int maxWGs)
{
    const int fit = std::max(1, maxWGs / (gx * slices));
    const int bands = (gy + fit - 1) / fit;
    return (gy + bands - 1) / bands;
}

// C -= A B^T (no masks, every tile) on `stream` with the k range cut into `slices` concurrent partial products: for the
// tall, narrow, deep-k updates of the look-ahead factorisation, whose few output tiles would otherwise each walk the whole
// k range on one CU while the others idle.  At most maxWGs workgroups per launch, as launch_gemm_capped.
int launch_gemm_splitk_capped(hipStream_t stream, const double* A, int lda, const double* B, int ldb, double* C, int ldc, int M, int N, int K,
                              int slices, double* P, int maxWGs)
{
    GemmArgs g{A, lda, MASK_NONE, B, ldb, MASK_NONE, P, N, nullptr, 0, M, N, K, 1.0, 0.0, 0};
    g.kSlices = slices;
    g.kChunk = ((K + slices - 1) / slices + 15) / 16 * 16;
    g.sliceStride = (size_t)M * N;
    const int gx = (N + 127) / 128, gy = (M + 127) / 128;
    const int rows = splitk_band_rows(gx, gy, slices, maxWGs);
    for (int r = 0; r < gy; r += rows) {
        g.row0 = r;
        bulk_kernel_launch(stream, dim3(gx, std::min(rows, gy - r), slices), g);
    }
    hipLaunchKernelGGL(gemm_slices_reduce_kernel, dim3(std::min<size_t>(2048, ((size_t)M * N / 2 + 255) / 256)), dim3(256), 0, stream, C, ldc, P, g.sliceStride, slices, M, N);
    HIP_TRY(hipGetLastError());
    return DCA_OK;
}

// C = alpha A B^T + beta C on `stream` in the stream-K form (gemm_nt_f64_streamk_kernel): W workgroups, at most `maxW`, as many as
// the scratch P (Pcap doubles) has slot pairs for and as there are iterations
int launch_gemm_streamk(hipStream_t stream, const double* A, int lda, const double* B, int ldb, int maskB, double* C, int ldc, int M, int N, int K,
                        double alpha, double beta, double* P, size_t Pcap, int maxW)
{
    if (K % 16 != 0) { dca_set_error("stream-K product: k range must be a multiple of 16"); return DCA_ERR_ARG; }
    StreamKArgs g{A, lda, B, ldb, maskB, C, ldc, M, N, K, alpha, beta, P, (N + 127) / 128, K / 16, 0, 0};
    const int gy = (M + 127) / 128, tiles = g.gx * gy;
    g.I = (long long)gy * g.KT;
    // groups: as many as the cap allows workgroups for, as the scratch has slots for, and no more than leave every piece
    // a few dozen k-tiles (a piece pays a pipeline start and a 128 KB store)
    g.W = (int)std::min<long long>({(long long)std::max(1, maxW / g.gx), (long long)(Pcap / ((size_t)2 * g.gx * 128 * 128)), std::max<long long>(1, g.I / 48)});
    if (g.W < 1) { dca_set_error("stream-K product: no scratch"); return DCA_ERR_NOMEM; }
    hipLaunchKernelGGL(gemm_nt_f64_streamk_kernel, dim3((unsigned)((g.W + 7) / 8 * 8 * g.gx)), dim3(256), (size_t)6 * 128 * 16 * sizeof(double), stream, g);
    hipLaunchKernelGGL(gemm_streamk_fixup_kernel, dim3(tiles), dim3(256), 0, stream, g);
    HIP_TRY(hipGetLastError());
    return DCA_OK;
}

struct BlockedCfg { int W, cap, overlap, minN, splitMinK, roundK, trsmSplit, streamK; };
static const BlockedCfg& blocked_cfg()
{
    static const BlockedCfg c = [] {
        // measured (tools/experiments/inv_sizes.sh; fused walk -> blocked, ms): n = 4032 3.75 -> 3.83, 5056 5.33 -> 5.24, 6016 8.00 -> 7.53,
        // 8000 14.36 -> 12.62, 10 048 23.48 -> 21.83; panels of 256 / 1024 columns lose 0.1 - 0.8 ms at every size
        // trsmSplit (DCA_CHOLINV_TRSM_SPLIT=1; round 5, measured, NOT adopted): only the next diagonal block's rows of a panel of
        // the factor on the chain, the rows below on the bulk stream -- 22.7 against 21.7 ms at n = 10 048 (13.0 / 12.6 at 8000,
        // 7.50 / 7.58 at 6016): the bulk stream is the longer of the two, and every cap from 248 to 2000 workgroups gives
        // 21.7 - 22.9 ms (160: 24.4), with the eight-wave bulk kernel 21.9 - 23.7 (profiles/r05_inverse_sweeps.txt)
        // streamK (DCA_CHOLINV_STREAMK=1; round 5, measured, NOT adopted): the bulk products in the stream-K form above -- 24.7 - 26.5 ms
        // against 21.6: equal k-tile counts per workgroup, but the pieces stand at different k of the operands, so the
        // 512 x K operand that the one-workgroup-per-tile form keeps in the L2 (all its workgroups walk k together) comes
        // from the fabric for every piece: 58 % of the matrix-core rate per CU against 91 % (tools/experiments/streamk_bench.hip,
        // profiles/r05_streamk_bench.txt: 40 TF on 256 CUs against 42 on 172)
        BlockedCfg v{512, 248, 1, 5000, 512, 300, 0, 0};
        if (const char* e = getenv("DCA_CHOLINV_PANEL")) v.W = std::max(128, atoi(e) / 128 * 128);   // 0 / unparsable -> 128; DCA_CHOLINV_BLOCKED=0 selects the fused walk
        if (const char* e = getenv("DCA_CHOLINV_SIDE_CAP")) v.cap = std::max(1, atoi(e));
        if (const char* e = getenv("DCA_CHOLINV_OVERLAP")) v.overlap = atoi(e);
        if (const char* e = getenv("DCA_CHOLINV_BLOCKED_MIN")) v.minN = atoi(e);
        if (const char* e = getenv("DCA_CHOLINV_ROUNDK")) v.roundK = atoi(e);
        if (const char* e = getenv("DCA_CHOLINV_TRSM_SPLIT")) v.trsmSplit = atoi(e);
        if (const char* e = getenv("DCA_CHOLINV_STREAMK")) v.streamK = atoi(e);
        if (const char* e = getenv("DCA_CHOLINV_SPLITK_MIN")) v.splitMinK = atoi(e);       // k per slice at least this (0: never split)
        if (const char* e = getenv("DCA_CHOLINV_BLOCKED")) if (atoi(e) == 0) v.minN = INT_MAX;
        return v;
    }();
    return c;
}

// X = L^-1 above the diagonal blocks b[lo] .. b[hi]: X21 = -X22 (L21 X11) per node, children first
int trtri_tree(dca_ctx* ctx, double* A, const double* Lm, int ld, const std::vector<int>& b, int lo, int hi, Arena& ws)
{
    if (hi - lo <= 1) return DCA_OK;
    int mid = lo + 1;                                            // the split nearest to half of the columns
    for (int k = lo + 1; k < hi; ++k)
        if (std::abs(2 * b[k] - b[lo] - b[hi]) < std::abs(2 * b[mid] - b[lo] - b[hi])) mid = k;
    DCA_TRY(trtri_tree(ctx, A, Lm, ld, b, lo, mid, ws));
    DCA_TRY(trtri_tree(ctx, A, Lm, ld, b, mid, hi, ws));
    const int n1 = b[mid] - b[lo], n2 = b[hi] - b[mid];
    double* M11 = A + (size_t)b[lo] * ld + b[lo];
    double* M12 = A + (size_t)b[lo] * ld + b[mid];
    double* M21 = A + (size_t)b[mid] * ld + b[lo];
    double* M22 = A + (size_t)b[mid] * ld + b[mid];
    const double* L21 = Lm + (size_t)b[mid] * ld + b[lo];
    const size_t mark = ws.top;
    double* Tt = ws.alloc((size_t)n1 * n2);
    if (!Tt) { dca_set_error("cholinv workspace exhausted"); return DCA_ERR_NOMEM; }
    // T^T[j][i] = sum_k X11^T[j][k] L21[i][k]  (X11^T rows: the mirrored upper part, k >= j), then X21[i][j] = -sum_k X22[i][k] T^T[j][k]
    DCA_TRY(launch_gemm(ctx, GemmArgs{M11, ld, MASK_UPPER, L21, ld, MASK_NONE, Tt, n2, nullptr, 0, n1, n2, n1, 1.0, 0.0, 0}));
    DCA_TRY(launch_gemm(ctx, GemmArgs{M22, ld, MASK_LOWER, Tt, n2, MASK_NONE, M21, ld, M12, ld, n2, n1, n2, -1.0, 0.0, 0, WALK_ROWS_REVERSED}));
    ws.top = mark;
    return DCA_OK;
}

// DCA_CHOLINV_TRACE=1 (measurement aid): timing events on both streams of the blocked inverse -- the kernel trace of the
// profiler serialises the two queues, these do not.  Printed to stderr when the inverse has finished.
struct StepTrace {
    bool on = false;
    std::vector<std::pair<std::string, hipEvent_t>> marks;
    ~StepTrace() { for (auto& m : marks) hipEventDestroy(m.second); }      // an early return does not leak the events
    void mark(hipStream_t st, const char* what, int j)
    {
        if (!on) return;
        hipEvent_t e = nullptr;
        if (hipEventCreate(&e) != hipSuccess) return;
        hipEventRecord(e, st);
        marks.emplace_back(std::string(what) + " " + std::to_string(j), e);
    }
    void dump()
    {
        if (!on || marks.empty()) return;
        hipDeviceSynchronize();
        for (size_t i = 0; i < marks.size(); ++i) {
            float ms = 0.f;
            hipEventElapsedTime(&ms, marks[0].second, marks[i].second);
            fprintf(stderr, "cholinv trace %9.1f us  %s\n", ms * 1e3, marks[i].first.c_str());
        }
        for (auto& m : marks) hipEventDestroy(m.second);
        marks.clear();
    }
};

int cholinv_blocked(dca_ctx* ctx, double* A, int n, Arena& ws, double* Lm, int* dInfo, SideSet* side)
{
    const BlockedCfg& cfg = blocked_cfg();
    static const bool traceOn = getenv("DCA_CHOLINV_TRACE") && atoi(getenv("DCA_CHOLINV_TRACE")) != 0;
    StepTrace tr;
    tr.on = traceOn;
    const int ld = n, W = cfg.W;
    std::vector<int> b;
    for (int c = 0; c < n; c += W) b.push_back(c);
    b.push_back(n);
    const int nb = (int)b.size() - 1;
    EventPool* pool = (side && cfg.overlap) ? event_pool_of(side) : nullptr;
    const bool twoStreams = pool != nullptr;
    hipStream_t bulk = twoStreams ? side->s[0] : ctx->stream;
    // events: 3 j trsm (L_j complete), 3 j + 1 rows (rows of panel j under its block complete), 3 j + 2 deep (deep-K update of panel j complete)
    auto ev = [&](int kind, int j) { return pool->get((size_t)3 * j + kind); };
    if (twoStreams && !ev(2, nb)) { dca_set_error("cholinv: event creation failed"); return DCA_ERR_HIP; }
    // partial products of the split-k updates: a fixed piece at the top of the arena (the chain's diagonal blocks use its bottom)
    const size_t sidePartialsCap = ws.cap / 2;
    double* sidePartials = ws.base + (ws.cap - sidePartialsCap);
    Arena chainWs{ws.base, ws.cap - sidePartialsCap};
    int rc = DCA_OK;
    bool bulkInFlight = false;
    auto bail = [&](int r) -> int {
        if (twoStreams && bulkInFlight) hipStreamSynchronize(bulk);     // the bulk writes A and reads Lm: drain it before the caller may free them
        return r;
    };
    // a failing event call inside the loop must not leave this frame before the bulk stream is drained (bail)
#define BLK_HIP(expr) if ((expr) != hipSuccess) { dca_set_error("%s failed (%s:%d)", #expr, __FILE__, __LINE__); rc = DCA_ERR_HIP; break; }
    for (int j = 0; j < nb && rc == DCA_OK; ++j) {
        const int c = b[j], w = b[j + 1] - c, m = n - c - w;
        double* D = A + (size_t)c * ld + c;
        tr.mark(ctx->stream, "chain: block begins", j);
        if ((rc = cholinv_rec(ctx, D, ld, w, c, chainWs, dInfo, nullptr)) != DCA_OK) break;
        tr.mark(ctx->stream, "chain: block factored", j);
        if (m == 0) break;
        // ---- chain: the panel of the factor below the block
        if (twoStreams && j > 0) BLK_HIP(hipStreamWaitEvent(ctx->stream, ev(1, j), 0));
        double* Lj = Lm + (size_t)(c + w) * ld + c;
        const int w1 = b[j + 2] - b[j + 1];
        // the chain needs only the rows of the NEXT diagonal block from this panel of the factor; with two streams the rows
        // below them are the bulk's (cfg.trsmSplit)
        const int mTop = (twoStreams && cfg.trsmSplit && m - w1 > 0) ? w1 : m;
        if ((rc = launch_gemm(ctx, GemmArgs{A + (size_t)(c + w) * ld + c, ld, MASK_NONE, D, ld, MASK_LOWER, Lj, ld, nullptr, 0, mTop, w, w, 1.0, 0.0, 0, WALK_COLUMNS_REVERSED})) != DCA_OK) break;
        tr.mark(ctx->stream, "chain: panel of the factor done", j);
        if (twoStreams) {
            BLK_HIP(hipEventRecord(ev(0, j), ctx->stream));
            BLK_HIP(hipStreamWaitEvent(bulk, ev(0, j), 0));
        }
        tr.mark(bulk, "bulk: begins step", j);
        if (mTop < m) {
            if (cfg.streamK) rc = launch_gemm_streamk(bulk, A + (size_t)(c + w + mTop) * ld + c, ld, D, ld, MASK_LOWER, Lj + (size_t)mTop * ld, ld, m - mTop, w, w, 1.0, 0.0,
                                                      sidePartials, sidePartialsCap, cfg.cap);
            else rc = launch_gemm_capped(bulk, GemmArgs{A + (size_t)(c + w + mTop) * ld + c, ld, MASK_NONE, D, ld, MASK_LOWER, Lj + (size_t)mTop * ld, ld, nullptr, 0,
                                                        m - mTop, w, w, 1.0, 0.0, 0}, cfg.cap);
            if (rc != DCA_OK) break;
            bulkInFlight = true;
        }
        // ---- bulk: rows of panel j + 1 under its diagonal block, from L_j
        if (m - w1 > 0) {
            if (cfg.streamK) rc = launch_gemm_streamk(bulk, Lj + (size_t)w1 * ld, ld, Lj, ld, MASK_NONE, A + (size_t)(c + w + w1) * ld + c + w, ld, m - w1, w1, w, -1.0, 1.0,
                                                      sidePartials, sidePartialsCap, cfg.cap);
            else rc = launch_gemm_capped(bulk, GemmArgs{Lj + (size_t)w1 * ld, ld, MASK_NONE, Lj, ld, MASK_NONE, A + (size_t)(c + w + w1) * ld + c + w, ld, nullptr, 0,
                                                        m - w1, w1, w, -1.0, 1.0, 0}, cfg.cap);
            if (rc != DCA_OK) break;
            bulkInFlight = true;
            if (twoStreams) BLK_HIP(hipEventRecord(ev(1, j + 1), bulk));
            tr.mark(bulk, "bulk: rows of the next panel done", j);
            // ---- bulk: panel j + 2 from all the panels up to j at once
            const int c2 = b[j + 2], w2 = b[j + 3 <= nb ? j + 3 : nb] - c2;
            if (w2 > 0) {
                const double* Lrows = Lm + (size_t)c2 * ld;
                // few output tiles and a deep k range: split k so that the launch has about `cap` workgroups
                const int K = c + w;
                // a launch of T tiles in s slices takes ceil(T s / cap) rounds of K / s (+ the tile's fixed part, ~64 k) each
                int slices = 1;
                if (cfg.splitMinK > 0) {
                    const int maxS = std::min({K / cfg.splitMinK, 16, (int)(sidePartialsCap / ((size_t)(n - c2) * w2))});
                    const int gx2 = (w2 + 127) / 128, gy2 = (n - c2 + 127) / 128;
                    long long best = LLONG_MAX;
                    for (int sl = 1; sl <= maxS || sl == 1; ++sl) {
                        // launches x (k per slice + what a launch costs beside its k walk, in k units: ~40 us)
                        const int rows = splitk_band_rows(gx2, gy2, sl, cfg.cap);
                        const long long cost = (long long)((gy2 + rows - 1) / rows) * (K / sl + cfg.roundK);
                        if (cost < best) { best = cost; slices = sl; }
                    }
                }
                if (cfg.streamK) rc = launch_gemm_streamk(bulk, Lrows, ld, Lrows, ld, MASK_NONE, A + (size_t)c2 * ld + c2, ld, n - c2, w2, K, -1.0, 1.0, sidePartials, sidePartialsCap, cfg.cap);
                else if (slices > 1) rc = launch_gemm_splitk_capped(bulk, Lrows, ld, Lrows, ld, A + (size_t)c2 * ld + c2, ld, n - c2, w2, K, slices, sidePartials, cfg.cap);
                else rc = launch_gemm_capped(bulk, GemmArgs{Lrows, ld, MASK_NONE, Lrows, ld, MASK_NONE, A + (size_t)c2 * ld + c2, ld, nullptr, 0,
                                                            n - c2, w2, K, -1.0, 1.0, 1}, cfg.cap);
                if (rc != DCA_OK) break;
                if (twoStreams) BLK_HIP(hipEventRecord(ev(2, j + 2), bulk));
                tr.mark(bulk, "bulk: deep update of panel j + 2 done", j);
            }
        }
        // ---- chain: the next diagonal block from L_j (after the deep-K update of the same block)
        if (twoStreams && j >= 1) BLK_HIP(hipStreamWaitEvent(ctx->stream, ev(2, j + 1), 0));
        if ((rc = launch_gemm(ctx, GemmArgs{Lj, ld, MASK_NONE, Lj, ld, MASK_NONE, A + (size_t)(c + w) * ld + c + w, ld, nullptr, 0, w1, w1, w, -1.0, 1.0, 1})) != DCA_OK) break;
    }
    if (rc != DCA_OK) return bail(rc);
    if (twoStreams && bulkInFlight) {
        if (hipEventRecord(ev(0, nb), bulk) != hipSuccess || hipStreamWaitEvent(ctx->stream, ev(0, nb), 0) != hipSuccess) {
            dca_set_error("cholinv: join of the bulk stream failed");
            return bail(DCA_ERR_HIP);
        }
    }
#undef BLK_HIP
    tr.mark(ctx->stream, "factorisation done", nb);
    rc = trtri_tree(ctx, A, Lm, ld, b, 0, nb, ws);
    tr.mark(ctx->stream, "triangular inverse done", nb);
    tr.dump();
    return rc;
}


// ---------------------------------------------------------------------------------------------------------------
// Round 6: the inverse as a symmetric BLOCK SWEEP (block Gauss-Jordan on an SPD matrix).
//
// The three-phase form above (factorisation, triangular inverse, X^T X) is three serial phases of which the first and
// the second run out of parallel work at their ends (a deep update of the look-ahead factorisation is 100 - 250 tiles)
// -- 33 / 53 / 58 TF at n = 10 048 where the GEMM alone sustains 69.  The sweep does the same n^3 flop in ONE phase whose
// every step is the same rank-w update of the WHOLE matrix.  With M symmetric (both halves stored), J the next w pivot
// columns and P = inv(M_JJ):
//     W      = M[:, J] P                       n x w x w
//     M[i,j] -= W[i, :] M[j, J]^T              every tile with i, j outside J  (lower tiles, mirrored: (n - w)^2 w flop)
//     M[:, J] = W,  M[J, :] = W^T,  M[J, J] = -P
// and after the last panel M = -inv(A) -- the couplings themselves.  M_JJ at the time its panel is swept is the Schur
// complement of the panels before it, so the pivots met are EXACTLY the pivots of the Cholesky factorisation of A: P comes
// from the same leaves (X = inv(chol(M_JJ)) by the fused recursion on the w x w block, P = X^T X) and a matrix that is not
// positive definite is reported with the same pivot index as before.
// Two streams.  The CHAIN (context stream) owns the diagonal: from P_p it forms the next pivot block itself,
//     S' = M[J', J'] - (M[J', J] P_p) M[J', J]^T   (three few-tile products), then X', P' ...
// and never waits for the bulk of its own step.  The BULK stream does W, then the tiles that the chain of the NEXT steps
// reads (column panel J' and the diagonal block after it: `prio`), then the rest, as PERSISTENT launches of at most `cap`
// workgroups of 128 x 128 tiles (one per CU, whole CUs stay free for the chain's kernels), every workgroup walking the
// tile list with stride `cap`; the list is ordered in bands of four tile rows so that the 31 workgroups that share an
// XCD's L2 work on a 4 x 8 block of tiles at a time (12 operand panels instead of 62).
enum { SWEEP_REST = 0, SWEEP_PRIO = 1 };
struct SweepArgs {
    const double* W; int ldw;      // n x w: the scaled panel
    const double* Q; int ldq;      // n x w: the pivot panel's columns as they were before the step (compact copy, all rows)
    double* M; int ld;             // n x n: the lower triangle is kept up to date
    int n, c, w;                   // pivot columns [c, c + w)
    int nt;                        // tile rows of M (128 rows each, the last one may be short)
    int skip0, skipN;              // tile rows left out: the pivot panel and the panel after it
    int mode;
    int pr0, prN;                  // SWEEP_PRIO: tile rows of the next panel
    int dg0, dgN;                  // diagonal block of the panel after the next: its lower tiles belong to PRIO, not to REST
    int nTiles;                    // length of the tile list (entries that decode to no tile are passed over)
    int* ctr;                      // 8 zeroed counters: the list is cut into 8 chunks, chunk x is handed out to the workgroups of XCD x first
    const unsigned* reserved = nullptr;   // bit per CU (cu_key()): a workgroup that finds itself on a reserved CU returns at once
};

// the CU this wave runs on: XCC_ID (hwreg 20) and SE_ID / SH_ID / CU_ID of HW_ID (hwreg 4): bits 15:13 / 12 / 11:8
__device__ __forceinline__ unsigned cu_key()
{
    const unsigned hw = __builtin_amdgcn_s_getreg(4 | (0 << 6) | (31 << 11));
    const unsigned xcc = __builtin_amdgcn_s_getreg(20 | (0 << 6) | (3 << 11));
    return (xcc & 7) * 128 + ((hw >> 8) & 0xff) % 128;
}

// tile list entry t -> tile (ti, tj), tj <= ti
__device__ __forceinline__ bool sweep_tile_of(const SweepArgs& g, int t, int& ti, int& tj)
{
    const int nR = g.nt - g.skipN;
    if (g.mode == SWEEP_REST) {
        // lower triangle over the nR remaining tile rows in bands of 4 rows, column by column inside a band; band k starts at 8 k^2 + 2 k
        int k = (int)((sqrtf(4.f + 32.f * (float)t) - 2.f) * (1.f / 16.f));
        while (8 * (k + 1) * (k + 1) + 2 * (k + 1) <= t) ++k;
        while (8 * k * k + 2 * k > t) --k;
        const int u = t - (8 * k * k + 2 * k);
        int a, b;
        if (u < 16 * k) { b = u >> 2; a = 4 * k + (u & 3); }
        else {
            const int v = u - 16 * k;                          // the band's own 4 x 4 triangle, column by column
            const int col = v < 4 ? 0 : v < 7 ? 1 : v < 9 ? 2 : 3;
            const int row = v < 4 ? v : v < 7 ? v - 3 : v < 9 ? v - 5 : 3;
            a = 4 * k + row; b = 4 * k + col;
        }
        if (a >= nR) return false;
        ti = a < g.skip0 ? a : a + g.skipN;
        tj = b < g.skip0 ? b : b + g.skipN;
        if (g.dgN > 0 && tj >= g.dg0 && ti < g.dg0 + g.dgN) return false;      // tj <= ti: both inside the block
        return true;
    }
    const int rect = g.prN * nR;
    if (t < rect) {
        const int cc = g.pr0 + t % g.prN;
        int r = t / g.prN;
        r = r < g.skip0 ? r : r + g.skipN;
        ti = max(r, cc); tj = min(r, cc);
        return true;
    }
    const int u = t - rect;
    int a = 0;
    while ((a + 1) * (a + 2) / 2 <= u) ++a;
    ti = g.dg0 + a; tj = g.dg0 + u - a * (a + 1) / 2;
    return true;
}

// 128 x 128 tiles, operands by LDS-DMA as in gemm_nt_f64_dma_kernel<4>; no triangular operand.  The accumulators START as
// -M[tile] (fetched while the first operand tiles are on their way) and the tile is stored as -acc: the read half of the
// read-modify-write costs no registers and no pass of its own.
// NST operand stages in LDS (32 KB each), NST - 1 of them in flight: every k-tile of 16 is a NEW 128-byte line of every
// operand row, so each step waits for a fetch that nobody has made before (the workgroups of an XCD walk k together, the
// first to ask misses the L2), and one workgroup per CU has nothing else to run meanwhile: with two stages the step took
// 2.5 us against 0.85 us of matrix-core time (round 6, first measurement: 82 us per K = 512 tile).
#ifndef DCA_SWEEP_ABLATE
#define DCA_SWEEP_ABLATE 0        // tools/experiments/sweep_bench.hip, timing only: 1 no load of the tile, 4 no store
#endif
template <int NST>
__global__ __launch_bounds__(256, 2)
void sweep_update_kernel(SweepArgs g)
{
    constexpr int BK = 16, TW = 4, BM = 128, BN = 128;
    constexpr int PW = BM / 8 / 4;
    constexpr int OP = BM * BK * (int)sizeof(double);
    typedef double double2_t __attribute__((ext_vector_type(2)));
    extern __shared__ __attribute__((aligned(16))) unsigned char dca_gemm_smem[];   // [NST][A | B]
    unsigned char* const smem = dca_gemm_smem;
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int fr = lane & 15, fg = lane >> 4, swz = (fr >> 1) & 7;
    const int nk = g.w / BK;
    __shared__ int s_tile;
    if (g.reserved) {
        const unsigned key = cu_key();
        if ((g.reserved[key >> 5] >> (key & 31)) & 1) return;
    }
    int q = (int)blockIdx.x % 8, tries = 0;                         // thread 0: the chunk it draws from, chunks found empty

    for (;;) {
        if (tid == 0) {
            int t = -1;
            while (tries < 8) {
                const int lo = (int)((long long)g.nTiles * q / 8), hi = (int)((long long)g.nTiles * (q + 1) / 8);
                const int v = lo + atomicAdd(&g.ctr[q], 1);
                if (v < hi) { t = v; break; }
                q = (q + 1) & 7; ++tries;                           // own chunk done: help the next XCD's
            }
            s_tile = t;
        }
        __syncthreads();
        const int t = s_tile;
        __syncthreads();
        if (t < 0) break;
        int ti, tj;
        if (!sweep_tile_of(g, t, ti, tj)) continue;                 // the same for the whole workgroup
        // the tile itself first (oldest in the memory queue: landed when the first operand stage has)
        double4_t acc[TW][TW];
        const int iBase = ti * BM + wm * 64 + (lane >> 4), jBase = tj * BN + wn * 64 + (lane & 15);
#pragma unroll
        for (int m = 0; m < TW; ++m)
#pragma unroll
            for (int n = 0; n < TW; ++n)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int i = min(iBase + m * 16 + 4 * r, g.n - 1), j = min(jBase + n * 16, g.n - 1);
                    acc[m][n][r] = (DCA_SWEEP_ABLATE & 1) ? 0.0 : -g.M[(size_t)i * g.ld + j];
                }
        const double* srcA[PW];
        const double* srcB[PW];
#pragma unroll
        for (int i = 0; i < PW; ++i) {
            const int R = 8 * (PW * wave + i) + (lane >> 3);
            const int piece = (lane & 7) ^ ((R >> 1) & 7);
            srcA[i] = g.W + (size_t)min(ti * BM + R, g.n - 1) * g.ldw + 2 * piece;
            srcB[i] = g.Q + (size_t)min(tj * BN + R, g.n - 1) * g.ldq + 2 * piece;
        }
        auto issue = [&](int st, int k0) {
#pragma unroll
            for (int i = 0; i < PW; ++i)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(srcA[i] + k0),
                                                 (__attribute__((address_space(3))) void*)(smem + st * 2 * OP + (PW * wave + i) * 1024), 16, 0, 0);
#pragma unroll
            for (int i = 0; i < PW; ++i)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(srcB[i] + k0),
                                                 (__attribute__((address_space(3))) void*)(smem + st * 2 * OP + OP + (PW * wave + i) * 1024), 16, 0, 0);
        };
        // wait until at most `tiles` operand stages of this wave are still in flight (2 PW loads each)
        auto wait_stages = [&](int tiles) {
            if (tiles >= 3) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(6 * PW) : "memory");
            else if (tiles == 2) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(4 * PW) : "memory");
            else if (tiles == 1) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(2 * PW) : "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        };
        auto mma_tile = [&](int st) {
            const unsigned char* as = smem + st * 2 * OP;
            const unsigned char* bs = as + OP;
#pragma unroll
            for (int kk = 0; kk < BK / 8; ++kk) {
                const int slot = (4 * kk + fg) ^ swz;
                double2_t a[TW], b[TW];
#pragma unroll
                for (int m = 0; m < TW; ++m) a[m] = *reinterpret_cast<const double2_t*>(as + (wm * 64 + 16 * m + fr) * 128 + slot * 16);
#pragma unroll
                for (int m = 0; m < TW; ++m) b[m] = *reinterpret_cast<const double2_t*>(bs + (wn * 64 + 16 * m + fr) * 128 + slot * 16);
#pragma unroll
                for (int h = 0; h < 2; ++h)
#pragma unroll
                    for (int m = 0; m < TW; ++m)
#pragma unroll
                        for (int n = 0; n < TW; ++n) acc[m][n] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[m][h], b[n][h], acc[m][n], 0, 0, 0);
            }
        };
        const int ahead = min(NST - 1, nk);
        for (int s = 0; s < ahead; ++s) issue(s, s * BK);
        wait_stages(ahead - 1);
        __syncthreads();
        for (int kt = 0, st = 0; kt < nk; ++kt) {
            const int nxt = kt + NST - 1;                             // its stage is the one tile kt - 1 used
            if (nxt < nk) issue(st == 0 ? NST - 1 : st - 1, nxt * BK);
            mma_tile(st);
            wait_stages(min(NST - 2, nk - kt - 2));                   // tile kt + 1 has landed
            __syncthreads();
            st = st == NST - 1 ? 0 : st + 1;
        }
        const bool diag = ti == tj;
        if (!(DCA_SWEEP_ABLATE & 4)) {
#pragma unroll
            for (int m = 0; m < TW; ++m)
#pragma unroll
                for (int n = 0; n < TW; ++n)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int i = iBase + m * 16 + 4 * r, j = jBase + n * 16;
                        if (i >= g.n || j >= g.n) continue;
                        if (diag && j > i) continue;
                        const double v = -acc[m][n][r];
                        g.M[(size_t)i * g.ld + j] = v;
                    }
        } else if (acc[0][0][0] == 1.2345e-300) g.M[0] = 0.0;
    }
}

// The swept panel in the lower triangle: M[i][c .. c + w) <- W[i] for the rows below the pivot block, M[c .. c + w)[j] <- W[j]^T
// for the columns left of it, the pivot block itself <- -P (32 x 32 pieces)
__global__ __launch_bounds__(256)
void sweep_finalize_kernel(double* __restrict__ M, int ld, int c, int w, const double* __restrict__ W, int ldw, const double* __restrict__ P, int ldp)
{
    __shared__ double tile[32][33];
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    const int k0 = blockIdx.x * 32, r0 = blockIdx.y * 32;
    if (r0 >= c + w) {
#pragma unroll
        for (int s = 0; s < 4; ++s) M[(size_t)(r0 + ty + 8 * s) * ld + c + k0 + tx] = W[(size_t)(r0 + ty + 8 * s) * ldw + k0 + tx];
        return;
    }
    if (r0 >= c) {
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const int r = r0 + ty + 8 * s;
            M[(size_t)r * ld + c + k0 + tx] = -P[(size_t)(r - c) * ldp + k0 + tx];
        }
        return;
    }
#pragma unroll
    for (int s = 0; s < 4; ++s) tile[ty + 8 * s][tx] = W[(size_t)(r0 + ty + 8 * s) * ldw + k0 + tx];
    __syncthreads();
#pragma unroll
    for (int s = 0; s < 4; ++s) M[(size_t)(c + k0 + ty + 8 * s) * ld + r0 + tx] = tile[tx][ty + 8 * s];
}

// Q[i][k] = M[i][c + k] as the symmetric matrix has it, read from the LOWER triangle only: rows below the panel's diagonal
// block as they stand, rows above it from the row panel left of the block (transposed through LDS).  The rows of the block
// itself are copied as they stand (nothing reads them).
__global__ __launch_bounds__(256)
void gather_panel_kernel(const double* __restrict__ M, int ld, int c, int w, double* __restrict__ Q, int ldq)
{
    __shared__ double tile[32][33];
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    const int k0 = blockIdx.x * 32, r0 = blockIdx.y * 32;
    if (r0 >= c) {
#pragma unroll
        for (int s = 0; s < 4; ++s) Q[(size_t)(r0 + ty + 8 * s) * ldq + k0 + tx] = M[(size_t)(r0 + ty + 8 * s) * ld + c + k0 + tx];
        return;
    }
#pragma unroll
    for (int s = 0; s < 4; ++s) tile[ty + 8 * s][tx] = M[(size_t)(c + k0 + ty + 8 * s) * ld + r0 + tx];
    __syncthreads();
#pragma unroll
    for (int s = 0; s < 4; ++s) Q[(size_t)(r0 + ty + 8 * s) * ldq + k0 + tx] = tile[tx][ty + 8 * s];
}

// upper triangle <- transposed lower triangle, 32 x 32 pieces (the sweep's last pass: the result is bit-symmetric)
__global__ __launch_bounds__(256)
void symmetrize_kernel(double* __restrict__ M, int ld, int nb)
{
    __shared__ double tile[32][33];
    // piece (bi, bj), bj < bi, from the linear index of the strict lower triangle of pieces
    const int t = blockIdx.x;
    int bi = (int)((sqrtf(8.f * (float)t + 1.f) + 1.f) * 0.5f);
    while (bi * (bi - 1) / 2 > t) --bi;
    while ((bi + 1) * bi / 2 <= t) ++bi;
    const int bj = t - bi * (bi - 1) / 2;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
#pragma unroll
    for (int s = 0; s < 4; ++s) tile[ty + 8 * s][tx] = M[(size_t)(32 * bi + ty + 8 * s) * ld + 32 * bj + tx];
    __syncthreads();
#pragma unroll
    for (int s = 0; s < 4; ++s) M[(size_t)(32 * bj + ty + 8 * s) * ld + 32 * bi + tx] = tile[tx][ty + 8 * s];
}
// ... and inside the diagonal pieces
__global__ __launch_bounds__(256)
void symmetrize_diag_kernel(double* __restrict__ M, int ld)
{
    const int b = blockIdx.x;
    for (int e = threadIdx.x; e < 32 * 32; e += 256) {
        const int r = e >> 5, cc = e & 31;
        if (cc > r) M[(size_t)(32 * b + r) * ld + 32 * b + cc] = M[(size_t)(32 * b + cc) * ld + 32 * b + r];
    }
}

__global__ __launch_bounds__(256)
void copy_block_kernel(double* __restrict__ dst, int ldd, const double* __restrict__ src, int lds, int rows, int cols)
{
    typedef double double2_t __attribute__((ext_vector_type(2)));
    const int perRow = cols / 2;
    for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < rows * perRow; e += gridDim.x * blockDim.x) {
        const int r = e / perRow, k = 2 * (e % perRow);
        *reinterpret_cast<double2_t*>(dst + (size_t)r * ldd + k) = *reinterpret_cast<const double2_t*>(src + (size_t)r * lds + k);
    }
}

__global__ __launch_bounds__(256)
void scale_matrix_kernel(double* __restrict__ M, size_t count, double f)
{
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < count; e += (size_t)gridDim.x * blockDim.x) M[e] *= f;
}

// HIP maps streams onto a few hardware queues (GPU_MAX_HW_QUEUES, 4 by default) and two streams on one queue run one after the
// other.  The sweep needs three streams that really overlap (measured at n = 10 048 inside the mfDCA chain, where the
// process holds a dozen streams: 26.0 ms with the sweep's streams sharing queues, 19.4 ms with them apart), so it PROBES:
// a kernel on stream a waits (at most 300 us) for a flag that a kernel enqueued afterwards on stream b sets.
__global__ void probe_wait_kernel(int* flag, int* result, long long ticks)
{
    const long long t0 = wall_clock64();
    int seen = 0;
    while (wall_clock64() - t0 < ticks) {
        if (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != 0) { seen = 1; break; }
        __builtin_amdgcn_s_sleep(8);
    }
    __hip_atomic_store(result, seen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
__global__ void probe_set_kernel(int* flag) { __hip_atomic_store(flag, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }

static bool streams_overlap(SideSet* S, hipStream_t a, hipStream_t b)
{
    if (a == b) return false;
    // flag and result live in pinned host memory the kernels reach directly: no memset / copy calls around the two launches
    if (!S->probe && hipHostMalloc(reinterpret_cast<void**>(&S->probe), 2 * sizeof(int), hipHostMallocMapped) != hipSuccess) { (void)hipGetLastError(); S->probe = nullptr; return true; }
    volatile int* h = S->probe;
    h[0] = 0; h[1] = 0;
    hipLaunchKernelGGL(probe_wait_kernel, dim3(1), dim3(1), 0, a, S->probe, S->probe + 1, 30000LL);      // wall clock: 100 MHz
    hipLaunchKernelGGL(probe_set_kernel, dim3(1), dim3(1), 0, b, S->probe);
    if (hipStreamSynchronize(a) != hipSuccess || hipStreamSynchronize(b) != hipSuccess) return true;
    return h[1] != 0;
}
// two streams of the set that overlap with `chain` and with each other (cached per chain stream); the set's first two if none are found
static void sweep_streams(SideSet* S, hipStream_t chain, hipStream_t& rest, hipStream_t& side)
{
    static const bool probeOn = !(getenv("DCA_SWEEP_PROBE") && atoi(getenv("DCA_SWEEP_PROBE")) == 0);
    rest = S->s[0]; side = S->s[1];
    if (!probeOn) return;
    if (S->selFor == chain && S->selRest) { rest = S->selRest; side = S->selSide; return; }
    if (hipStreamSynchronize(chain) != hipSuccess) return;              // the probe's kernel must be the stream's next
    std::vector<hipStream_t> cand(S->s, S->s + kSideDepths);
    cand.insert(cand.end(), S->extra.begin(), S->extra.end());
    std::vector<hipStream_t> good;                                      // overlap with the chain
    for (size_t i = 0; good.size() < 2 && i < 16; ++i) {
        if (i >= cand.size()) {
            hipStream_t ns = nullptr;
            if (hipStreamCreateWithFlags(&ns, hipStreamNonBlocking) != hipSuccess) { (void)hipGetLastError(); break; }
            S->extra.push_back(ns);
            cand.push_back(ns);
        }
        if (!streams_overlap(S, chain, cand[i])) continue;            // (two streams on one queue fail in whichever order they are asked)
        if (good.size() == 1 && !streams_overlap(S, good[0], cand[i])) continue;
        good.push_back(cand[i]);
    }
    if (good.size() == 2) { rest = good[0]; side = good[1]; }
    S->selFor = chain; S->selRest = rest; S->selSide = side;
}

// CUs of the chain's own (round 6).  A CU-masked stream dispatches slowly (section 4 of DESIGN.md); instead the update
// kernel looks up the CU it finds itself on and returns at once on a reserved one.  The dispatcher hands a workgroup to a
// shader engine before it looks for room (measured: with fewer reserved CUs than engines a one-workgroup kernel waits 100 - 140 us
// beside a full update launch, with one per engine 42 us = alone), so the unit is one CU per engine = 4 per XCD = 32.
// Which CU ids exist is probed once per set: 4096 workgroups of 64 KB LDS that stay 20 us each visit every CU.
__global__ __launch_bounds__(256) void cu_probe_kernel(unsigned* present)
{
    extern __shared__ unsigned char dca_gemm_smem[];
    if (threadIdx.x == 0) {
        const unsigned key = cu_key();
        atomicOr(&present[key >> 5], 1u << (key & 31));
        const long long t0 = wall_clock64();
        while (wall_clock64() - t0 < 2000) {}
        if (dca_gemm_smem[0] == 77 && t0 == 1) present[0] = 0;
    }
}
static const unsigned* sweep_reserved_cus(SideSet* S, hipStream_t st, int perXcd)
{
    if (perXcd <= 0) return nullptr;
    if (S->reserved && S->reservedPerXcd == perXcd) return S->reserved;
    if (!S->reserved && hipMalloc(reinterpret_cast<void**>(&S->reserved), 32 * sizeof(unsigned)) != hipSuccess) { (void)hipGetLastError(); S->reserved = nullptr; return nullptr; }
    unsigned present[32] = {}, res[32] = {};
    if (hipMemsetAsync(S->reserved, 0, sizeof present, st) != hipSuccess) return nullptr;
    hipLaunchKernelGGL(cu_probe_kernel, dim3(4096), dim3(256), 65536, st, S->reserved);
    if (hipMemcpyAsync(present, S->reserved, sizeof present, hipMemcpyDeviceToHost, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
    // per XCD: engine by engine the lowest CU not yet taken, until perXcd are (cu_key(): 32 ids per engine, four engines per XCD)
    for (int x = 0; x < 8; ++x) {
        int taken = 0;
        for (int round = 0; round < 32 && taken < perXcd; ++round)
            for (int se = 0; se < 4 && taken < perXcd; ++se)
                for (int id = 0; id < 32; ++id) {
                    const unsigned key = x * 128 + se * 32 + id;
                    if (!((present[key >> 5] >> (key & 31)) & 1) || ((res[key >> 5] >> (key & 31)) & 1)) continue;
                    res[key >> 5] |= 1u << (key & 31); ++taken;
                    break;
                }
    }
    if (hipMemcpyAsync(S->reserved, res, sizeof res, hipMemcpyHostToDevice, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
    S->reservedPerXcd = perXcd;
    return S->reserved;
}

struct SweepCfg { int minN, wide, narrow, wideMinN, cap, stages, perCu, maskCus, prioCap, reserve, reserveMaxN, factorMaxN; };
void sweep_update_launch(hipStream_t stream, int G, int stages, int perCu, const SweepArgs& g)
{
    // dynamic LDS: what the stages need; with perCu == 1 never less than 96 KB, so that a CU takes ONE of these workgroups
    const size_t stage = (size_t)2 * 128 * 16 * sizeof(double);
    const size_t lds = perCu >= 2 ? stages * stage : std::max<size_t>(stages, 3) * stage;
    if (stages >= 4) hipLaunchKernelGGL(sweep_update_kernel<4>, dim3(G), dim3(256), lds, stream, g);
    else if (stages == 3) hipLaunchKernelGGL(sweep_update_kernel<3>, dim3(G), dim3(256), lds, stream, g);
    else hipLaunchKernelGGL(sweep_update_kernel<2>, dim3(G), dim3(256), lds, stream, g);
}
static const SweepCfg& sweep_cfg()
{
    static const SweepCfg c = [] {
        SweepCfg v{2560, 512, 256, 7000, 0, 2, 2, 0, 64, 0, 7000, 4500};      // measured (ms, sweep / three-phase): n = 2048 1.43 / 1.30, 3072 2.16 / 2.31, 4032 3.41 / 3.55, 6016 6.46 / 7.54, 8000 12.25 / 13.25, 10 048 20.6 / 22.4, 12 032 34.0 / 34.9
        if (const char* e = getenv("DCA_SWEEP_PRIO_CAP")) v.prioCap = std::max(8, atoi(e) / 8 * 8);
        if (const char* e = getenv("DCA_SWEEP_RESERVE")) v.reserve = std::max(0, std::min(16, atoi(e)));          // CUs per XCD left to the chain's kernels
        if (const char* e = getenv("DCA_SWEEP_RESERVE_MAX_N")) v.reserveMaxN = atoi(e);
        if (const char* e = getenv("DCA_SWEEP_FACTOR_MAX_N")) v.factorMaxN = atoi(e);       // below: next pivot block from X (F F^T), P on the side stream
        if (const char* e = getenv("DCA_SWEEP_MIN")) v.minN = atoi(e);
        if (const char* e = getenv("DCA_SWEEP")) if (atoi(e) == 0) v.minN = INT_MAX;
        if (const char* e = getenv("DCA_SWEEP_PANEL")) v.wide = v.narrow = std::max(128, atoi(e) / 128 * 128);
        if (const char* e = getenv("DCA_SWEEP_CAP")) v.cap = std::max(8, atoi(e) / 8 * 8);
        if (const char* e = getenv("DCA_SWEEP_STAGES")) v.stages = std::max(2, std::min(4, atoi(e)));
        if (const char* e = getenv("DCA_SWEEP_PER_CU")) v.perCu = atoi(e);
        if (const char* e = getenv("DCA_SWEEP_MASK")) v.maskCus = std::max(0, std::min(256, atoi(e) / 8 * 8));      // 0: plain streams
        return v;
    }();
    return c;
}

int sweep_kernels_prepare(int device)
{
    static std::mutex mu;
    static std::vector<int> done;
    std::lock_guard<std::mutex> lk(mu);
    for (int d : done) if (d == device) return DCA_OK;
    HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(sweep_update_kernel<2>), hipFuncAttributeMaxDynamicSharedMemorySize, 6 * 128 * 16 * 8));
    HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(sweep_update_kernel<3>), hipFuncAttributeMaxDynamicSharedMemorySize, 6 * 128 * 16 * 8));
    HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(sweep_update_kernel<4>), hipFuncAttributeMaxDynamicSharedMemorySize, 8 * 128 * 16 * 8));
    HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(cu_probe_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 65536));
    done.push_back(device);
    return DCA_OK;
}

int cholinv_sweep(dca_ctx* ctx, double* A, int n, double* work, int* dInfo, SideSet* side)
{
    const SweepCfg& cfg = sweep_cfg();
    DCA_TRY(sweep_kernels_prepare(ctx->device));
    struct Bounds { Bounds(int n) { if (n >= 6000) { tl_small32_max = 16; tl_deep_max_tiles = 16; } } ~Bounds() { tl_small32_max = -1; tl_deep_max_tiles = -1; } } bounds(n);
    static const bool traceOn = getenv("DCA_CHOLINV_TRACE") && atoi(getenv("DCA_CHOLINV_TRACE")) != 0;
    StepTrace tr;
    tr.on = traceOn;
    const int ld = n;
    const int B = n >= cfg.wideMinN ? cfg.wide : cfg.narrow;
    std::vector<int> b;
    for (int c = 0; c < n; c += B) b.push_back(c);
    b.push_back(n);
    const int np = (int)b.size() - 1;
    const int nt = (n + 127) / 128;
    EventPool* pool = event_pool_of(side);
    // Three streams.  chain (the context's): pivot blocks.  rest: the bulk of every step's update, one persistent launch after the
    // other.  side: everything else of a step -- W, the swept panel, the tiles the next steps' pivots need (prio), the next
    // panel's copy -- next to the rest launch of the step before or of its own.
    static const int restIdx = getenv("DCA_SWEEP_REST_STREAM") ? atoi(getenv("DCA_SWEEP_REST_STREAM")) % kSideDepths : 0;
    static const int sideIdx = getenv("DCA_SWEEP_SIDE_STREAM") ? atoi(getenv("DCA_SWEEP_SIDE_STREAM")) % kSideDepths : 1;
    hipStream_t chain = ctx->stream, rest = side->s[restIdx], sd = side->s[sideIdx];
    if (!getenv("DCA_SWEEP_REST_STREAM") && !getenv("DCA_SWEEP_SIDE_STREAM")) sweep_streams(side, chain, rest, sd);
    int bulkCus = 256;
    if (cfg.maskCus > 0 && cfg.maskCus < 256 && side_set_masked(side, cfg.maskCus)) { rest = side->masked[0]; sd = side->masked[1]; bulkCus = cfg.maskCus; }
    const int capAll = std::max(1, cfg.perCu) * bulkCus;
    int capRest = cfg.cap > 0 ? cfg.cap : capAll * 3 / 4 / 8 * 8;           // the rest launch leaves room for the side stream's and the chain's kernels
    int capPrio = cfg.prioCap;
    const bool factorForm = n < cfg.factorMaxN;
    const unsigned* reserved = n < cfg.reserveMaxN ? sweep_reserved_cus(side, chain, cfg.reserve) : nullptr;
    if (reserved) {
        // the launches are sized so that the workgroups that stay fill the CUs that are not reserved
        const int freeCus = 256 - 8 * cfg.reserve;
        if (cfg.cap <= 0) capRest = (2 * 256 - cfg.prioCap) / 8 * 8;
        capPrio = cfg.prioCap * 256 / freeCus / 8 * 8;
    }
    // events of step p: 0 P is there (chain), 1 prio tiles done (side), 2 the chain has read the pivot panel in M (chain), 3 W is there
    // (side), 4 rest done (rest)
    constexpr int EV = 5;
    auto ev = [&](int kind, int p) { return pool->get((size_t)EV * p + kind); };
    if (!ev(EV - 1, np)) { dca_set_error("cholinv: event creation failed"); return DCA_ERR_HIP; }
    // workspace: W[2] | Q[2] | P[2] | S[2] | Wn | arena of the recursion on a pivot block | tile counters
    const size_t BB = (size_t)B * B, NB_ = (size_t)n * B;
    double* W[2] = {work, work + NB_};
    double* Q[2] = {work + 2 * NB_, work + 3 * NB_};
    double* P[2] = {work + 4 * NB_, work + 4 * NB_ + BB};
    double* S[2] = {P[1] + BB, P[1] + 2 * BB};
    double* Wn = S[1] + BB;
    Arena chainWs{Wn + BB, 2 * BB};
    int* ctr = reinterpret_cast<int*>(Wn + 3 * BB);                 // [np][2][8]
    if (4 * NB_ + 7 * BB + (size_t)np * 8 + 8 > (size_t)2 * n * n) { dca_set_error("cholinv sweep: workspace too small"); return DCA_ERR_NOMEM; }
    HIP_TRY(hipMemsetAsync(ctr, 0, (size_t)np * 16 * sizeof(int), chain));
    bool sideInFlight = false;
    int rc = DCA_OK;
    auto copy_block = [&](double* dst, int ldd, const double* src, int rows, int cols) {
        hipLaunchKernelGGL(copy_block_kernel, dim3(std::min(256, (rows * cols / 2 + 255) / 256)), dim3(256), 0, chain, dst, ldd, src, ld, rows, cols);
    };
#define SWEEP_HIP(expr) if ((expr) != hipSuccess) { dca_set_error("cholinv sweep: %s failed", #expr); rc = DCA_ERR_HIP; break; }
    copy_block(S[0], B, A, b[1], b[1]);
    if (hipEventRecord(ev(0, np), chain) != hipSuccess || hipStreamWaitEvent(sd, ev(0, np), 0) != hipSuccess) { dca_set_error("cholinv sweep: fork failed"); return DCA_ERR_HIP; }
    hipLaunchKernelGGL(gather_panel_kernel, dim3(b[1] / 32, n / 32), dim3(256), 0, sd, A, ld, 0, b[1], Q[0], B);
    sideInFlight = true;
    for (int p = 0; p < np && rc == DCA_OK; ++p) {
        const int c = b[p], w = b[p + 1] - c;
        const bool last = p + 1 == np;
        const int c1 = last ? n : b[p + 1], w1 = last ? 0 : b[p + 2] - c1;
        const int c2 = c1 + w1, w2 = (last || p + 2 == np) ? 0 : b[p + 3] - c2;
        double* Sp = S[p & 1];
        double* Pp = P[p & 1];
        double* Wp = W[p & 1];
        double* Qp = Q[p & 1];
        // ---- chain: X = inv(chol(S)), P = X^T X
        tr.mark(chain, "chain: pivot block begins", p);
        if ((rc = cholinv_rec(ctx, Sp, B, w, c, chainWs, dInfo, nullptr)) != DCA_OK) break;
        // chain-bound sizes (factorForm): P = X^T X is the side stream's, the chain goes on from X itself
        const GemmArgs pArgs{Sp, B, MASK_UPPER, Sp, B, MASK_UPPER, Pp, B, Pp, B, w, w, w, 1.0, 0.0, 1};
        if (!factorForm && (rc = launch_gemm(ctx, pArgs)) != DCA_OK) break;
        tr.mark(chain, "chain: P done", p);
        SWEEP_HIP(hipEventRecord(ev(0, p), chain));
        if (!last) {
            // the next pivot block from this one: S' = M[J', J'] - (M[J', J] P) M[J', J]^T, or with F = M[J', J] X^T (X lower: half
            // the flop, one product less on the chain) S' = M[J', J'] - F F^T
            if (p >= 1) SWEEP_HIP(hipStreamWaitEvent(chain, ev(1, p - 1), 0));
            const double* MnJ = A + (size_t)c1 * ld + c;
            if (factorForm) rc = launch_gemm(ctx, GemmArgs{MnJ, ld, MASK_NONE, Sp, B, MASK_LOWER, Wn, B, nullptr, 0, w1, w, w, 1.0, 0.0, 0, WALK_COLUMNS_REVERSED});
            else rc = launch_gemm(ctx, GemmArgs{MnJ, ld, MASK_NONE, Pp, B, MASK_NONE, Wn, B, nullptr, 0, w1, w, w, 1.0, 0.0, 0});
            if (rc != DCA_OK) break;
            // (the addend comes straight from M: the block is not copied first ...
            GemmArgs sArgs{Wn, B, MASK_NONE, factorForm ? Wn : MnJ, factorForm ? B : ld, MASK_NONE, S[(p + 1) & 1], B, nullptr, 0, w1, w1, w, -1.0, 1.0, 1};
            // ... at the chain-bound sizes (n = 4032: 3.40 -> 3.31 with F, -> 3.20 ms without the copy kernel; at n = 10 048 the
            // copy-free form measured 0.3 ms SLOWER on the same box: 20.1 / 20.5 ms)
            static const int copyEnv = getenv("DCA_SWEEP_COPY") ? atoi(getenv("DCA_SWEEP_COPY")) : -1;
            const bool copyFirst = copyEnv >= 0 ? copyEnv != 0 : !factorForm;
            if (copyFirst) copy_block(S[(p + 1) & 1], B, A + (size_t)c1 * ld + c1, w1, w1);
            else { sArgs.Cin = A + (size_t)c1 * ld + c1; sArgs.ldcin = ld; }
            if ((rc = launch_gemm(ctx, sArgs)) != DCA_OK) break;
            SWEEP_HIP(hipEventRecord(ev(2, p), chain));
            tr.mark(chain, "chain: next pivot block formed", p);
        }
        // ---- side: W = Q P (Q: this panel's copy, made at the end of the side stream's previous step), then the swept panel
        SWEEP_HIP(hipStreamWaitEvent(sd, ev(0, p), 0));
        tr.mark(sd, "side: step begins", p);
        if (factorForm && (rc = launch_gemm_on(sd, pArgs)) != DCA_OK) break;
        if ((rc = launch_gemm_on(sd, GemmArgs{Qp, B, MASK_NONE, Pp, B, MASK_NONE, Wp, B, nullptr, 0, n, w, w, 1.0, 0.0, 0})) != DCA_OK) break;
        SWEEP_HIP(hipEventRecord(ev(3, p), sd));
        tr.mark(sd, "side: W done", p);
        if (!last) SWEEP_HIP(hipStreamWaitEvent(sd, ev(2, p), 0));
        hipLaunchKernelGGL(sweep_finalize_kernel, dim3(w / 32, n / 32), dim3(256), 0, sd, A, ld, c, w, Wp, B, Pp, B);
        SweepArgs g{Wp, B, Qp, B, A, ld, n, c, w, nt, c / 128, (c1 + w1 + 127) / 128 - c / 128, SWEEP_REST, c1 / 128, (c1 + w1 + 127) / 128 - c1 / 128,
                    c2 / 128, w2 > 0 ? (c2 + w2 + 127) / 128 - c2 / 128 : 0, 0, nullptr, reserved};
        if (w1 == 0) { g.prN = 0; g.skipN = nt - g.skip0; }
        const int nR = nt - g.skipN;
        if (!last) {
            // ---- side: the next panel's tiles and the diagonal block after it (they were tiles of the rest launch one step ago), then the
            // next panel's copy
            if (p >= 1) SWEEP_HIP(hipStreamWaitEvent(sd, ev(4, p - 1), 0));
            tr.mark(sd, "side: prio begins", p);
            SweepArgs gp = g;
            gp.mode = SWEEP_PRIO;
            gp.nTiles = g.prN * nR + g.dgN * (g.dgN + 1) / 2;
            gp.ctr = ctr + (size_t)p * 16 + 8;
            if (gp.nTiles > 0) sweep_update_launch(sd, std::min(capPrio, reserved ? std::max(64, (gp.nTiles + 7) / 8 * 8 * 8 / 7) : (gp.nTiles + 7) / 8 * 8), cfg.stages, cfg.perCu, gp);
            SWEEP_HIP(hipEventRecord(ev(1, p), sd));
            tr.mark(sd, "side: prio done", p);
            hipLaunchKernelGGL(gather_panel_kernel, dim3(w1 / 32, n / 32), dim3(256), 0, sd, A, ld, c1, w1, Q[(p + 1) & 1], B);
        }
        // ---- rest
        if (nR > 0) {
            SWEEP_HIP(hipStreamWaitEvent(rest, ev(3, p), 0));
            g.mode = SWEEP_REST;
            const int bands = (nR + 3) / 4;
            g.nTiles = 8 * bands * bands + 2 * bands;
            g.ctr = ctr + (size_t)p * 16;
            tr.mark(rest, "rest: begins", p);
            sweep_update_launch(rest, std::min(last ? capAll : capRest, reserved ? std::max(64, (g.nTiles + 7) / 8 * 8 * 8 / 7) : (g.nTiles + 7) / 8 * 8), cfg.stages, cfg.perCu, g);
            tr.mark(rest, "rest: done", p);
        }
        SWEEP_HIP(hipEventRecord(ev(4, p), rest));
        if (hipGetLastError() != hipSuccess) { dca_set_error("cholinv sweep: launch failed"); rc = DCA_ERR_HIP; }
    }
#undef SWEEP_HIP
    if (rc != DCA_OK) {
        if (sideInFlight) { hipStreamSynchronize(rest); hipStreamSynchronize(sd); }   // they write A and read the workspace: drain them before the caller may free either
        tr.dump();
        return rc;
    }
    // the lower triangle is the result: mirror it, once the last panel is written (side) and the last tiles are (rest)
    const int nb32 = n / 32;
    if (hipEventRecord(ev(1, np), sd) != hipSuccess || hipStreamWaitEvent(rest, ev(1, np), 0) != hipSuccess) {
        hipStreamSynchronize(rest); hipStreamSynchronize(sd);
        dca_set_error("cholinv sweep: join of the side stream failed");
        return DCA_ERR_HIP;
    }
    hipLaunchKernelGGL(symmetrize_kernel, dim3(nb32 * (nb32 - 1) / 2), dim3(256), 0, rest, A, ld, nb32);
    hipLaunchKernelGGL(symmetrize_diag_kernel, dim3(nb32), dim3(256), 0, rest, A, ld);
    if (hipEventRecord(ev(2, np), rest) != hipSuccess || hipStreamWaitEvent(chain, ev(2, np), 0) != hipSuccess) {
        hipStreamSynchronize(rest);
        dca_set_error("cholinv sweep: join of the rest stream failed");
        return DCA_ERR_HIP;
    }
    tr.mark(chain, "sweep done", np);
    tr.dump();
    return DCA_OK;
}

}  // namespace

int dca_spd_inverse_device(dca_ctx* ctx, double* dA, int n, double* dWork, int* info_out, double scale, double** result)
{
    if (n % 64 != 0 || n <= 0) { dca_set_error("dca_spd_inverse_device: n must be a positive multiple of 64"); return DCA_ERR_ARG; }
    DCA_TRY(gemm_kernels_prepare(ctx->device));
    ScopedKernelClock kc(ctx, "mf_inverse");
    int* dInfo = nullptr;
    HIP_TRY(dca_dev_malloc(reinterpret_cast<void**>(&dInfo), sizeof(int)));
    HIP_TRY(hipMemsetAsync(dInfo, 0, sizeof(int), ctx->stream));
    Arena ws{dWork, (size_t)n * n};
    int rc;
    bool swept = false;
    {
        ScopedKernelClock kr(ctx, "mf_inverse_recursion");
        // a set of side streams only where the recursion will use one (see cholinv_rec); everything they run is joined
        // into ctx->stream before the recursion returns, so the set can go back as soon as the launches are enqueued
        const int sweepB = n >= sweep_cfg().wideMinN ? sweep_cfg().wide : sweep_cfg().narrow;
        const bool wantSweep = n >= sweep_cfg().minN && n >= 2 * sweepB &&
                               (size_t)4 * n * sweepB + (size_t)7 * sweepB * sweepB + (size_t)(n / sweepB + 2) * 8 + 8 <= (size_t)2 * n * n;
        SideSet* side = (n >= 2048 || wantSweep) ? side_set_acquire(ctx->device) : nullptr;
        if (wantSweep && side) {
            // the block sweep (round 6): -inv(A) in place
            rc = cholinv_sweep(ctx, dA, n, dWork, dInfo, side);
            swept = true;
        }
        // the blocked form keeps the factor's panels in the second half of the workspace, which X^T X overwrites at the end
        else if (n >= blocked_cfg().minN && n > 2 * blocked_cfg().W) rc = cholinv_blocked(ctx, dA, n, ws, dWork + (size_t)n * n, dInfo, side);
        else rc = cholinv_rec(ctx, dA, n, n, 0, ws, dInfo, side);
        side_set_release(side);
    }
    if (rc == DCA_OK && swept) {
        if (scale != -1.0) {
            hipLaunchKernelGGL(scale_matrix_kernel, dim3(2048), dim3(256), 0, ctx->stream, dA, (size_t)n * n, -scale);
            HIP_TRY(hipGetLastError());
        }
        *result = dA;
    } else if (rc == DCA_OK) {
        double* out = dWork + (size_t)n * n;
        // scale * inv(A)[i][j] = scale * sum_{k >= max(i,j)} X[k][i] X[k][j] = scale * sum_k Xt[i][k] Xt[j][k]
        ScopedKernelClock kx(ctx, "mf_inverse_xtx");
        // Opt-in (DCA_CHOLINV_XTX_SPLIT=1), measured and NOT adopted: with X = [[X11, 0], [X21, X22]] the product splits into
        // out11 = X21^T X21 + X11^T X11,  out21 = X22^T X21,  out22 = X22^T X22, which puts half of the flop into products
        // with no or one triangular operand (128 x 128 tiles) -- but four launches with four tails instead of one:
        // 6.25 - 6.36 ms against 5.91 - 5.99 at n = 10 048, 1.62 against 1.32 at n = 6000.
        static const bool split = getenv("DCA_CHOLINV_XTX_SPLIT") && atoi(getenv("DCA_CHOLINV_XTX_SPLIT")) != 0;
        const int n1 = (n / 128 / 2) * 128, n2 = n - n1;
        if (split && n1 >= 2048) {
            const double* Xt12 = dA + n1;                                  // X21^T: rows 0 .. n1, k over n2
            const double* Xt22 = dA + (size_t)n1 * n + n1;                 // X22^T (upper part: k >= row)
            double* o21 = out + (size_t)n1 * n;
            double* o12 = out + n1;
            double* o22 = out + (size_t)n1 * n + n1;
            rc = launch_gemm(ctx, GemmArgs{Xt12, n, MASK_NONE, Xt12, n, MASK_NONE, out, n, out, n, n1, n1, n2, scale, 0.0, 1});
            if (rc == DCA_OK) rc = launch_gemm(ctx, GemmArgs{dA, n, MASK_UPPER, dA, n, MASK_UPPER, out, n, out, n, n1, n1, n1, scale, 1.0, 1});
            if (rc == DCA_OK) rc = launch_gemm(ctx, GemmArgs{Xt22, n, MASK_UPPER, Xt12, n, MASK_NONE, o21, n, o12, n, n2, n1, n2, scale, 0.0, 0});
            if (rc == DCA_OK) rc = launch_gemm(ctx, GemmArgs{Xt22, n, MASK_UPPER, Xt22, n, MASK_UPPER, o22, n, o22, n, n2, n2, n2, scale, 0.0, 1});
        } else {
            rc = launch_gemm(ctx, GemmArgs{dA, n, MASK_UPPER, dA, n, MASK_UPPER, out, n, out, n, n, n, n, scale, 0.0, 1});
        }
        *result = out;
    }
    int info = 0;
    hipError_t e = hipMemcpyAsync(&info, dInfo, sizeof(int), hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    dca_dev_free(dInfo);
    if (e != hipSuccess) { dca_set_error("cholinv: %s", hipGetErrorString(e)); return DCA_ERR_HIP; }
    if (info_out) *info_out = info;
    return rc;
}
