// Sequence weights on MI355X: count_n = #{m : ident(n,m) >= T}, w_n = 1/count_n.
// Reference: PlmDCA::computeSeqsWeight (pydca/plmdca/plmdca_numerics.cpp:611-671, float
// compare) and compute_sequences_weight (pydca/meanfield_dca/msa_numerics.py:13-50, double
// compare).  The reference's floating-point test  ident/L > seqid  is monotone in the
// integer ident, so it is evaluated once on the host, in the reference's precision, for
// ident = 0..L and turned into an integer threshold T: the kernel is integer-exact.
//
// Integer work, VALU-bound (N^2 L / 2 site comparisons against N L bytes of input).  The
// alignment is first re-coded into BIT PLANES: for every sequence and every group of 32 sites,
// PL = 5 (q <= 32) or 3 (q <= 8) dwords, dword p holding bit p of the 32 states.  Two sequences
// then differ at the sites whose bit is set in (a0^b0)|(a1^b1)|...: PL xor + 2 or3 + one
// popcount-accumulate = 8 VALU instructions per 32 sites and pair, against 24 for the byte-wise
// form (v_xad_u32 / v_and / v_bcnt per 4 sites; 14.1 ms at D) and 32 for the first version (19.5 ms).
// 64x64 sequence tiles (upper triangle of tile pairs only), a 4x4 register block per thread,
// plane rows staged in LDS and read with 8-byte loads (row stride 26 / 18 dwords: conflict free).
// A tile whose pairs can all no longer reach the threshold skips its remaining sites.
#include <vector>

#include "dca_internal.h"

namespace {

constexpr int kTile = 64;      // sequences per tile side
constexpr int kKG = 4;         // 32-site groups per LDS stage

// Column order for the bit planes.  The identity of two sequences does not depend on the order in which their sites are
// compared, but the early exit of weights_count_kernel does: a wave stops when ALL its 1024 pairs have passed L - T
// mismatches, so the variable columns should come first and the conserved ones last.  Columns are ranked by their
// collision probability sum_a count(a)^2 (unweighted; ascending, ties by index): D 3.97 -> 3.01 ms, E 19.5 -> 17.5 ms
// (then 2.59 / 13.0 ms with the cheaper exit test, the skipped epilogue of finished waves and 16-byte staging loads)
// including the two small kernels below; the counts are the same integers whatever the order (DCA_WEIGHTS_ORDER=file: file order).
__global__ __launch_bounds__(256)
void weights_column_hist_kernel(const uint8_t* __restrict__ X, uint32_t* __restrict__ hist, int N, int L, int Ls, int seqPerBlock)
{
    __shared__ uint32_t lh[32][256];
    const int j = blockIdx.x * 256 + threadIdx.x;
    for (int a = 0; a < 32; ++a) lh[a][threadIdx.x] = 0;
    const int n0 = blockIdx.y * seqPerBlock, n1 = min(N, n0 + seqPerBlock);
    if (j < L)
        for (int n = n0; n < n1; ++n) lh[X[(size_t)n * Ls + j] & 31][threadIdx.x]++;       // own column: no conflicts
    if (j < L)
        for (int a = 0; a < 32; ++a)
            if (lh[a][threadIdx.x]) atomicAdd(&hist[j * 32 + a], lh[a][threadIdx.x]);       // integer sums: order-free
}

__global__ __launch_bounds__(1024)
void weights_column_rank_kernel(const uint32_t* __restrict__ hist, int* __restrict__ perm, int L, int Ls)
{
    extern __shared__ unsigned long long score[];
    if (!hist) {                                                // alignments too long for the LDS ranking: file order
        for (int j = threadIdx.x; j < Ls; j += blockDim.x) perm[j] = j;
        return;
    }
    for (int j = threadIdx.x; j < L; j += blockDim.x) {
        unsigned long long sc = 0;
        for (int a = 0; a < 32; ++a) { const unsigned long long c = hist[j * 32 + a]; sc += c * c; }
        score[j] = sc;
    }
    __syncthreads();
    for (int j = threadIdx.x; j < Ls; j += blockDim.x) {
        if (j >= L) { perm[j] = j; continue; }                  // padding sites (state 0 in every row) stay behind
        const unsigned long long sj = score[j];
        int rank = 0;
        for (int k = 0; k < L; ++k) rank += (score[k] < sj || (score[k] == sj && k < j)) ? 1 : 0;
        perm[rank] = j;
    }
}

template <int PL>
__global__ void weights_bitplanes_kernel(const uint8_t* __restrict__ X, const int* __restrict__ perm, uint32_t* __restrict__ P, int N, int Ls)
{
    constexpr int PLP = (PL + 1) & ~1;
    const int G = Ls / 32;
    const size_t idx = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (idx >= (size_t)N * G) return;
    const size_t n = idx / G;
    const int g = (int)(idx % G);
    const uint8_t* row = X + n * Ls;
    uint32_t planes[PL];
#pragma unroll
    for (int p = 0; p < PL; ++p) planes[p] = 0;
    for (int k = 0; k < 32; ++k) {
        const uint32_t st = row[perm[g * 32 + k]];
#pragma unroll
        for (int p = 0; p < PL; ++p) planes[p] |= ((st >> p) & 1u) << k;
    }
#pragma unroll
    for (int p = 0; p < PLP; ++p) P[idx * PLP + p] = p < PL ? planes[p] : 0u;
}

constexpr unsigned kWorkSlots = 4096;

template <int PL>
__global__ __launch_bounds__(256)
void weights_count_kernel(const uint32_t* __restrict__ P, uint32_t* __restrict__ counts, int N, int L, int G, int thresh,
                          int tilesPerSide, int part, int parts, unsigned long long* __restrict__ work)
{
    constexpr int PLP = (PL + 1) & ~1;
    constexpr int ROWDW = kKG * PLP;
    constexpr int STRIDE = ROWDW + 2;
    __shared__ __attribute__((aligned(8))) uint32_t sA[kTile * STRIDE];
    __shared__ __attribute__((aligned(8))) uint32_t sB[kTile * STRIDE];
    __shared__ unsigned sCol[kTile];
    // identity is symmetric: only tile pairs with column tile >= row tile are computed; an
    // off-diagonal tile also credits its columns' sequences (rows of the mirrored tile)
    // The 1-D grid walks the tile-pair matrix in 32 x 32 super-tiles so that the workgroups in flight
    // share row and column tiles in L2 (a row-major walk streams the whole plane array per tile row:
    // 6 GB of fabric traffic at config D for 16 MB of planes).
    // Sharded (parts > 1): this launch covers the virtual workgroups  v = blockIdx.x * parts + part, i.e. every
    // parts-th tile pair of every super-tile -- 1 / parts of the comparisons, spread evenly over the triangle.
    constexpr int S = 32;
    const int superPerSide = (tilesPerSide + S - 1) / S;
    const unsigned vb = blockIdx.x * (unsigned)parts + (unsigned)part;
    if (vb >= (unsigned)(superPerSide * superPerSide * S * S)) return;
    const int sid = vb / (S * S), within = vb % (S * S);
    const int tileY = (sid / superPerSide) * S + within / S;
    const int tileX = (sid % superPerSide) * S + within % S;
    if (tileY >= tilesPerSide || tileX >= tilesPerSide || tileX < tileY) return;
    const bool offDiag = tileX > tileY;
    const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
    const int rowBase = tileY * kTile, colBase = tileX * kTile;
    if (threadIdx.x < kTile) sCol[threadIdx.x] = 0;
    const int rowDwords = G * PLP;
    // pairs with a sequence index past N start beyond every threshold (they never count and never hold a tile back)
    unsigned mism[4][4];
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int c = 0; c < 4; ++c) mism[r][c] = (rowBase + ty + 16 * r < N && colBase + tx + 16 * c < N) ? 0u : 0x40000000u;

    // Mismatch counts only grow, so a pair that has passed L - thresh mismatches can no longer reach the
    // identity threshold; once that holds for every pair of the tile the rest of the sites are skipped
    // (exact: such pairs add nothing to the counts).  Pairs with a sequence index past N count as passed.
    // D (synthetic, Dirichlet(0.3) profiles, founder families of ~5): unrelated pairs pass after 128-256 of
    // the 500 sites, but a third of the tiles hold a same-family pair that needs ~430: 7.3 -> 5.3 ms.  (Walking
    // a strip of column tiles per workgroup to save dispatches and row-tile loads was slower: 7.0 ms.)
    // The same test per wave and 32-site group: a wave (16 x 64 pairs of the tile) whose pairs have all passed stops
    // comparing and only keeps staging -- the family pair that holds a tile to the end sits in one of its four waves,
    // and short alignments (config E: 150 sites, passed after ~64) are over before the first stage of 128 sites ends.
    const unsigned maxMism = (unsigned)(L - thresh);
    bool waveDone = false;       // wave-uniform
    unsigned groupsDone = 0;     // wave-uniform: 32-site groups this wave really compared (what the early exit leaves of G)
    auto all_passed = [&]() {          // smallest of the 16 counts against the bound: 8 x v_min3 instead of 16 compares and ands
        unsigned mn = mism[0][0];
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int c = 0; c < 4; ++c) mn = min(mn, mism[r][c]);
        return mn > maxMism;
    };

    for (int g0 = 0; g0 < G; g0 += kKG) {
        if (__syncthreads_and(waveDone)) break;      // also the barrier that protects sA / sB
        // a stage is ROWDW dwords (64 or 96 bytes, 16-byte aligned: G is a multiple of kKG) of every row: 16-byte loads,
        // 8-byte LDS stores (the LDS row stride is only 8-byte aligned)
        constexpr int Q4 = ROWDW / 4;
        for (int t = threadIdx.x; t < kTile * Q4; t += 256) {
            const int r = t / Q4, k4 = t % Q4;
            uint4 a = make_uint4(0, 0, 0, 0), b = a;
            if (rowBase + r < N) a = *reinterpret_cast<const uint4*>(P + (size_t)(rowBase + r) * rowDwords + g0 * PLP + 4 * k4);
            if (colBase + r < N) b = *reinterpret_cast<const uint4*>(P + (size_t)(colBase + r) * rowDwords + g0 * PLP + 4 * k4);
            uint2* da = reinterpret_cast<uint2*>(&sA[r * STRIDE + 4 * k4]);
            uint2* db = reinterpret_cast<uint2*>(&sB[r * STRIDE + 4 * k4]);
            da[0] = make_uint2(a.x, a.y); da[1] = make_uint2(a.z, a.w);
            db[0] = make_uint2(b.x, b.y); db[1] = make_uint2(b.z, b.w);
        }
        __syncthreads();
#pragma unroll
        for (int gg = 0; gg < kKG; ++gg) {
            if (g0 + gg > 0 && !waveDone) waveDone = __all(all_passed());
            if (waveDone) continue;
            ++groupsDone;
            uint32_t a[4][PLP], b[4][PLP];
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int h = 0; h < PLP / 2; ++h) {
                    const uint2 v = *reinterpret_cast<const uint2*>(&sA[(ty + 16 * r) * STRIDE + gg * PLP + 2 * h]);
                    a[r][2 * h] = v.x; a[r][2 * h + 1] = v.y;
                }
#pragma unroll
            for (int c = 0; c < 4; ++c)
#pragma unroll
                for (int h = 0; h < PLP / 2; ++h) {
                    const uint2 v = *reinterpret_cast<const uint2*>(&sB[(tx + 16 * c) * STRIDE + gg * PLP + 2 * h]);
                    b[c][2 * h] = v.x; b[c][2 * h + 1] = v.y;
                }
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    uint32_t d = a[r][0] ^ b[c][0];
#pragma unroll
                    for (int p = 1; p < PL; ++p) d |= a[r][p] ^ b[c][p];
                    mism[r][c] += __popc(d);
                }
        }
        if (!waveDone) waveDone = __all(all_passed());
    }
    // work[]: wave x 32-site groups compared (each = 16 pairs per lane x (PL xor/or + 1 popcount-add) VALU instructions):
    // the issued integer work that bench.py prices against the integer-VALU rate
    // (kWorkSlots counters, summed on the host: one address for the 1.2 million waves of config D serialised their atomics
    // into 27 ms)
    if (work && (threadIdx.x & 63) == 0 && groupsDone) atomicAdd(&work[(blockIdx.x * 4u + (threadIdx.x >> 6)) % kWorkSlots], (unsigned long long)groupsDone);
    // ident = L - mismatches (padding sites are state 0 in every row and never mismatch).  A wave whose pairs have all
    // passed the bound has nothing to count (ident >= thresh <=> mismatches <= L - thresh): most waves skip the epilogue.
    if (!waveDone) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        unsigned cnt = 0;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const int m = colBase + tx + 16 * c;
            if (m < N && (int)(L - mism[r][c]) >= thresh) cnt++;
        }
        // sum over the 16 tx lanes that share this row (lanes differ in the low 4 bits)
        for (int off = 8; off > 0; off >>= 1) cnt += __shfl_xor(cnt, off);
        const int n = rowBase + ty + 16 * r;
        if (tx == 0 && n < N && cnt) atomicAdd(&counts[n], cnt);
    }
    }
    if (offDiag) {
        if (!waveDone) {
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            unsigned cnt = 0;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int n = rowBase + ty + 16 * r;
                if (n < N && (int)(L - mism[r][c]) >= thresh) cnt++;
            }
            if (cnt) atomicAdd(&sCol[tx + 16 * c], cnt);     // integer LDS atomics: order-independent
        }
        }
        __syncthreads();
        const int m = colBase + threadIdx.x;
        if (threadIdx.x < kTile && m < N && sCol[threadIdx.x]) atomicAdd(&counts[m], sCol[threadIdx.x]);
    }
}

__global__ void weights_finish_kernel(const uint32_t* __restrict__ counts, double* __restrict__ wd, int N)
{
    const int n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n < N) wd[n] = 1.0 / (double)counts[n];
}

}  // namespace

int dca_weights_compute(dca_ctx* ctx, double seqid, int compare_precision, int part, int parts, bool finish)
{
    if (parts < 1 || part < 0 || part >= parts) { dca_set_error("weights: bad part / parts"); return DCA_ERR_ARG; }
    const int N = ctx->N, L = ctx->L;
    // smallest ident for which the reference's test is true, evaluated in its precision
    int thresh = L + 1;
    for (int k = 0; k <= L; ++k) {
        bool hit;
        if (compare_precision == DCA_F32) hit = ((float)k / (float)L) > (float)seqid;
        else hit = ((double)k / (double)L) > seqid;
        if (hit) { thresh = k; break; }
    }
    HIP_TRY(hipMemsetAsync(ctx->dCounts, 0, (size_t)N * sizeof(uint32_t), ctx->stream));
    {
        ScopedKernelClock kc(ctx, "weights");
        const int G = ctx->Ls / 32;
        const bool small = ctx->q <= 8;
        const int PLP = small ? 4 : 6;
        // column order (most variable sites first), from unweighted single-site counts:
        // worth its two small kernels and the gathered plane build only for large problems (C, N = 10k: 0.15 -> 0.22 ms with it)
        // (DCA_WEIGHTS_ORDER=file / variable forces one or the other)
        constexpr int kSeqPerBlock = 256;
        const char* orderEnv = getenv("DCA_WEIGHTS_ORDER");
        const bool wantRanked = orderEnv ? (orderEnv[0] == 'v') : (double)N * N * L >= 2e11;
        const bool ranked = wantRanked && (size_t)L * sizeof(unsigned long long) <= 60000;
        uint32_t *dP = nullptr, *dHist = nullptr;
        int* dPerm = nullptr;
        unsigned long long* dWork = nullptr;
        hipError_t ea = dca_dev_malloc(reinterpret_cast<void**>(&dP), (size_t)N * G * PLP * sizeof(uint32_t));
        if (ea == hipSuccess) ea = dca_dev_malloc(reinterpret_cast<void**>(&dPerm), (size_t)ctx->Ls * sizeof(int));
        // the work counter (a 64-bit atomic per wave, a 32 KB read-back: what bench.py prices the kernel with) only on request
        const bool countWork = getenv("DCA_WEIGHTS_WORK") && atoi(getenv("DCA_WEIGHTS_WORK")) != 0;       // read per call: bench.py asks for ONE counted pass
        ctx->weightsWork[0] = ctx->weightsWork[1] = 0;
        if (countWork) {
            if (ea == hipSuccess) ea = dca_dev_malloc(reinterpret_cast<void**>(&dWork), kWorkSlots * sizeof(unsigned long long));
            if (ea == hipSuccess) ea = hipMemsetAsync(dWork, 0, kWorkSlots * sizeof(unsigned long long), ctx->stream);
        }
        if (ea == hipSuccess && ranked) {            // the site histogram only exists for the ranked order
            ea = dca_dev_malloc(reinterpret_cast<void**>(&dHist), (size_t)L * 32 * sizeof(uint32_t));
            if (ea == hipSuccess) ea = hipMemsetAsync(dHist, 0, (size_t)L * 32 * sizeof(uint32_t), ctx->stream);
        }
        if (ea != hipSuccess) {                      // nothing of the scratch is left behind
            dca_dev_free(dP); dca_dev_free(dHist); dca_dev_free(dPerm); dca_dev_free(dWork);
            dca_set_error("weights scratch: %s", hipGetErrorString(ea));
            return DCA_ERR_HIP;
        }
        if (ranked)
            hipLaunchKernelGGL(weights_column_hist_kernel, dim3(ceil_div(L, 256), ceil_div(N, kSeqPerBlock)), dim3(256), 0, ctx->stream,
                               ctx->dX, dHist, N, L, ctx->Ls, kSeqPerBlock);
        hipLaunchKernelGGL(weights_column_rank_kernel, dim3(1), dim3(1024), ranked ? (size_t)L * sizeof(unsigned long long) : 0, ctx->stream,
                           ranked ? dHist : nullptr, dPerm, L, ctx->Ls);
        const unsigned tb = (unsigned)(((size_t)N * G + 255) / 256);
        const int tilesPerSide = ceil_div(N, kTile);
        const int superPerSide = ceil_div(tilesPerSide, 32);
        dim3 grid((unsigned)ceil_div(superPerSide * superPerSide * 32 * 32, parts));
        if (small) {
            hipLaunchKernelGGL(weights_bitplanes_kernel<3>, dim3(tb), dim3(256), 0, ctx->stream, ctx->dX, dPerm, dP, N, ctx->Ls);
            hipLaunchKernelGGL(weights_count_kernel<3>, grid, dim3(256), 0, ctx->stream, dP, ctx->dCounts, N, L, G, thresh, tilesPerSide, part, parts, dWork);
        } else {
            hipLaunchKernelGGL(weights_bitplanes_kernel<5>, dim3(tb), dim3(256), 0, ctx->stream, ctx->dX, dPerm, dP, N, ctx->Ls);
            hipLaunchKernelGGL(weights_count_kernel<5>, grid, dim3(256), 0, ctx->stream, dP, ctx->dCounts, N, L, G, thresh, tilesPerSide, part, parts, dWork);
        }
        hipError_t e = hipStreamSynchronize(ctx->stream);
        if (e == hipSuccess && dWork) {
            std::vector<unsigned long long> slots(kWorkSlots);
            e = hipMemcpy(slots.data(), dWork, kWorkSlots * sizeof(unsigned long long), hipMemcpyDeviceToHost);
            unsigned long long done = 0;
            for (unsigned long long v : slots) done += v;
            // without the early exit every wave of this part's tile pairs (upper triangle, diagonal included) compares all G groups
            const unsigned long long tp = (unsigned long long)tilesPerSide * (tilesPerSide + 1) / 2;
            ctx->weightsWork[0] = done;
            ctx->weightsWork[1] = (tp + parts - 1 - part) / parts * 4ull * (unsigned long long)G;
        }
        ctx->weightsPlanes = small ? 3 : 5;
        dca_dev_free(dWork);
        dca_dev_free(dP);
        dca_dev_free(dHist);
        dca_dev_free(dPerm);
        if (e != hipSuccess) { dca_set_error("weights kernel: %s", hipGetErrorString(e)); return DCA_ERR_HIP; }
    }
    ctx->have_weights = false;
    ctx->have_counts = false;          // ctx->dCounts holds this part's counts only; dca_weights_finish declares them complete
    return finish ? dca_weights_finish(ctx) : DCA_OK;
}

int dca_weights_finish(dca_ctx* ctx)
{
    const int N = ctx->N;
    hipLaunchKernelGGL(weights_finish_kernel, dim3(ceil_div(N, 256)), dim3(256), 0, ctx->stream, ctx->dCounts, ctx->dWd, N);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    // Meff in double, ascending n (deterministic)
    std::vector<double> w(N);
    HIP_TRY(hipMemcpy(w.data(), ctx->dWd, (size_t)N * sizeof(double), hipMemcpyDeviceToHost));
    double s = 0;
    for (int n = 0; n < N; ++n) s += w[n];
    ctx->meff = s;
    ctx->have_weights = true;
    ctx->have_counts = true;
    return DCA_OK;
}
