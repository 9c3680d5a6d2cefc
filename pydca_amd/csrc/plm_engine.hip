// plmDCA on MI355X: objective + "gradient" of the reference (PlmDCA::gradient,
// pydca/plmdca/plmdca_numerics.cpp:436-607) and the L-BFGS driver the backend runs
// (pydca/plmdca/plmdcaBackend.cpp:47-146, lbfgs/lib/lbfgs.cpp:248-644, :815-1004,
// :1128-1295), re-designed for gfx950.
//
// One evaluation =
//   expand : packed x -> symmetric table  W[(j,b)][(i,a)]              (HBM-bound, P floats in)
//   logits : S[n][(i,a)] = sum_j W[(j,x_nj)][(i,a)]                     (register gather, source row selected by M0)
//   softmax: per site, scan over n with the reference's carried-over probabilities,
//            R[n][(i,a)] = w_n (p_ni(a) - delta(a,x_ni)),  fx -= w_n log p_ni(x_ni)   (lanes = sites)
//   scatter: G[(j,b)][(i,a)] = sum_n [x_nj = b] R[n][(i,a)]             (LDS gather, destination sum selected by M0)
//   fold   : g = 2 lambda x + G + G^T in the packed layout, regulariser value
// The two N*L^2*q stages (logits, scatter) are gathers with a wave-uniform row index: lanes are
// columns and the VGPR index mode picks the register (generated inner blocks, tools/gen_plm_asm.py).
// They are bound on chip (fp32 adds + one SALU per 512-byte row piece); see DESIGN.md section 4.
#include <algorithm>
#include <chrono>
#include <cmath>

#include "dca_internal.h"

namespace {

template <typename T> struct V16;
template <> struct V16<float> { using type = float4; };
template <> struct V16<double> { using type = double2; };


#ifdef DCA_FAST_EXP
__device__ __forceinline__ float t_exp(float v) { return __expf(v); }
#else
__device__ __forceinline__ float t_exp(float v) { return expf(v); }
#endif
__device__ __forceinline__ double t_exp(double v) { return exp(v); }
__device__ __forceinline__ float t_log(float v) { return logf(v); }
__device__ __forceinline__ double t_log(double v) { return log(v); }

__host__ __device__ __forceinline__ size_t pair_index(int L, int i, int j)
{
    return (size_t)L * (L - 1) / 2 - (size_t)(L - i) * (L - i - 1) / 2 + (size_t)(j - i - 1);
}

// block (i<j) from linear pair index (host side builds the table once)
struct PairIJ { uint16_t i, j; };

// ------------------------------------------------------------------ expand
// W[(j,b)][(i,a)] = W[(i,a)][(j,b)] = J_ij(a,b); diagonal blocks and padding stay 0.
// One workgroup per site pair; the q x q block goes through LDS so that both
// writes are runs of q contiguous elements.
// Column window [s0, s1) of sites (the whole alignment unless the column-strip decomposition is on): W, S, R and G hold the
// columns of those sites only, re-based to column 0; their rows always cover all sites.
template <typename T>
__global__ void plm_expand_kernel(const T* __restrict__ x, T* __restrict__ W, const PairIJ* __restrict__ pairs,
                                  int L, int q, int Cs, int s0, int s1)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char dca_smem[];
    T* tile = reinterpret_cast<T*>(dca_smem);
    const int p = blockIdx.x;
    const int i = pairs[p].i, j = pairs[p].j;
    const bool iIn = i >= s0 && i < s1, jIn = j >= s0 && j < s1;
    if (!iIn && !jIn) return;                    // uniform over the workgroup
    const int q2 = q * q;
    const T* src = x + (size_t)L * q + (size_t)p * q2;
    for (int t = threadIdx.x; t < q2; t += blockDim.x) tile[t] = src[t];   // tile[a*q+b]
    __syncthreads();
    for (int t = threadIdx.x; t < q2; t += blockDim.x) {
        const int r = t / q, c = t % q;
        // row (i,a=r), columns (j,b=c): contiguous in b
        if (jIn) W[(size_t)(i * q + r) * Cs + (j - s0) * q + c] = tile[r * q + c];
        // row (j,b=r), columns (i,a=c): contiguous in a
        if (iIn) W[(size_t)(j * q + r) * Cs + (i - s0) * q + c] = tile[c * q + r];
    }
}

// ------------------------------------------------------------------ logits
// S[n][c] = sum_j W[j*q + x_nj][c].  The transpose of the scatter kernel: a workgroup owns a block
// of sequences (NS per wave) and one 512-byte column strip (lane = 8 bytes of a row) and walks the
// sites in tiles of JT = 128/q sites whose q rows each are double-buffered in LDS by LDS-DMA.
// For one site a wave pulls the q rows into q register pairs (ds_read_b64, immediate offsets) and
// then adds, for each of its NS sequences, the row of that sequence's state: the SOURCE register
// is selected with the VGPR index mode (M0 = 0x2000 | 2 x state, src1 relative), so a
// (sequence, site) pair costs one SALU write of M0 and one packed add; the NS running sums are
// fixed registers.  No per-lane LDS addresses, hence no bank conflicts and no row permutation.
// Inner block: generated assembly (tools/gen_plm_asm.py -> logits_gather_asm.inc), accumulator and
// row registers pinned.  Workgroup shape per q (generator LOGITS_CFG): q=21 runs 8 waves x 96
// sequences on 256 VGPRs (768 sequences share one staged tile; the fixed per-site cost of fetching
// the q rows is spread over 96 adds; the site's 48 state dwords live in ONE SGPR set refilled in
// place, in thirds, see the generator), q=5 runs 16 waves x 48 sequences on 128.
typedef float dca_v32f __attribute__((ext_vector_type(32)));
typedef float dca_v16f __attribute__((ext_vector_type(16)));
typedef float dca_v8f __attribute__((ext_vector_type(8)));
typedef float dca_v2f __attribute__((ext_vector_type(2)));
typedef uint32_t dca_v4u __attribute__((ext_vector_type(4)));

#include "logits_gather_asm.inc"

// "q = 25" in the helpers and kernel templates below is the SITE-PAIR ALPHABET of q = 5 (float32 only, round 5): the unit a
// gather block walks is a pair of neighbouring sites (2 jp, 2 jp + 1) with the combined state 5 x1 + x2.  Logits: the 25
// sums W[(j1, b1)] + W[(j2, b2)] are formed once per wave and pair in registers (10 row reads + 25 packed adds), after which
// a sequence costs ONE M0 write and ONE indexed add per PAIR of sites -- (25 + 80) adds per 160 (sequence, site) units.
// Scatter: 25 accumulators per pair selected by the combined state, one add per row and pair, marginalised to 5 + 5 sums
// when the workgroup stores.  An odd L pairs its last site with a padding site (state 0; its rows of W are zero, its rows
// of G lie in the padding of the allocation).  The sums are re-associated, so this is a float32 formulation; the
// float64 (parity) mode keeps the per-site blocks.
constexpr int kPairQ = 25;
__host__ __device__ constexpr int logits_waves(int q) { return q == 21 ? DCA_LOGITS_WAVES_Q21 : q == kPairQ ? DCA_LOGITS_WAVES_Q25 : DCA_LOGITS_WAVES_Q5; }
__host__ __device__ constexpr int logits_nseq(int q) { return q == 21 ? DCA_LOGITS_NSEQ_Q21 : q == kPairQ ? DCA_LOGITS_NSEQ_Q25 : DCA_LOGITS_NSEQ_Q5; }   // per wave
__host__ __device__ constexpr int logits_seq_per_wg(int q) { return logits_waves(q) * logits_nseq(q); }
// 64-byte lines that the 2*nseq bytes of one wave's state words of one site can span (their offset
// is a multiple of 2*nseq)
__host__ __device__ constexpr int logits_lines_per_site(int nseq)
{
    const int bytes = 2 * nseq;
    const int g = (bytes & -bytes) > 64 ? 64 : (bytes & -bytes);
    return (64 - g + bytes + 63) / 64;
}
__host__ __device__ constexpr int logits_jt(int q) { return q == 21 ? 6 : q == kPairQ ? 12 : 25; }   // sites (q = 25: site pairs) per LDS tile (<= 128 rows)
__host__ __device__ constexpr int logits_tile_rows(int q) { return q == kPairQ ? 12 * 2 * 5 : logits_jt(q) * q; }

// XL[j][n] = 0x2000 | 2 * x_nj (M0 image: src1-relative + register-pair offset); state 0 past N and for
// the padding sites j >= L of the last tile (their rows of W are zero)
__global__ void plm_build_logit_states_kernel(const uint8_t* __restrict__ X, uint16_t* __restrict__ XL, int N, int Npad,
                                              int L, int Ls)
{
    const int n = blockIdx.x * blockDim.x + threadIdx.x;
    const int j = blockIdx.y;
    if (n >= Npad) return;
    XL[(size_t)j * Npad + n] = (uint16_t)(0x2000u | ((n < N && j < L) ? 2u * X[(size_t)n * Ls + j] : 0u));
}

// site-pair alphabet (q = 5): the same M0 images over the combined state 5 x_{n,2jp} + x_{n,2jp+1}; `tag` = 0x2000 for the
// logits kernel (row stride Npad, sequences from 0), 0x9000 for the scatter kernel (row stride NT, owned sequences from halo);
// state 0 past N, for the padding pairs of the last tile and for the padding site that an odd L pairs its last site with
__global__ void plm_build_pair_states_kernel(const uint8_t* __restrict__ X, uint16_t* __restrict__ XP, int N, int stride, int L, int Ls,
                                             int first, uint32_t tag)
{
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    const int jp = blockIdx.y;
    if (k >= stride) return;
    const int n = first + k;
    uint32_t st = 0;
    if (n < N && 2 * jp < L) {
        const uint8_t* row = X + (size_t)n * Ls;
        st = 5u * row[2 * jp] + (2 * jp + 1 < L ? (uint32_t)row[2 * jp + 1] : 0u);
    }
    XP[(size_t)jp * stride + k] = (uint16_t)(tag | 2u * st);
}

template <int NP>
struct LogitsAcc { dca_v16f p[NP]; };          // sequence s of the wave: p[s / 8][2 * (s % 8) .. +1]

template <int NSEQ, int S = 0>
__device__ __forceinline__ void logits_store(const LogitsAcc<NSEQ / 8>& acc, unsigned char* rowBase, size_t rowStrideBytes, int rowsLeft)
{
    if constexpr (S < NSEQ) {
        if (S < rowsLeft)
            *reinterpret_cast<dca_v2f*>(rowBase + (size_t)S * rowStrideBytes) =
                dca_v2f{acc.p[S / 8][2 * (S % 8)], acc.p[S / 8][2 * (S % 8) + 1]};
        logits_store<NSEQ, S + 1>(acc, rowBase, rowStrideBytes, rowsLeft);
    }
}

// timing experiments only (results are wrong when set): compile with -DDCA_LOGITS_ABLATE=<bits> /
// -DDCA_SCATTER_ABLATE=<bits> -- 1 no per-tile barrier, 2 / 8 no staging of the next tile, 4 no wait for the landed pieces
#ifndef DCA_LOGITS_ABLATE
#define DCA_LOGITS_ABLATE 0
#endif
#ifndef DCA_SCATTER_ABLATE
#define DCA_SCATTER_ABLATE 0
#endif

// JTV (site-pair alphabet only): site pairs per LDS tile, 12 (0), 11 or 10 -- the engine takes the count that pads ceil(L / 2) least
template <typename T, int Q, int JTV = 0>
__global__ __launch_bounds__(logits_waves(Q) * 64)
void plm_logits_kernel(const T* __restrict__ W, const uint16_t* __restrict__ XL, T* __restrict__ S,
                       int N, int Npad, int L, int Cs, int numColTiles, int numNBlocks)
{
    constexpr int WAVES = logits_waves(Q);
    constexpr int NSEQ = logits_nseq(Q);
    constexpr int JT = JTV ? JTV : logits_jt(Q);
    constexpr int TROWS = JTV ? JTV * 2 * 5 : logits_tile_rows(Q);      // rows of W per tile (Q = 25: site pairs = 2 sites of 5 rows)
    static_assert(JTV == 0 || (Q == kPairQ && (JTV == 11 || JTV == 10)), "tile variants exist for the site-pair alphabet only");
    constexpr int CW = 512 / (int)sizeof(T);
    constexpr int TILE = 128 * 512;                 // bytes of one LDS buffer (TROWS <= 128 rows)
    static_assert(Q != kPairQ || sizeof(T) == 4, "the site-pair alphabet is a float32 formulation");
    constexpr int PIECES = TILE / 1024;             // 1 KiB (two rows) per LDS-DMA instruction
    constexpr int DMA_PER_WAVE = (PIECES + WAVES - 1) / WAVES;
    extern __shared__ __attribute__((aligned(16))) unsigned char dca_smem[];

    // XCD-aware decode: all sequence blocks of one column strip run on one XCD (workgroup id % 8) so that its slice
    // of W is served by that XCD's L2 -- for the strips that come in full sets of eight.  The sequence blocks of the
    // numColTiles % 8 strips left over go round ALL XCDs (one strip at a time): handing those strips to XCDs 0 .. r-1
    // whole left the other XCDs idle for a round (D: 83 strips, 23 rounds on three XCDs against 21 on five).
    const int id = blockIdx.x;
    const int fullCT = (numColTiles / kNumXcd) * kNumXcd;
    int ct, nb;
    if (id < fullCT * numNBlocks) {
        const int xcd = id % kNumXcd, k = id / kNumXcd;
        ct = (k / numNBlocks) * kNumXcd + xcd;
        nb = k % numNBlocks;
    } else {
        const int r = id - fullCT * numNBlocks;
        ct = fullCT + r / numNBlocks;
        nb = r % numNBlocks;
    }

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n0 = nb * (WAVES * NSEQ) + wave * NSEQ;

    LogitsAcc<NSEQ / 8> acc;
#pragma unroll
    for (int i = 0; i < NSEQ / 8; ++i)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc.p[i][e] = 0.f;

    const unsigned char* Wbytes = reinterpret_cast<const unsigned char*>(W + (size_t)ct * CW);   // the strip, wave-uniform
    const size_t rowStrideBytes = (size_t)Cs * sizeof(T);
    // LDS-DMA piece p of a tile = rows 2p, 2p+1 (lanes 0-31 / 32-63, 16 bytes per lane); tile jt = rows
    // [jt*JT*Q, +128) of W (the allocation is padded so that the last tile can over-read)
    const uint32_t voff = (uint32_t)((lane >> 5) * rowStrideBytes + (lane & 31) * 16);
    const uint32_t ginc = (uint32_t)(WAVES * 2 * rowStrideBytes);
    auto stage = [&](int jt, int buf) {       // all pieces of this wave at once: only for tile 0
#pragma unroll
        for (int i = 0; i < DMA_PER_WAVE; ++i) {
            const int pairIdx = wave + i * WAVES;                     // wave-uniform
            if (PIECES % WAVES != 0 && pairIdx >= PIECES) break;
            __builtin_amdgcn_global_load_lds(
                (const __attribute__((address_space(1))) void*)(Wbytes + (size_t)(jt * TROWS + pairIdx * 2) * rowStrideBytes + voff),
                (__attribute__((address_space(3))) void*)(dca_smem + buf * TILE + pairIdx * 1024), 16, 0, 0);
        }
    };

    // The state words are a stream (N*L*2 bytes per strip, far beyond the L2), so the scalar loads
    // of the inner block, issued one site ahead, would wait for HBM at every site.  One vector
    // load per wave and tile touches every 64-byte line of the NEXT tile's state words (lane ->
    // (site, line)), a whole tile ahead; its data goes to a scratch corner of the LDS and is
    // never read -- the point is that the scalar loads then hit the L2 (D: 7.98 -> 7.55 ms, E: 0.79 ->
    // 0.72 ms).  The scatter kernel loads its state words a quarter tile ahead and gains nothing from this.
    constexpr int LPS = logits_lines_per_site(NSEQ);
    static_assert(JT * LPS <= 64, "one prefetch lane per (site, line)");
    const int pfSite = min(lane / LPS, JT - 1);
    const size_t pfLane = (size_t)pfSite * Npad * 2 + (size_t)n0 * 2 + min((lane % LPS) * 64, NSEQ * 2 - 4);
    auto prefetch_states = [&](int jt) {
        __builtin_amdgcn_global_load_lds(
            (const __attribute__((address_space(1))) void*)(reinterpret_cast<const unsigned char*>(XL) + (size_t)jt * JT * Npad * 2 + pfLane),
            (__attribute__((address_space(3))) void*)(dca_smem + 2 * TILE + wave * 256), 4, 0, 0);
    };

    const int numJT = (L + JT - 1) / JT;
    const uint32_t ldsBase = (uint32_t)(uintptr_t)dca_smem + lane * 8;
    stage(0, 0);
    for (int jt = 0; jt < numJT; ++jt) {
        const int buf = jt & 1;
        if (!(DCA_LOGITS_ABLATE & 4)) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's pieces of tile jt have landed
        if (!(DCA_LOGITS_ABLATE & 1)) __syncthreads();                 // ... everyone's; and tile jt-1 is no longer read
        if (jt + 1 < numJT && !(DCA_LOGITS_ABLATE & 8)) prefetch_states(jt + 1);
        const uint16_t* sp = XL + (size_t)jt * JT * Npad + n0;     // wave-uniform
        const uint32_t vbase = ldsBase + buf * TILE;
        const uint32_t strideBytes = (uint32_t)Npad * 2u;
        // the block also issues this wave's LDS-DMA pieces of tile jt+1 (piece i = wave + i*WAVES), spread over its sites
        const uint32_t npc = __builtin_amdgcn_readfirstlane((jt + 1 < numJT && !(DCA_LOGITS_ABLATE & 2)) ? (uint32_t)((PIECES - wave + WAVES - 1) / WAVES) : 0u);
        const unsigned char* gbase = Wbytes + (size_t)((jt + 1) * TROWS + wave * 2) * rowStrideBytes;     // wave-uniform
        const uint32_t ldst = (uint32_t)(uintptr_t)dca_smem + (buf ^ 1) * TILE + wave * 1024;
        if constexpr (Q == kPairQ && JTV == 11) DCA_LOGITS_Q25J11_F32(vbase, sp, strideBytes, npc, gbase, ginc, voff, ldst, acc);
        else if constexpr (Q == kPairQ && JTV == 10) DCA_LOGITS_Q25J10_F32(vbase, sp, strideBytes, npc, gbase, ginc, voff, ldst, acc);
        else if constexpr (Q == kPairQ) DCA_LOGITS_Q25_F32(vbase, sp, strideBytes, npc, gbase, ginc, voff, ldst, acc);
        else if constexpr (Q == 21 && sizeof(T) == 4) DCA_LOGITS_Q21_F32(vbase, sp, strideBytes, npc, gbase, ginc, voff, ldst, acc);
        else if constexpr (Q == 21) DCA_LOGITS_Q21_F64(vbase, sp, strideBytes, npc, gbase, ginc, voff, ldst, acc);
        else if constexpr (sizeof(T) == 4) DCA_LOGITS_Q5_F32(vbase, sp, strideBytes, npc, gbase, ginc, voff, ldst, acc);
        else DCA_LOGITS_Q5_F64(vbase, sp, strideBytes, npc, gbase, ginc, voff, ldst, acc);
    }
    if (n0 < N)
        logits_store<NSEQ>(acc, reinterpret_cast<unsigned char*>(S + (size_t)n0 * Cs + (size_t)ct * CW) + lane * 8,
                     rowStrideBytes, N - n0);
}

// ------------------------------------------------------------------ softmax scan
// Lanes = sites, the q states of a site live in registers, so the softmax needs no
// cross-lane traffic.  Each wave owns one chunk of consecutive sequences and walks it
// serially carrying p_{n-1} (plmdca_numerics.cpp:492-530).  In chunked mode a chunk
// starts `warm` sequences early from a zero carry: the carry enters the logits with
// weight <= 1 and d softmax has 1-norm <= 1/2, so the start-up error shrinks by >= 2x
// per step (2^-40 after the default 40) -- far below float/double rounding.
// In: S (logit sums).  Out: R = w_n (p - delta) in a SEPARATE array, fxPart[2 wave], [2 wave + 1] = -sum w_n log p(x_ni) (hi, lo).
// (Not in place: a chunk's warm-up rows belong to its predecessors, which would be overwriting them with R at
// the same time -- chunk 0 has no warm-up and writes row t at its step t while chunk 1 reads it at its step t.)
//
// Memory access: the 64 sites of a wave are one contiguous 64*q*sizeof(T)-byte span of a row.
// It is fetched with 16-byte loads (prefetched DEPTH rows ahead into registers), transposed
// through a wave-private LDS buffer (lane l then reads its q values at stride q: conflict free
// for odd q) and written back the same way, instead of q strided 4-byte accesses per lane.
typedef uint4 __attribute__((may_alias)) dca_u4a;

// The objective is summed in double-double (error-free TwoSum): N*L terms in whatever order the kernels meet them
// would otherwise leave ~1e-13 of rounding noise in fx, the line search interpolates on DIFFERENCES of fx, and over 100
// iterations of an optimisation that does not converge that noise grew to 6e-4 in the scores at config E
// (profiles/r03_e_sensitivity_cap100_plain_sums.json).  An (almost) exact sum does not depend on the order: chunked scan, serial
// chain, any sharding and the float64 oracle (Neumaier sums) then see the same fx to the last bit or two.
__device__ __forceinline__ void dd_add(double& hi, double& lo, double v)
{
    const double s = hi + v;
    const double bb = s - hi;
    lo += (hi - (s - bb)) + (v - bb);
    hi = s;
}
__device__ __forceinline__ void dd_add2(double& hi, double& lo, double vh, double vl) { dd_add(hi, lo, vh); lo += vl; }
__device__ __forceinline__ void dd_wave_reduce(double& hi, double& lo)       // fixed tree over the 64 lanes; lane 0 holds the sum
{
    for (int off = 32; off > 0; off >>= 1) {
        const double vh = __shfl_down(hi, off), vl = __shfl_down(lo, off);
        dd_add2(hi, lo, vh, vl);
    }
}

__device__ __forceinline__ void dd_block_reduce(double& hi, double& lo, double* redHi, double* redLo)      // result in thread 0
{
    redHi[threadIdx.x] = hi;
    redLo[threadIdx.x] = lo;
    __syncthreads();
    for (int st = blockDim.x / 2; st > 0; st >>= 1) {
        if ((int)threadIdx.x < st) dd_add2(redHi[threadIdx.x], redLo[threadIdx.x], redHi[threadIdx.x + st], redLo[threadIdx.x + st]);
        __syncthreads();
    }
    hi = redHi[0];
    lo = redLo[0];
    __syncthreads();
}

template <typename T, int Q>
__global__ __launch_bounds__(256)
void plm_softmax_kernel(const T* __restrict__ SR, T* __restrict__ Rout, const T* __restrict__ x, const uint8_t* __restrict__ X,
                        const T* __restrict__ w, double* __restrict__ fxPart,
                        int N, int L, int Ls, int Cs, int halo, int chunk, int warm, int carry, int numChunks, double* __restrict__ colPart)
{
    constexpr int ROWB = 64 * Q * (int)sizeof(T);        // bytes of a wave's span of one row
    constexpr int NP = (ROWB + 1023) / 1024;             // 16-byte pieces per lane
#ifdef DCA_SOFTMAX_DEPTH
    constexpr int DEPTH = DCA_SOFTMAX_DEPTH;             // experiments
#else
    constexpr int DEPTH = NP > 6 ? 2 : 3;                // rows in flight
#endif
    extern __shared__ __attribute__((aligned(16))) unsigned char dca_smem[];
    const int lane = threadIdx.x & 63;
    const int wv = threadIdx.x >> 6;
    const int chunkId = blockIdx.y * 4 + wv;
    const int i0 = blockIdx.x * 64;
    const int i = i0 + lane;
    unsigned char* sIn = dca_smem + (size_t)wv * (2 * NP * 1024);
    unsigned char* sOut = sIn + NP * 1024;
    const int rowBytes = (min(64, L - i0) * Q * (int)sizeof(T) + 15) & ~15;
    double facc = 0.0, flo = 0.0;
    // float64, q = 5 (round 5): the double-double column sums of R (the field gradients, plm_colsum_*) are taken here, where
    // R is made, instead of in one more pass over it -- five more double-double accumulators per lane (for q = 21 the 42
    // registers do not fit beside the row buffers).  One partial per (chunk, site, state); colPart == nullptr: not wanted.
    constexpr bool COLSUM = sizeof(T) == 8 && Q == 5;
    [[maybe_unused]] double chi[COLSUM ? Q : 1], clo[COLSUM ? Q : 1];
    if constexpr (COLSUM) {
#pragma unroll
        for (int a = 0; a < Q; ++a) chi[a] = clo[a] = 0.0;
    }
    if (chunkId < numChunks) {
        const int s = halo + chunkId * chunk;
        const int e = min(s + chunk, N);
        const int ws = carry ? max(0, s - warm) : s;
        T h[Q], p[Q];
#pragma unroll
        for (int a = 0; a < Q; ++a) { h[a] = (i < L) ? x[(size_t)i * Q + a] : (T)0; p[a] = 0; }
        uint4 buf[DEPTH][NP];
        int xs[DEPTH];
        T wns[DEPTH];
        auto fetch = [&](int n, int d) {
            const unsigned char* row = reinterpret_cast<const unsigned char*>(SR + (size_t)n * Cs + (size_t)i0 * Q);
#pragma unroll
            for (int pc = 0; pc < NP; ++pc) {
                const int off = pc * 1024 + lane * 16;
                buf[d][pc] = (off < rowBytes) ? *reinterpret_cast<const dca_u4a*>(row + off) : make_uint4(0, 0, 0, 0);
            }
            xs[d] = (i < L) ? (int)X[(size_t)n * Ls + i] : 0;
            wns[d] = w[n];
        };
#pragma unroll
        for (int d = 0; d < DEPTH; ++d)
            if (ws + d < e) fetch(ws + d, d);
        for (int n0 = ws; n0 < e; n0 += DEPTH) {
#pragma unroll
            for (int d = 0; d < DEPTH; ++d) {
                const int n = n0 + d;
                if (n < e) {
#pragma unroll
                    for (int pc = 0; pc < NP; ++pc) *reinterpret_cast<dca_u4a*>(sIn + pc * 1024 + lane * 16) = buf[d][pc];
                    __builtin_amdgcn_wave_barrier();
                    T z[Q];
#pragma unroll
                    for (int a = 0; a < Q; ++a) z[a] = reinterpret_cast<const T*>(sIn)[lane * Q + a] + h[a];
                    const int xi = xs[d];
                    const T wn = wns[d];
                    __builtin_amdgcn_wave_barrier();
                    if (n + DEPTH < e) fetch(n + DEPTH, d);
                    if (carry) {
#pragma unroll
                        for (int a = 0; a < Q; ++a) z[a] += p[a];
                    }
                    T m = z[0];
#pragma unroll
                    for (int a = 1; a < Q; ++a) m = z[a] > m ? z[a] : m;
                    T sum = 0;
#pragma unroll
                    for (int a = 0; a < Q; ++a) { p[a] = t_exp(z[a] - m); sum += p[a]; }
                    const T inv = (T)1 / sum;
#pragma unroll
                    for (int a = 0; a < Q; ++a) p[a] *= inv;
                    if (n >= s) {
                        T px = p[0];
#pragma unroll
                        for (int a = 1; a < Q; ++a) px = (a == xi) ? p[a] : px;
                        if (i < L) dd_add(facc, flo, -(double)(wn * t_log(px)));
#pragma unroll
                        for (int a = 0; a < Q; ++a) {
                            T r = wn * p[a];
                            if (a == xi) r -= wn;
                            reinterpret_cast<T*>(sOut)[lane * Q + a] = r;
                            if constexpr (COLSUM) dd_add(chi[a], clo[a], (double)r);
                        }
                        __builtin_amdgcn_wave_barrier();
                        unsigned char* row = reinterpret_cast<unsigned char*>(Rout + (size_t)n * Cs + (size_t)i0 * Q);
#pragma unroll
                        for (int pc = 0; pc < NP; ++pc) {
                            const int off = pc * 1024 + lane * 16;
                            if (off < rowBytes) *reinterpret_cast<dca_u4a*>(row + off) = *reinterpret_cast<const dca_u4a*>(sOut + off);
                        }
                        __builtin_amdgcn_wave_barrier();
                    }
                }
            }
        }
    }
    if constexpr (COLSUM) {
        if (colPart && chunkId < numChunks && i < L) {
#pragma unroll
            for (int a = 0; a < Q; ++a) {
                const size_t o = 2 * ((size_t)chunkId * L * Q + (size_t)i * Q + a);
                colPart[o] = chi[a];
                colPart[o + 1] = clo[a];
            }
        }
    }
    // fixed-order wave reduction, one (hi, lo) partial per wave
    dd_wave_reduce(facc, flo);
    if (lane == 0) {
        const size_t slot = ((size_t)blockIdx.y * gridDim.x + blockIdx.x) * 4 + wv;
        fxPart[2 * slot] = facc;
        fxPart[2 * slot + 1] = flo;
    }
}

// ------------------------------------------------------------------ scatter (as a gather)
// G[(j,b)][c] = sum_{n : x_nj = b} R[n][c].  A workgroup owns 32 sites (two per wave) and one
// 512-byte column strip of R (lane = 8 bytes of a row) and walks the owned sequences in
// 128-row tiles that are double-buffered in LDS by LDS-DMA (global_load_lds_dwordx4, 1 KiB per
// wave instruction, no staging registers): tile c+1 streams in while tile c is gathered, one
// barrier per tile.  The q accumulators of a site sit in fixed VGPRs and the one that a row adds
// to is selected with the gfx9 VGPR index mode (s_set_gpr_idx_on; M0 = 0x9000 | 2 x state), so the rows
// are visited in sequence order with immediate LDS offsets: one ds_read_b64 per row shared by the
// wave's two sites and one packed add per (row, site).  The M0 images of the states (XT2) reach SGPRs
// through scalar loads, one s_load_dwordx16 per site and quarter tile, issued a quarter ahead.  The inner block is generated assembly
// (tools/gen_scatter_asm.py -> scatter_gather_asm.inc): 84 accumulator + 16 data-ring registers
// are pinned, which is why the kernel is built for 128 VGPRs (16 waves = one workgroup per CU).
// The sums of a (site, state) run over n in ascending order: deterministic.
constexpr int kNC = 128;           // sequences per scatter tile
constexpr int kRowBytes = 512;     // bytes of one staged row (64 lanes x 8 B)
constexpr int kScatWavesC = 16;
constexpr int kCanonBlock = 16384;  // float64 mode: sequences per block of the canonical summation order (= ORACLE_CANONICAL_BLOCK)
static_assert(kCanonBlock % kNC == 0, "canonical blocks are whole tiles");

// XT2[j][k] = 0x9000 | 2 * x_{halo+k, j}: the M0 image that selects the accumulator of the state
// (index-enable bits for src0 and dst + register-pair offset); state 0 past N (zero rows); row stride NT
__global__ void plm_build_states_kernel(const uint8_t* __restrict__ X, uint16_t* __restrict__ XT2, int N, int L, int Ls,
                                        int halo, int NT)
{
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    const int j = blockIdx.y;
    if (k >= NT) return;
    const int n = halo + k;
    XT2[(size_t)j * NT + k] = (uint16_t)(0x9000u | ((n < N) ? 2u * X[(size_t)n * Ls + j] : 0u));
}

#include "scatter_gather_asm.inc"

// accumulators of one site as the register tuples the generated assembly pins
template <int Q> struct SiteAcc;
template <> struct SiteAcc<21> {
    dca_v32f a; dca_v8f b; dca_v2f c;
    __device__ __forceinline__ void zero() {
#pragma unroll
        for (int i = 0; i < 32; ++i) a[i] = 0.f;
#pragma unroll
        for (int i = 0; i < 8; ++i) b[i] = 0.f;
        c[0] = c[1] = 0.f;
    }
    template <int S> __device__ __forceinline__ dca_v2f get() const {
        if constexpr (S < 16) return dca_v2f{a[2 * S], a[2 * S + 1]};
        else if constexpr (S < 20) return dca_v2f{b[2 * (S - 16)], b[2 * (S - 16) + 1]};
        else return c;
    }
};
template <> struct SiteAcc<5> {
    dca_v8f a; dca_v2f b;
    __device__ __forceinline__ void zero() {
#pragma unroll
        for (int i = 0; i < 8; ++i) a[i] = 0.f;
        b[0] = b[1] = 0.f;
    }
    template <int S> __device__ __forceinline__ dca_v2f get() const {
        if constexpr (S < 4) return dca_v2f{a[2 * S], a[2 * S + 1]};
        else return b;
    }
};

template <> struct SiteAcc<kPairQ> {            // a site PAIR of q = 5: accumulator 5 b1 + b2
    dca_v32f a; dca_v16f b; dca_v2f c;
    __device__ __forceinline__ void zero() {
#pragma unroll
        for (int i = 0; i < 32; ++i) a[i] = 0.f;
#pragma unroll
        for (int i = 0; i < 16; ++i) b[i] = 0.f;
        c[0] = c[1] = 0.f;
    }
    template <int S> __device__ __forceinline__ dca_v2f get() const {
        if constexpr (S < 16) return dca_v2f{a[2 * S], a[2 * S + 1]};
        else if constexpr (S < 24) return dca_v2f{b[2 * (S - 16)], b[2 * (S - 16) + 1]};
        else return c;
    }
};

// site pair -> the 5 + 5 rows of its two sites: G[(2 jp, b1)] = sum_b2 A[5 b1 + b2], G[(2 jp + 1, b2)] = sum_b1 A[5 b1 + b2],
// each in ascending order of the summed state; rowBase = row (2 jp, 0) of the strip
template <int B = 0>
__device__ __forceinline__ void scatter_store_pair(const SiteAcc<kPairQ>& acc, unsigned char* rowBase, size_t rowStrideBytes)
{
    if constexpr (B < 5) {
        const dca_v2f first = (((acc.template get<5 * B>() + acc.template get<5 * B + 1>()) + acc.template get<5 * B + 2>()) +
                               acc.template get<5 * B + 3>()) + acc.template get<5 * B + 4>();
        const dca_v2f second = (((acc.template get<B>() + acc.template get<5 + B>()) + acc.template get<10 + B>()) +
                                acc.template get<15 + B>()) + acc.template get<20 + B>();
        *reinterpret_cast<dca_v2f*>(rowBase + (size_t)B * rowStrideBytes) = first;
        *reinterpret_cast<dca_v2f*>(rowBase + (size_t)(5 + B) * rowStrideBytes) = second;
        scatter_store_pair<B + 1>(acc, rowBase, rowStrideBytes);
    }
}

template <int Q, int S = 0>
__device__ __forceinline__ void scatter_store_site(const SiteAcc<Q>& acc, unsigned char* rowBase, size_t rowStrideBytes)
{
    if constexpr (S < Q) {
        *reinterpret_cast<dca_v2f*>(rowBase + (size_t)S * rowStrideBytes) = acc.template get<S>();
        scatter_store_site<Q, S + 1>(acc, rowBase, rowStrideBytes);
    }
}

// float64 mode, end of a canonical block that is not the workgroup's first: G = G + (the block's sums) -- the running sum
// of the finished blocks on the left, as the float64 oracle of the test suite adds them (its ORACLE_CANONICAL_BLOCK).  A lane's 8
// bytes are one double; every address is read and written by this lane only.
template <int Q, int S, int GROUP, int K = 0>
__device__ __forceinline__ void scatter_add_rows_f64(const SiteAcc<Q>& acc, const double (&v)[GROUP], unsigned char* rowBase, uint32_t laneOff,
                                                     size_t rowStrideBytes)
{
    if constexpr (K < GROUP && S + K < Q) {
        *reinterpret_cast<double*>(rowBase + (size_t)(S + K) * rowStrideBytes + laneOff) = v[K] + __builtin_bit_cast(double, acc.template get<S + K>());
        scatter_add_rows_f64<Q, S, GROUP, K + 1>(acc, v, rowBase, laneOff, rowStrideBytes);
    }
}

// rowBase: the wave-uniform address of row (site, 0) of the strip, laneOff = 8 * lane -- kept apart so that the row
// addresses are scalar base + 32-bit lane offset (no 64-bit address registers per row).
template <int Q, int S = 0>
__device__ __forceinline__ void scatter_add_site_f64(const SiteAcc<Q>& acc, unsigned char* rowBase, uint32_t laneOff, size_t rowStrideBytes)
{
    if constexpr (S < Q) {
        constexpr int GROUP = 7;          // rows in flight: the accumulators are pinned and the kernel has 128 registers
        double v[GROUP];
#pragma unroll
        for (int k = 0; k < GROUP; ++k)
            if (S + k < Q) v[k] = *reinterpret_cast<const double*>(rowBase + (size_t)(S + k) * rowStrideBytes + laneOff);
        scatter_add_rows_f64<Q, S, GROUP>(acc, v, rowBase, laneOff, rowStrideBytes);
        asm volatile("" ::: "memory");
        scatter_add_site_f64<Q, S + GROUP>(acc, rowBase, laneOff, rowStrideBytes);
    }
}

template <typename T, int Q, int JW, int WAVES_>
__global__ __launch_bounds__(WAVES_ * 64)
void plm_scatter_kernel(const T* __restrict__ R, const uint16_t* __restrict__ XT2,
                        T* __restrict__ G, int N, int L, int Cs, int halo, int numChunks, int NT, int ctBase, int numPairs, int splitX,
                        int numJG, int chunksPerSplit, size_t slabElems, int blockChunks,
                        int firstBlocksX, int ctBase2, int numPairs2, int splitX2, int chunksPerSplit2)
{
    constexpr int WAVES = WAVES_;                      // 16; 8 or 4 in the float64 mode on alignments with few column strips (configure)
    constexpr int JG = WAVES * JW;                     // sites (Q = 25: site pairs; L is then their number) per workgroup
    constexpr int QROWS = Q == kPairQ ? 10 : Q;        // rows of G per unit
    constexpr int CW = kRowBytes / (int)sizeof(T);     // columns per strip
    constexpr int DMA_PER_WAVE = kNC / 2 / WAVES;      // LDS-DMA instructions per wave and tile
    constexpr int TILE = kNC * kRowBytes;
    static_assert(kNC % (2 * WAVES) == 0, "tile rows must divide over the waves");
    static_assert(JW == 2 && (Q == 21 || Q == 5 || (Q == kPairQ && sizeof(T) == 4)), "no generated gather block for this shape");
    static_assert(WAVES == 16 || (sizeof(T) == 8 && (WAVES == 8 || WAVES == 4)), "no generated gather block for this workgroup size");
    extern __shared__ __attribute__((aligned(16))) unsigned char dca_smem[];

    // workgroup id -> (XCD, (column strip, tile-range split) pair, site group): the numJG site groups of a pair run on
    // the same XCD (id % 8) so that their reads of the strip can meet in that XCD's L2.  The main launch has one pair
    // per strip and the splits in blockIdx.y; the launch for the strips left over after the full sets of eight
    // (launch_eval) carries a finer split in the pair index (splitX) so that it fills all XCDs for a fraction of a round.
    // A launch may carry a SECOND set of (strip, split) pairs behind the first firstBlocksX workgroups of every grid row -- the
    // left-over strips with their finer split (round 6: as a launch of their own they ran AFTER the main one, which at config C
    // leaves 32 CUs idle for its whole length: 269 + 48 us; merged they fill those CUs).  The second set has no blockIdx.y.
    int id = blockIdx.x;
    if (id >= firstBlocksX) {
        if (blockIdx.y != 0) return;
        id -= firstBlocksX; ctBase = ctBase2; numPairs = numPairs2; splitX = splitX2; chunksPerSplit = chunksPerSplit2; blockChunks = 0;
    }
    const int xcd = id % kNumXcd, k = id / kNumXcd;
    const int pr = (k / numJG) * kNumXcd + xcd;
    const int jg = k % numJG;
    if (pr >= numPairs) return;
    const int ct = ctBase + pr / splitX;
    const int split = blockIdx.y + pr % splitX;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int j0 = jg * JG + wave * JW;

    // blockIdx.y splits the tile range; every split writes its own slab of G (summed by
    // the fold kernels in a fixed order), so small L*q shapes still fill the chip.
    const int cBegin = split * chunksPerSplit;
    const int cEnd = min(numChunks, cBegin + chunksPerSplit);

    SiteAcc<Q> acc[JW];
    const uint32_t* xs[JW];
#pragma unroll
    for (int jj = 0; jj < JW; ++jj) {
        acc[jj].zero();
        xs[jj] = reinterpret_cast<const uint32_t*>(XT2 + (size_t)min(j0 + jj, L - 1) * NT);
    }

    // LDS-DMA piece p of a tile = rows 2p, 2p+1 (lanes 0-31 / 32-63, 16 bytes per lane); a wave stages pieces
    // 4*wave .. 4*wave+3.  R has kNC zero rows behind row N-1, so the last tile needs no special case.
    const unsigned char* Rbytes = reinterpret_cast<const unsigned char*>(R + (size_t)ct * CW);    // the strip, wave-uniform
    const size_t rowStrideBytes = (size_t)Cs * sizeof(T);
    const uint32_t voff = (uint32_t)((lane >> 5) * rowStrideBytes + (lane & 31) * 16);
    const uint32_t ginc = (uint32_t)(2 * rowStrideBytes);
    auto tile_src = [&](int c) { return Rbytes + (size_t)(halo + c * kNC + wave * DMA_PER_WAVE * 2) * rowStrideBytes; };
    auto stage = [&](int c, int buf) {        // all four pieces at once: only for the first tile
#pragma unroll
        for (int i = 0; i < DMA_PER_WAVE && !(DCA_SCATTER_ABLATE & 8); ++i)
            __builtin_amdgcn_global_load_lds(
                (const __attribute__((address_space(1))) void*)(tile_src(c) + (size_t)i * ginc + voff),
                (__attribute__((address_space(3))) void*)(dca_smem + buf * TILE + (wave * DMA_PER_WAVE + i) * 1024), 16, 0, 0);
    };

    if (cBegin < cEnd) stage(cBegin, 0);
    const uint32_t ldsBase = (uint32_t)(uintptr_t)dca_smem + lane * 8;
    T* const Gslab = G + (size_t)split * slabElems;
    auto run_tiles = [&](int cFrom, int cTo) {
        for (int c = cFrom; c < cTo; ++c) {
            const int buf = (c - cBegin) & 1;
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's pieces of tile c have landed
            if (!(DCA_SCATTER_ABLATE & 1)) __syncthreads();                 // ... everyone's; and tile c-1 is no longer read
            const uint32_t vbase = ldsBase + buf * TILE;
            // two sites per wave: the state words come through scalar loads inside the block, which also
            // issues the wave's four LDS-DMA pieces of tile c+1, one per quarter tile
            const uint32_t* sp0 = xs[0] + c * (kNC / 2);
            const uint32_t* sp1 = xs[1] + c * (kNC / 2);
            const uint32_t npc = __builtin_amdgcn_readfirstlane((c + 1 < cEnd && !(DCA_SCATTER_ABLATE & 8)) ? 1u : 0u);
            const unsigned char* gbase = tile_src(c + 1);
            const uint32_t ldst = (uint32_t)(uintptr_t)dca_smem + (buf ^ 1) * TILE + wave * DMA_PER_WAVE * 1024;
            uint32_t vtmp;
            [[maybe_unused]] uint32_t vw;       // LDS address / staging registers of the generator's register-staged variant
            [[maybe_unused]] dca_v4u stg;       // (DCA_GEN_SC_STAGE=vgpr; the shipped LDS-DMA blocks do not use them)
            if constexpr (Q == kPairQ)
                DCA_GATHER_Q25_F32_SMEM(vbase, sp0, sp1, npc, gbase, ginc, voff, ldst, vtmp, stg, vw, acc[0].a, acc[0].b, acc[0].c, acc[1].a, acc[1].b, acc[1].c);
            else if constexpr (Q == 21 && sizeof(T) == 4)
                DCA_GATHER_Q21_F32_SMEM(vbase, sp0, sp1, npc, gbase, ginc, voff, ldst, vtmp, stg, vw, acc[0].a, acc[0].b, acc[0].c, acc[1].a, acc[1].b, acc[1].c);
            else if constexpr (Q == 21 && WAVES == 16)
                DCA_GATHER_Q21_F64_SMEM(vbase, sp0, sp1, npc, gbase, ginc, voff, ldst, vtmp, stg, vw, acc[0].a, acc[0].b, acc[0].c, acc[1].a, acc[1].b, acc[1].c);
            else if constexpr (Q == 21 && WAVES == 8)
                DCA_GATHER_Q21_F64_SMEM_W8(vbase, sp0, sp1, npc, gbase, ginc, voff, ldst, vtmp, stg, vw, acc[0].a, acc[0].b, acc[0].c, acc[1].a, acc[1].b, acc[1].c);
            else if constexpr (Q == 21)
                DCA_GATHER_Q21_F64_SMEM_W4(vbase, sp0, sp1, npc, gbase, ginc, voff, ldst, vtmp, stg, vw, acc[0].a, acc[0].b, acc[0].c, acc[1].a, acc[1].b, acc[1].c);
            else if constexpr (sizeof(T) == 4)
                DCA_GATHER_Q5_F32_SMEM(vbase, sp0, sp1, npc, gbase, ginc, voff, ldst, vtmp, stg, vw, acc[0].a, acc[0].b, acc[1].a, acc[1].b);
            else if constexpr (WAVES == 16)
                DCA_GATHER_Q5_F64_SMEM(vbase, sp0, sp1, npc, gbase, ginc, voff, ldst, vtmp, stg, vw, acc[0].a, acc[0].b, acc[1].a, acc[1].b);
            else if constexpr (WAVES == 8)
                DCA_GATHER_Q5_F64_SMEM_W8(vbase, sp0, sp1, npc, gbase, ginc, voff, ldst, vtmp, stg, vw, acc[0].a, acc[0].b, acc[1].a, acc[1].b);
            else
                DCA_GATHER_Q5_F64_SMEM_W4(vbase, sp0, sp1, npc, gbase, ginc, voff, ldst, vtmp, stg, vw, acc[0].a, acc[0].b, acc[1].a, acc[1].b);
        }
    };
    auto row_base = [&](int jj, uint32_t laneOff) {
        return reinterpret_cast<unsigned char*>(Gslab + (size_t)(j0 + jj) * QROWS * Cs + (size_t)ct * CW) + laneOff;
    };
    auto row_base_uniform = [&](int jj) { return reinterpret_cast<unsigned char*>(Gslab + (size_t)(j0 + jj) * QROWS * Cs + (size_t)ct * CW); };

    if constexpr (sizeof(T) == 8) {
        // float64 (parity) mode: the chains run over canonical blocks of blockChunks tiles (kCanonBlock sequences), each
        // summed from zero; a workgroup with several blocks stores the first block's sums and adds every later block to the
        // running sum in G: ((B0 + B1) + B2) + ..., the oracle's order (blockChunks = 0: one chain over the whole range)
        const int blockLen = blockChunks > 0 ? blockChunks : max(1, cEnd - cBegin);
        for (int cb = cBegin; cb < cEnd || cb == cBegin; cb += blockLen) {
#pragma unroll
            for (int jj = 0; jj < JW; ++jj) acc[jj].zero();
            run_tiles(cb, min(cEnd, cb + blockLen));
            // the lane offset is re-made per block: as a loop invariant the 2 Q row addresses of the flush were hoisted out of
            // the block loop and spilled (142 registers of the 128 this kernel is built for)
            uint32_t laneOff = lane * 8;
            asm volatile("" : "+v"(laneOff));
#pragma unroll
            for (int jj = 0; jj < JW; ++jj)
                if (j0 + jj < L) {
                    if (cb == cBegin) scatter_store_site<Q>(acc[jj], row_base(jj, laneOff), rowStrideBytes);
                    else scatter_add_site_f64<Q>(acc[jj], row_base_uniform(jj), laneOff, rowStrideBytes);
                }
        }
    } else {
        run_tiles(cBegin, cEnd);
#pragma unroll
        for (int jj = 0; jj < JW; ++jj)
            if (j0 + jj < L) {
                if constexpr (Q == kPairQ) scatter_store_pair(acc[jj], row_base(jj, lane * 8), rowStrideBytes);
                else scatter_store_site<Q>(acc[jj], row_base(jj, lane * 8), rowStrideBytes);
            }
    }
}

// G[0] += G[1] + ... + G[nsplit-1], fixed order (deterministic).  Used when there are more than two slabs
// (deep, narrow alignments); with a few slabs the fold kernels add them on the fly.
template <typename T>
__global__ void plm_sum_slabs_kernel(T* __restrict__ G, size_t slabElems, int nsplit)
{
    // 16 bytes per lane and load, four slabs' loads in flight before their adds (round 6: config E sums 17 slabs of 2.9 MB --
    // 27 us with one 4-byte load per add, the adds of an element in the same ascending slab order as before)
    using V = typename V16<T>::type;
    constexpr int VEC = 16 / (int)sizeof(T);
    const size_t nv = slabElems / VEC, stride = (size_t)gridDim.x * blockDim.x, t0 = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    auto add = [](V& a, const V& v) {
        a.x += v.x; a.y += v.y;
        if constexpr (sizeof(T) == 4) { a.z += v.z; a.w += v.w; }
    };
    const bool aligned = (slabElems % VEC) == 0 && (reinterpret_cast<uintptr_t>(G) & 15) == 0;
    if (aligned) {
        for (size_t iv = t0; iv < nv; iv += stride) {
            V a = reinterpret_cast<const V*>(G)[iv];
            int sidx = 1;
            for (; sidx + 4 <= nsplit; sidx += 4) {
                const V b0 = reinterpret_cast<const V*>(G + (size_t)sidx * slabElems)[iv];
                const V b1 = reinterpret_cast<const V*>(G + (size_t)(sidx + 1) * slabElems)[iv];
                const V b2 = reinterpret_cast<const V*>(G + (size_t)(sidx + 2) * slabElems)[iv];
                const V b3 = reinterpret_cast<const V*>(G + (size_t)(sidx + 3) * slabElems)[iv];
                add(a, b0); add(a, b1); add(a, b2); add(a, b3);
            }
            for (; sidx < nsplit; ++sidx) add(a, reinterpret_cast<const V*>(G + (size_t)sidx * slabElems)[iv]);
            reinterpret_cast<V*>(G)[iv] = a;
        }
        return;
    }
    for (size_t i = t0; i < slabElems; i += stride) {
        T a = G[i];
        for (int sidx = 1; sidx < nsplit; ++sidx) a += G[(size_t)sidx * slabElems + i];
        G[i] = a;
    }
}

// The same for a column range whose slab count differs from the rest (the left-over strips of the scatter kernel):
// slab 0 receives the sum of slabs 0 .. nsplit-1, slabs 1 .. nzero-1 are cleared so that later sums over them add nothing.
template <typename T>
__global__ void plm_sum_slabs_cols_kernel(T* __restrict__ G, size_t slabElems, int Cs, int col0, int ncols, int rows, int nsplit, int nzero)
{
    const size_t idx = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (idx >= (size_t)rows * ncols) return;
    const size_t off = (idx / ncols) * (size_t)Cs + col0 + idx % ncols;
    T a = G[off];
    for (int sidx = 1; sidx < nsplit; ++sidx) a += G[(size_t)sidx * slabElems + off];
    G[off] = a;
    for (int sidx = 1; sidx < nzero; ++sidx) G[(size_t)sidx * slabElems + off] = (T)0;
}

// ------------------------------------------------------------------ column sums of R (float64 mode)
// g[h_i(a)] needs sum_n R[n][(i,a)].  The float32 path reads it off G (sum over the states of site 0's rows); in float64
// mode -- the parity mode -- it is summed in double-double, i.e. independently of the order, like the objective: the
// oracle compensates the same sums (ORACLE_CANONICAL_F64), so both round the same exact value.  One more pass over R.
constexpr int kColSumRowBlocks = 64;
template <typename T>
__global__ __launch_bounds__(256)
void plm_colsum_parts_kernel(const T* __restrict__ R, int N, int Cs, int Lq, double* __restrict__ parts)
{
    __shared__ double redHi[4][64], redLo[4][64];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + lane;
    const int rpb = (N + gridDim.y - 1) / gridDim.y;
    const int r0 = blockIdx.y * rpb, r1 = min(N, r0 + rpb);
    double hi = 0.0, lo = 0.0;
    if (c < Lq)
        for (int n = r0 + wv; n < r1; n += 4) dd_add(hi, lo, (double)R[(size_t)n * Cs + c]);
    redHi[wv][lane] = hi; redLo[wv][lane] = lo;
    __syncthreads();
    if (wv == 0 && c < Lq) {
        for (int w = 1; w < 4; ++w) dd_add2(hi, lo, redHi[w][lane], redLo[w][lane]);
        parts[2 * ((size_t)blockIdx.y * Lq + c)] = hi;
        parts[2 * ((size_t)blockIdx.y * Lq + c) + 1] = lo;
    }
}
// the softmax kernel's per-chunk partials (q = 5): row block b of the output = the chunks b, b + gridDim.y, ... in that order
__global__ __launch_bounds__(256)
void plm_colsum_chunks_kernel(const double* __restrict__ chunkParts, int numChunks, int Lq, double* __restrict__ parts)
{
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= Lq) return;
    double hi = 0.0, lo = 0.0;
    for (int k = blockIdx.y; k < numChunks; k += gridDim.y) dd_add2(hi, lo, chunkParts[2 * ((size_t)k * Lq + c)], chunkParts[2 * ((size_t)k * Lq + c) + 1]);
    parts[2 * ((size_t)blockIdx.y * Lq + c)] = hi;
    parts[2 * ((size_t)blockIdx.y * Lq + c) + 1] = lo;
}
__global__ void plm_colsum_final_kernel(const double* __restrict__ parts, int nblocks, int Lq, double* __restrict__ colSum)
{
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= Lq) return;
    double hi = 0.0, lo = 0.0;
    for (int b = 0; b < nblocks; ++b) dd_add2(hi, lo, parts[2 * ((size_t)b * Lq + c)], parts[2 * ((size_t)b * Lq + c) + 1]);
    colSum[c] = hi + lo;
}

// ------------------------------------------------------------------ column-strip decomposition: parameter pieces
// pairs (j, i), j in [j0, j1) (sender's sites), i in [i0, i1) (receiver's sites, i0 >= j1), between the packed vector and
// a dense [j][i][q*q] message
template <typename T, bool PACK>
__global__ void strip_pairs_copy_kernel(T* __restrict__ x, T* __restrict__ buf, int L, int q, int j0, int j1, int i0, int i1)
{
    const int ni = i1 - i0;
    const int j = j0 + blockIdx.x / ni, i = i0 + blockIdx.x % ni;
    const int q2 = q * q;
    T* px = x + (size_t)L * q + pair_index(L, j, i) * q2;
    T* pb = buf + (size_t)blockIdx.x * q2;
    for (int t = threadIdx.x; t < q2; t += blockDim.x) {
        if (PACK) pb[t] = px[t];
        else px[t] = pb[t];
    }
}

// ------------------------------------------------------------------ fold
// g[J_ij(a,b)] = 2 lambda_J J + G[(j,b)][(i,a)] + G[(i,a)][(j,b)]   (plmdca_numerics.cpp:541-602:
// the site-i and the site-j conditional both contribute), regulariser value per pair
// (:473-486) as a double partial.
// G arrives as `nsplit` slabs (one per tile-range split of the scatter grid); they are summed here in
// slab order, which is what a separate pass over the slabs would produce.
template <typename T>
__device__ __forceinline__ T slab_sum(const T* __restrict__ G, size_t off, size_t slabElems, int nsplit)
{
    T a = G[off];
    for (int sidx = 1; sidx < nsplit; ++sidx) a += G[(size_t)sidx * slabElems + off];
    return a;
}

// One WAVE per site pair (four pairs per workgroup): a pair is q*q = 441 elements, and with a workgroup per pair the
// small configurations were bound by workgroup dispatch and three dependent global round trips per workgroup
// (config C: 19 900 workgroups, 0.143 ms for 0.35 GB).
constexpr int kFoldWaves = 4;
constexpr int kMaxStripRanks = 64;
// Column-strip decomposition: rank r holds the columns of sites [site0[r], site0[r+1]) and folds the pairs (i, j), i < j,
// whose FIRST site it holds.  Site i's conditional of such a pair lies in its own G; site j's lies in the G of the rank that
// holds j's columns, which has sent its rows of this rank's sites: recv[r'] = (this rank's L q rows) x recvCs[r'] columns.
struct StripMap {
    int rank = 0, world = 1, s0 = 0, s1 = 0;
    int site0[kMaxStripRanks + 1];
    const void* recv[kMaxStripRanks];
    int recvCs[kMaxStripRanks];
};
// g[h_i(a)] = 2 lambda_h h + sum_n R[n][(i,a)]; the column sum of R is the sum over b of
// any site's rows of G (site 0 here).  (:463-471, :538-539, :573-578)
// colSum (float64 mode): the column sums of R summed order-independently by plm_colsum_* below; else they are taken
// from G as described above.
template <typename T>
__device__ __forceinline__ void fold_fields_body(const T* __restrict__ x, const T* __restrict__ G, T* __restrict__ g,
                                                 double* __restrict__ regPart, int Lq, int q, int Cs, T lambdaH, int addReg,
                                                 size_t slabElems, int nsplit, const double* __restrict__ colSum, int blk)
{
    __shared__ double red[256];
    const int c = blk * blockDim.x + threadIdx.x;
    double reg = 0.0;
    if (c < Lq) {
        const T xv = x[c];
        T gv = addReg ? (T)2 * lambdaH * xv : (T)0;
        T s = 0;
        if (colSum) s = (T)colSum[c];
        else for (int b = 0; b < q; ++b) s += slab_sum(G, (size_t)b * Cs + c, slabElems, nsplit);
        g[c] = gv + s;
        if (addReg) reg = (double)lambdaH * (double)xv * (double)xv;
    }
    __shared__ double redLo[256];
    red[threadIdx.x] = reg;
    redLo[threadIdx.x] = 0.0;
    __syncthreads();
    for (int s = blockDim.x / 2; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) dd_add2(red[threadIdx.x], redLo[threadIdx.x], red[threadIdx.x + s], redLo[threadIdx.x + s]);
        __syncthreads();
    }
    if (threadIdx.x == 0) { regPart[2 * (size_t)blk] = red[0]; regPart[2 * (size_t)blk + 1] = redLo[0]; }
}
template <typename T>
__global__ void plm_fold_fields_kernel(const T* __restrict__ x, const T* __restrict__ G, T* __restrict__ g,
                                       double* __restrict__ regPart, int Lq, int q, int Cs, T lambdaH, int addReg,
                                       size_t slabElems, int nsplit, const double* __restrict__ colSum)
{
    fold_fields_body<T>(x, G, g, regPart, Lq, q, Cs, lambdaH, addReg, slabElems, nsplit, colSum, (int)blockIdx.x);
}
// what the pair fold carries behind its own workgroups when the fields ride in its launch (one GPU: round 6)
template <typename T> struct FoldFieldsArgs { const T* x; T* g; double* regPart; int Lq; T lambdaH; const double* colSum; int pairBlocks; };
template <typename T>
__global__ __launch_bounds__(64 * kFoldWaves)
void plm_fold_pairs_kernel(const T* __restrict__ x, const T* __restrict__ G, T* __restrict__ g,
                           const PairIJ* __restrict__ pairs, double* __restrict__ regPart,
                           int L, int q, int Cs, T lambdaJ, int addReg, size_t slabElems, int nsplit, int pairBegin, int pairEnd,
                           const StripMap sm, const FoldFieldsArgs<T> ff)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char dca_smem[];
    if (ff.pairBlocks >= 0 && (int)blockIdx.x >= ff.pairBlocks) {      // the field fold's workgroups, behind the pairs'
        fold_fields_body<T>(ff.x, G, ff.g, ff.regPart, ff.Lq, q, Cs, ff.lambdaH, addReg, slabElems, nsplit, ff.colSum, (int)blockIdx.x - ff.pairBlocks);
        return;
    }
    const int q2 = q * q;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    T* tile = reinterpret_cast<T*>(dca_smem) + (size_t)wave * ((q2 + 3) / 4 * 4);     // G[(j,b)][(i,a)] stored as tile[b*q+a]
    const int p = pairBegin + blockIdx.x * kFoldWaves + wave;
    if (p >= pairEnd) return;                                  // wave-uniform; no workgroup barriers below
    const int i = pairs[p].i, j = pairs[p].j;
    const int ic = (i - sm.s0) * q;                            // site i's first column in this rank's window
    for (int t = lane; t < q2; t += 64) {
        const int b = t / q, a = t % q;
        tile[t] = slab_sum(G, (size_t)(j * q + b) * Cs + ic + a, slabElems, nsplit);
    }
    // site j's conditional: this rank's G when it holds j's columns too, else the rows its holder has sent
    const T* Gj = G;
    size_t jRow = (size_t)i * q, jCs = (size_t)Cs;
    int jc = (j - sm.s0) * q, jSplit = nsplit;
    if (j >= sm.s1) {
        int r = sm.rank + 1;
        while (j >= sm.site0[r + 1]) ++r;
        Gj = static_cast<const T*>(sm.recv[r]);
        jRow = (size_t)(i - sm.s0) * q; jCs = (size_t)sm.recvCs[r]; jc = (j - sm.site0[r]) * q; jSplit = 1;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    const size_t base = (size_t)L * q + (size_t)p * q2;
    double reg = 0.0, regLo = 0.0;
    for (int t = lane; t < q2; t += 64) {
        const int a = t / q, b = t % q;
        const T xv = x[base + t];
        // (2 lambda x + site i's conditional) + site j's conditional: the order of the reference's one-thread merge
        // (plmdca_numerics.cpp:570-602 in ascending site order) and of the oracle
        T gv = addReg ? (T)2 * lambdaJ * xv : (T)0;
        gv += tile[b * q + a];                                                            // G[(j,b)][(i,a)]: column of site i
        gv += slab_sum(Gj, (jRow + a) * jCs + jc + b, slabElems, jSplit);                 // G[(i,a)][(j,b)]: column of site j
        g[base + t] = gv;
        if (addReg) dd_add(reg, regLo, (double)lambdaJ * (double)xv * (double)xv);
    }
    dd_wave_reduce(reg, regLo);                                               // fixed tree
    if (lane == 0) { regPart[2 * (size_t)p] = reg; regPart[2 * (size_t)p + 1] = regLo; }
}

// ------------------------------------------------------------------ L-BFGS vector kernels
constexpr int kVecBlocks = 1024;
constexpr int kVecThreads = 256;

// All vector kernels stream 16 bytes per lane and load (4 floats / 2 doubles): with 4-byte loads a
// wave has too few bytes in flight to approach HBM bandwidth.  Element k of a pack is element
// iv*VEC + k; the n % VEC tail is handled by the first threads with scalar accesses.  The mapping
// of elements to threads is fixed, so every reduction is deterministic.
template <typename T> struct Pack { T v[16 / sizeof(T)]; };
template <typename T> __device__ __forceinline__ Pack<T> ldp_at(const T* p, size_t iv)
{
    using V = typename V16<T>::type;
    const V r = reinterpret_cast<const V*>(p)[iv];
    Pack<T> o;
    if constexpr (sizeof(T) == 4) { o.v[0] = r.x; o.v[1] = r.y; o.v[2] = r.z; o.v[3] = r.w; }
    else { o.v[0] = r.x; o.v[1] = r.y; }
    return o;
}
template <typename T> __device__ __forceinline__ void stp_at(T* p, size_t iv, const Pack<T>& o)
{
    using V = typename V16<T>::type;
    V r;
    if constexpr (sizeof(T) == 4) { r.x = o.v[0]; r.y = o.v[1]; r.z = o.v[2]; r.w = o.v[3]; }
    else { r.x = o.v[0]; r.y = o.v[1]; }
    reinterpret_cast<V*>(p)[iv] = r;
}
// The vectors of one call all start at the SAME element offset of 256-byte aligned allocations (base + vlo), so they share
// their misalignment.  With sequence sharding vlo is a multiple of four elements (set_slices); with column strips (exchange
// mode 4) it is the start of the rank's pair range, L q + pairs q^2 -- any parity.  ALIGNP names one of the vectors: the
// head_ elements in front of its first 16-byte boundary are handled with the tail, one element per thread, and the packs
// start at that boundary (ldp / stp inside the loop body index from there), so every 16-byte access is aligned.
#define ldp(p, iv) ldp_at((p) + head_, iv)
#define stp(p, iv, o) stp_at((p) + head_, iv, o)
#define DCA_VEC_LOOP(n, ALIGNP, BODY_PACK, BODY_TAIL) DCA_VEC_LOOP_G(n, ALIGNP, gridDim.x, BODY_PACK, BODY_TAIL)
/* GRID: the number of workgroups that walk the vector (a launch may carry other workgroups behind them) */
#define DCA_VEC_LOOP_G(n, ALIGNP, GRID, BODY_PACK, BODY_TAIL)                                             \
    {                                                                                                      \
        constexpr int VEC = 16 / (int)sizeof(T);                                                           \
        const size_t lead_ = ((16 - (reinterpret_cast<uintptr_t>(ALIGNP) & 15)) & 15) / sizeof(T);         \
        const size_t head_ = lead_ < (size_t)(n) ? lead_ : (size_t)(n);                                    \
        const size_t nv_ = ((n) - head_) / VEC, stride_ = (size_t)(GRID) * blockDim.x;                     \
        const size_t t0_ = blockIdx.x * (size_t)blockDim.x + threadIdx.x;                                  \
        for (size_t iv = t0_; iv < nv_; iv += stride_) { BODY_PACK }                                       \
        const size_t rest_ = (n) - nv_ * VEC;                     /* head_ + tail, fewer than 2 VEC */      \
        for (size_t r_ = t0_; r_ < rest_; r_ += stride_) {                                                 \
            const size_t i = r_ < head_ ? r_ : r_ + nv_ * VEC;                                             \
            BODY_TAIL                                                                                      \
        }                                                                                                  \
    }

template <typename T>
__global__ void vec_neg_kernel(T* __restrict__ d, const T* __restrict__ g, size_t n)
{
    DCA_VEC_LOOP(n, d,
        Pack<T> a = ldp(g, iv);
        _Pragma("unroll") for (int k = 0; k < VEC; ++k) a.v[k] = -a.v[k];
        stp(d, iv, a);,
        d[i] = -g[i];)
}
template <typename T>
__global__ void vec_axpy_kernel(T* __restrict__ y, T a, const T* __restrict__ x, size_t n)
{
    DCA_VEC_LOOP(n, y,
        Pack<T> yy = ldp(y, iv); const Pack<T> xx = ldp(x, iv);
        _Pragma("unroll") for (int k = 0; k < VEC; ++k) yy.v[k] += a * xx.v[k];
        stp(y, iv, yy);,
        y[i] += a * x[i];)
}
template <typename T>
__global__ void vec_scale_kernel(T* __restrict__ y, T a, size_t n)
{
    DCA_VEC_LOOP(n, y,
        Pack<T> yy = ldp(y, iv);
        _Pragma("unroll") for (int k = 0; k < VEC; ++k) yy.v[k] *= a;
        stp(y, iv, yy);,
        y[i] *= a;)
}
// x = xp + stp*d, as lbfgs.cpp:902-903 (copy, then add the rounded product)
template <typename T>
__global__ void vec_step_kernel(T* __restrict__ x, const T* __restrict__ xp, T stpv, const T* __restrict__ d, size_t n)
{
    DCA_VEC_LOOP(n, x,
        const Pack<T> dd = ldp(d, iv); Pack<T> xx = ldp(xp, iv);
        _Pragma("unroll") for (int k = 0; k < VEC; ++k) { const T v = stpv * dd.v[k]; xx.v[k] = xx.v[k] + v; }
        stp(x, iv, xx);,
        { const T v = stpv * d[i]; x[i] = xp[i] + v; })
}

__device__ __forceinline__ void block_reduce_store(double v, double* red, double* out)
{
    red[threadIdx.x] = v;
    __syncthreads();
    for (int s = blockDim.x / 2; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
        __syncthreads();
    }
    if (threadIdx.x == 0) *out = red[0];
    __syncthreads();
}

// Dot-product accumulators of the optimiser.  float32 vectors: products and sums in double (already far more exact than the
// reference's float sums).  float64 vectors (the parity mode): the ROUNDED products are summed in double-double, like the
// objective -- a plain double sum of P = 5.5e7 products carries ~1e-13 of order-dependent rounding, g.d / y.s / y.y and the
// Gram entries steer the line search and scale the direction, and the optimisation amplifies such noise from iteration to
// iteration.  The float64 oracle compensates the same sums (Neumaier), so both see the sum of the
// same rounded products to the last bit or two, whatever the order.  Partials travel as (hi, lo) pairs in both cases.
template <bool DD> struct DotAcc;
template <> struct DotAcc<false> {
    double hi = 0.0;
    static constexpr double lo = 0.0;
    __device__ __forceinline__ void add(double a, double b) { hi += a * b; }
    __device__ __forceinline__ void wave_reduce() { for (int off = 32; off > 0; off >>= 1) hi += __shfl_down(hi, off); }
};
template <> struct DotAcc<true> {
    double hi = 0.0, lo = 0.0;
    __device__ __forceinline__ void add(double a, double b) { dd_add(hi, lo, __dmul_rn(a, b)); }
    __device__ __forceinline__ void wave_reduce() { dd_wave_reduce(hi, lo); }
};
// the workgroup's waves leave their (hi, lo) in red[wave][2 * v], [2 * v + 1]; thread v < nv adds them in wave order
template <int NV>
__device__ __forceinline__ void dot_block_store(double (*red)[2 * NV], int nv, double* __restrict__ partials, unsigned grid = 0)
{
    __syncthreads();
    if ((int)threadIdx.x < nv) {
        double hi = 0.0, lo = 0.0;
        for (int w = 0; w < (int)blockDim.x / 64; ++w) dd_add2(hi, lo, red[w][2 * threadIdx.x], red[w][2 * threadIdx.x + 1]);
        const size_t slot = (size_t)threadIdx.x * (grid ? grid : gridDim.x) + blockIdx.x;
        partials[2 * slot] = hi;
        partials[2 * slot + 1] = lo;
    }
}

// (hi, lo) partials [2 * (k*gridDim.x + block)] for k = 0..2 : a.b, c.c, a.a   (g.d, x.x, g.g)
template <typename T>
__global__ __launch_bounds__(kVecThreads)
void vec_dot3_kernel(const T* __restrict__ a, const T* __restrict__ b, const T* __restrict__ c, size_t n,
                     double* __restrict__ partials)
{
    __shared__ double red[kVecThreads / 64][6];
    DotAcc<sizeof(T) == 8> s0, s1, s2;
    DCA_VEC_LOOP(n, a,
        const Pack<T> pa = ldp(a, iv); const Pack<T> pb = ldp(b, iv); const Pack<T> pc = ldp(c, iv);
        _Pragma("unroll") for (int k = 0; k < VEC; ++k) {
            const double av = pa.v[k]; const double bv = pb.v[k]; const double cv = pc.v[k];
            s0.add(av, bv); s1.add(cv, cv); s2.add(av, av);
        },
        { const double av = a[i]; const double bv = b[i]; const double cv = c[i]; s0.add(av, bv); s1.add(cv, cv); s2.add(av, av); })
    s0.wave_reduce(); s1.wave_reduce(); s2.wave_reduce();
    if ((threadIdx.x & 63) == 0) {
        double* r = red[threadIdx.x >> 6];
        r[0] = s0.hi; r[1] = s0.lo; r[2] = s1.hi; r[3] = s1.lo; r[4] = s2.hi; r[5] = s2.lo;
    }
    dot_block_store<3>(red, 3, partials);
}
template <typename T>
__global__ __launch_bounds__(kVecThreads)
void vec_dot_kernel(const T* __restrict__ a, const T* __restrict__ b, size_t n, double* __restrict__ partials)
{
    __shared__ double red[kVecThreads / 64][2];
    DotAcc<sizeof(T) == 8> s0;
    DCA_VEC_LOOP(n, a,
        const Pack<T> pa = ldp(a, iv); const Pack<T> pb = ldp(b, iv);
        _Pragma("unroll") for (int k = 0; k < VEC; ++k) s0.add((double)pa.v[k], (double)pb.v[k]);,
        s0.add((double)a[i], (double)b[i]);)
    s0.wave_reduce();
    if ((threadIdx.x & 63) == 0) { red[threadIdx.x >> 6][0] = s0.hi; red[threadIdx.x >> 6][1] = s0.lo; }
    dot_block_store<1>(red, 1, partials);
}
// L-BFGS direction in one pass instead of 2m dependent dot/axpy rounds: every vector of the
// two-loop recursion (lbfgs.cpp:568-601) lies in span{g, s_k, y_k}, so the recursion can be run
// on 2m+1 coefficients once the Gram entries it needs are known: for the newest pair e and every
// slot k: s_k.g, y_k.g, s_e.y_k, y_e.s_k, y_e.y_k (25 dot products; the entries between older pairs are
// kept from earlier iterations).  vec_diff_gram_kernel below produces them in the pass that forms the pair.
struct VecPtrs5 { const void* s[5]; const void* y[5]; };
struct DirCoefs { double g; double s[5]; double y[5]; };

// Optimiser scalars that live on the device: dot products of the stored pairs and the coefficients of the current
// search direction in {g, s_k, y_k}.  The two-loop recursion runs here (one thread), so an iteration needs ONE host
// round trip -- the line search's decision after an evaluation -- instead of two.
struct LbfgsDev { double SY[5][5]; double YY[5][5]; double ys[5]; DirCoefs cf; };
constexpr int kSlotDginit = 30;      // dScal slot of g.d for the next line search

// lbfgs.cpp:568-601 on the coefficients; scal[1..2] = y.s, y.y of the newest pair e, scal[3..27] the 25 Gram entries
// [kind * 5 + k]: s_k.g, y_k.g, s_e.y_k, s_k.y_e, y_e.y_k; gg = g.g of the accepted point (the host has it).
__global__ void lbfgs_two_loop_kernel(double* __restrict__ scal, LbfgsDev* __restrict__ st, int e, int endNext, int bound, double gg)
{
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    constexpr int M = 5;
    const double ys = scal[1], yy = scal[2];
    const double* G5 = scal + 3;
    double Sg[M], Yg[M], alpha[M];
    st->ys[e] = ys;
    for (int k2 = 0; k2 < M; ++k2) {
        Sg[k2] = G5[k2]; Yg[k2] = G5[5 + k2];
        st->SY[e][k2] = G5[10 + k2];            // s_e . y_k
        st->SY[k2][e] = G5[15 + k2];            // s_k . y_e
        st->YY[e][k2] = G5[20 + k2];
        st->YY[k2][e] = G5[20 + k2];
        alpha[k2] = 0.0;
    }
    st->SY[e][e] = ys; st->YY[e][e] = yy;
    DirCoefs cf;
    cf.g = -1.0;
    for (int k2 = 0; k2 < M; ++k2) cf.s[k2] = cf.y[k2] = 0.0;
    int j = endNext;
    for (int i = 0; i < bound; ++i) {
        j = (j + M - 1) % M;
        double sd = cf.g * Sg[j];
        for (int k2 = 0; k2 < M; ++k2) sd += cf.y[k2] * st->SY[j][k2];   // d has no s components yet
        alpha[j] = sd / st->ys[j];
        cf.y[j] -= alpha[j];
    }
    const double scale = ys / yy;
    cf.g *= scale;
    for (int k2 = 0; k2 < M; ++k2) cf.y[k2] *= scale;
    for (int i = 0; i < bound; ++i) {
        double yd = cf.g * Yg[j];
        for (int k2 = 0; k2 < M; ++k2) yd += cf.s[k2] * st->SY[k2][j] + cf.y[k2] * st->YY[j][k2];
        const double beta = yd / st->ys[j];
        cf.s[j] += alpha[j] - beta;
        j = (j + 1) % M;
    }
    st->cf = cf;
    // g.d for the next line search, from the same coefficients
    double gd = cf.g * gg;
    for (int k2 = 0; k2 < M; ++k2) gd += cf.s[k2] * Sg[k2] + cf.y[k2] * Yg[k2];
    scal[kSlotDginit] = gd;
}

// s_e = x - xp, y_e = g - gp (lbfgs.cpp:546-558) are formed, stored and used in one pass, so the newest pair is not
// read back and g is read once (14 vector passes; 19 as two kernels, 0.75 -> 0.6 ms at D).  partials[v * gridDim.x +
// block]: v = 0, 1 are y_e.s_e and y_e.y_e, v = 2 + kind * 5 + k the Gram entries (kinds in the order above).
template <typename T, int E>
__global__ __launch_bounds__(kVecThreads)
void vec_diff_gram_kernel(VecPtrs5 P, T* __restrict__ se, T* __restrict__ ye, const T* __restrict__ x, const T* __restrict__ xp,
                          const T* __restrict__ g, const T* __restrict__ gp, size_t n, double* __restrict__ partials)
{
    // E = slot of the newest pair, a template parameter: as a run-time value the test `k != e` stood in front of every pair's two
    // loads, which the compiler then issued and WAITED for pair by pair -- six round trips per pack with two to four loads in
    // flight (config D: 3.7 TB/s where the other vector kernels reach 5.3 - 6.4).  All twelve loads of a pack are issued before
    // the first store (the stores may alias the history for all the compiler knows).
    __shared__ double red[kVecThreads / 64][54];
    DotAcc<sizeof(T) == 8> acc[27];
    DCA_VEC_LOOP(n, se,
        const Pack<T> px = ldp(x, iv); const Pack<T> pxp = ldp(xp, iv); const Pack<T> pg = ldp(g, iv); const Pack<T> pgp = ldp(gp, iv);
        Pack<T> psk[5]; Pack<T> pyk[5];
        _Pragma("unroll") for (int k = 0; k < 5; ++k)
            if (k != E) { psk[k] = ldp(static_cast<const T*>(P.s[k]), iv); pyk[k] = ldp(static_cast<const T*>(P.y[k]), iv); }
        Pack<T> pse; Pack<T> pye;
        _Pragma("unroll") for (int u = 0; u < VEC; ++u) {
            pse.v[u] = px.v[u] - pxp.v[u]; pye.v[u] = pg.v[u] - pgp.v[u];
            acc[0].add((double)pye.v[u], (double)pse.v[u]); acc[1].add((double)pye.v[u], (double)pye.v[u]);
        }
        psk[E] = pse; pyk[E] = pye;
        stp(se, iv, pse); stp(ye, iv, pye);
        _Pragma("unroll") for (int k = 0; k < 5; ++k) {
            _Pragma("unroll") for (int u = 0; u < VEC; ++u) {
                const double gv = pg.v[u]; const double sev = pse.v[u]; const double yev = pye.v[u];
                const double sk = psk[k].v[u]; const double yk = pyk[k].v[u];
                acc[2 + k].add(sk, gv); acc[7 + k].add(yk, gv); acc[12 + k].add(sev, yk); acc[17 + k].add(yev, sk); acc[22 + k].add(yev, yk);
            }
        },
        { const T sev_ = x[i] - xp[i]; const T yev_ = g[i] - gp[i]; se[i] = sev_; ye[i] = yev_;
          const double gv = g[i]; const double sev = sev_; const double yev = yev_;
          acc[0].add(yev, sev); acc[1].add(yev, yev);
          _Pragma("unroll") for (int k = 0; k < 5; ++k) {
              const double sk = k == E ? sev : (double)static_cast<const T*>(P.s[k])[i]; const double yk = k == E ? yev : (double)static_cast<const T*>(P.y[k])[i];
              acc[2 + k].add(sk, gv); acc[7 + k].add(yk, gv); acc[12 + k].add(sev, yk); acc[17 + k].add(yev, sk); acc[22 + k].add(yev, yk);
          } })
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
#pragma unroll
    for (int v = 0; v < 27; ++v) {
        acc[v].wave_reduce();
        if (lane == 0) { red[wv][2 * v] = acc[v].hi; red[wv][2 * v + 1] = acc[v].lo; }
    }
    dot_block_store<27>(red, 27, partials);
}

// d = c.g * g + sum_k c.s[k] * s_k + c.y[k] * y_k
template <typename T>
__global__ void vec_compose_kernel(T* __restrict__ d, const T* __restrict__ g, VecPtrs5 P, const DirCoefs* __restrict__ cp, size_t n)
{
    const DirCoefs c = *cp;
    DCA_VEC_LOOP(n, d,
        const Pack<T> pg = ldp(g, iv);
        double v[VEC];
        _Pragma("unroll") for (int u = 0; u < VEC; ++u) v[u] = c.g * (double)pg.v[u];
        _Pragma("unroll") for (int k = 0; k < 5; ++k) {
            const Pack<T> psk = ldp(static_cast<const T*>(P.s[k]), iv); const Pack<T> pyk = ldp(static_cast<const T*>(P.y[k]), iv);
            _Pragma("unroll") for (int u = 0; u < VEC; ++u) v[u] += c.s[k] * (double)psk.v[u] + c.y[k] * (double)pyk.v[u];
        }
        Pack<T> o;
        _Pragma("unroll") for (int u = 0; u < VEC; ++u) o.v[u] = (T)v[u];
        stp(d, iv, o);,
        { double v = c.g * (double)g[i];
          _Pragma("unroll") for (int k = 0; k < 5; ++k)
              v += c.s[k] * (double)static_cast<const T*>(P.s[k])[i] + c.y[k] * (double)static_cast<const T*>(P.y[k])[i];
          d[i] = (T)v; })
}
#undef ldp
#undef stp

// out[k] = sum_b of the (hi, lo) pairs partials[2 * (k*nb + b)], k < nk, rounded once; one block per k, fixed tree
__global__ __launch_bounds__(256)
void vec_final_kernel(const double* __restrict__ partials, int nb, int nk, double* __restrict__ out)
{
    __shared__ double redHi[256], redLo[256];
    const int k = blockIdx.x;
    double hi = 0.0, lo = 0.0;
    for (int b = threadIdx.x; b < nb; b += blockDim.x) dd_add2(hi, lo, partials[2 * ((size_t)k * nb + b)], partials[2 * ((size_t)k * nb + b) + 1]);
    dd_block_reduce(hi, lo, redHi, redLo);
    if (threadIdx.x == 0) out[k] = hi + lo;
}
// out[0] = (add ? out[0] : 0) + sum partials[0..n)
__global__ void sum_partials_kernel(const double* __restrict__ partials, int n, double* __restrict__ out, int add)
{
    __shared__ double red[1024];
    double s = 0;
    for (int b = threadIdx.x; b < n; b += blockDim.x) s += partials[b];
    red[threadIdx.x] = s;
    __syncthreads();
    for (int st = blockDim.x / 2; st > 0; st >>= 1) {
        if ((int)threadIdx.x < st) red[threadIdx.x] += red[threadIdx.x + st];
        __syncthreads();
    }
    if (threadIdx.x == 0) out[0] = (add ? out[0] : 0.0) + red[0];
}

// ---- the same for (hi, lo) pairs (the objective's partial sums)
// block b sums its contiguous chunk of the n pairs into pair b of out
__global__ __launch_bounds__(256)
void dd_sum_chunks_kernel(const double* __restrict__ parts, int n, double* __restrict__ out)
{
    __shared__ double redHi[256], redLo[256];
    const int chunk = (n + gridDim.x - 1) / gridDim.x;
    const int lo_ = blockIdx.x * chunk, hi_ = min(n, lo_ + chunk);
    double hi = 0.0, lo = 0.0;
    for (int b = lo_ + threadIdx.x; b < hi_; b += blockDim.x) dd_add2(hi, lo, parts[2 * (size_t)b], parts[2 * (size_t)b + 1]);
    dd_block_reduce(hi, lo, redHi, redLo);
    if (threadIdx.x == 0) { out[2 * blockIdx.x] = hi; out[2 * blockIdx.x + 1] = lo; }
}
// out[0] = the sum of the nA pairs of A and the nB pairs of B, rounded once
__global__ __launch_bounds__(1024)
void dd_sum_final_kernel(const double* __restrict__ A, int nA, const double* __restrict__ B, int nB, double* __restrict__ out)
{
    __shared__ double redHi[1024], redLo[1024];
    double hi = 0.0, lo = 0.0;
    for (int b = threadIdx.x; b < nA; b += blockDim.x) dd_add2(hi, lo, A[2 * (size_t)b], A[2 * (size_t)b + 1]);
    for (int b = threadIdx.x; b < nB; b += blockDim.x) dd_add2(hi, lo, B[2 * (size_t)b], B[2 * (size_t)b + 1]);
    dd_block_reduce(hi, lo, redHi, redLo);
    if (threadIdx.x == 0) out[0] = hi + lo;
}

// first stage for long partial vectors: block b sums its contiguous chunk (fixed tree) into out[b]
constexpr int kSumStageBlocks = 64;

// The two reductions that end an evaluation of the optimiser -- fx from its per-pair / per-chunk partial sums (dd_sum_chunks_kernel,
// dd_sum_final_kernel) and the three dot products of the line search (vec_dot3_kernel, vec_final_kernel) -- as TWO launches instead
// of four: the workgroups behind the first kVecBlocks of the first launch sum the fx chunks, the workgroup behind the dot products'
// of the second finishes fx.  Every sum is formed by the same code over the same operands in the same order as in the separate
// kernels (the vector walk with its grid given, the 256-thread tree inside the 1024-thread workgroups), so the bits are theirs; what
// goes is two launch boundaries and ~11 us of two tiny kernels per evaluation (config C: 1.6 % of the step).
template <typename T>
__global__ __launch_bounds__(kVecThreads)
void vec_dot3_fx_kernel(const T* __restrict__ a, const T* __restrict__ b, const T* __restrict__ c, size_t n, double* __restrict__ partials,
                        const double* __restrict__ fxParts, int nFxParts, double* __restrict__ fxChunks)
{
    if (blockIdx.x >= (unsigned)kVecBlocks) {
        __shared__ double redHi[256], redLo[256];
        const int blk = (int)blockIdx.x - kVecBlocks;
        const int chunk = (nFxParts + kSumStageBlocks - 1) / kSumStageBlocks;
        const int lo_ = blk * chunk, hi_ = min(nFxParts, lo_ + chunk);
        double hi = 0.0, lo = 0.0;
        for (int p = lo_ + threadIdx.x; p < hi_; p += blockDim.x) dd_add2(hi, lo, fxParts[2 * (size_t)p], fxParts[2 * (size_t)p + 1]);
        dd_block_reduce(hi, lo, redHi, redLo);
        if (threadIdx.x == 0) { fxChunks[2 * blk] = hi; fxChunks[2 * blk + 1] = lo; }
        return;
    }
    __shared__ double red[kVecThreads / 64][6];
    DotAcc<sizeof(T) == 8> s0, s1, s2;
    DCA_VEC_LOOP_G(n, a, kVecBlocks,
        const Pack<T> pa = ldp_at(a + head_, iv); const Pack<T> pb = ldp_at(b + head_, iv); const Pack<T> pc = ldp_at(c + head_, iv);
        _Pragma("unroll") for (int k = 0; k < VEC; ++k) {
            const double av = pa.v[k]; const double bv = pb.v[k]; const double cv = pc.v[k];
            s0.add(av, bv); s1.add(cv, cv); s2.add(av, av);
        },
        { const double av = a[i]; const double bv = b[i]; const double cv = c[i]; s0.add(av, bv); s1.add(cv, cv); s2.add(av, av); })
    s0.wave_reduce(); s1.wave_reduce(); s2.wave_reduce();
    if ((threadIdx.x & 63) == 0) {
        double* r = red[threadIdx.x >> 6];
        r[0] = s0.hi; r[1] = s0.lo; r[2] = s1.hi; r[3] = s1.lo; r[4] = s2.hi; r[5] = s2.lo;
    }
    dot_block_store<3>(red, 3, partials, kVecBlocks);
}
// 1024 threads per workgroup.  Workgroups 0 .. nk - 1: vec_final_kernel's sum of dot product k with its 256 threads (the others only
// keep the barriers company); workgroup nk: dd_sum_final_kernel's sum of fx with all 1024.
__global__ __launch_bounds__(1024)
void vec_final_fx_kernel(const double* __restrict__ partials, int nb, int nk, double* __restrict__ out,
                         const double* __restrict__ A, int nA, const double* __restrict__ B, int nB, double* __restrict__ fxOut)
{
    __shared__ double redHi[1024], redLo[1024];
    const int k = blockIdx.x;
    double hi = 0.0, lo = 0.0;
    if (k < nk) {
        constexpr int NT = 256;
        if ((int)threadIdx.x < NT) {
            for (int b = threadIdx.x; b < nb; b += NT) dd_add2(hi, lo, partials[2 * ((size_t)k * nb + b)], partials[2 * ((size_t)k * nb + b) + 1]);
            redHi[threadIdx.x] = hi;
            redLo[threadIdx.x] = lo;
        }
        __syncthreads();
        for (int st = NT / 2; st > 0; st >>= 1) {
            if ((int)threadIdx.x < st) dd_add2(redHi[threadIdx.x], redLo[threadIdx.x], redHi[threadIdx.x + st], redLo[threadIdx.x + st]);
            __syncthreads();
        }
        if (threadIdx.x == 0) out[k] = redHi[0] + redLo[0];
        return;
    }
    for (int b = threadIdx.x; b < nA; b += blockDim.x) dd_add2(hi, lo, A[2 * (size_t)b], A[2 * (size_t)b + 1]);
    for (int b = threadIdx.x; b < nB; b += blockDim.x) dd_add2(hi, lo, B[2 * (size_t)b], B[2 * (size_t)b + 1]);
    dd_block_reduce(hi, lo, redHi, redLo);
    if (threadIdx.x == 0) fxOut[0] = hi + lo;
}
__global__ void sum_chunks_kernel(const double* __restrict__ partials, int n, double* __restrict__ out)
{
    __shared__ double red[256];
    const int chunk = (n + gridDim.x - 1) / gridDim.x;
    const int lo = blockIdx.x * chunk, hi = min(n, lo + chunk);
    double s = 0;
    for (int b = lo + threadIdx.x; b < hi; b += blockDim.x) s += partials[b];
    block_reduce_store(s, red, out + blockIdx.x);
}

#ifdef DCA_ROUND_ABLATE
// ANALYSIS BUILD ONLY (make ablate -> lib/libdca_hip_ablate.so; the shipped library has no such switch): the float64 engine
// rounds the output of selected stages to float32, DCA_ROUND_F32_STAGES = bit mask (1 W, 2 S, 4 R, 8 G, 16 g, 32 x, 64 d).
// Rounding a stage's OUTPUT is a lower bound on what computing that stage in float32 would do to the result; used to
// decide which mixed-precision pipelines can keep protocol P3 (tests/analysis/mixed_precision_table.py).
__global__ void round_to_f32_kernel(double* __restrict__ p, size_t n)
{
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = (double)(float)p[i];
}
inline int round_stage_mask() { const char* e = getenv("DCA_ROUND_F32_STAGES"); return e ? atoi(e) : 0; }
#define DCA_ROUND_STAGE(bit, ptr, count)                                                                             \
    do {                                                                                                             \
        if constexpr (sizeof(T) == 8)                                                                                \
            if (round_stage_mask() & (bit))                                                                          \
                hipLaunchKernelGGL(round_to_f32_kernel, dim3(2048), dim3(256), 0, ctx->stream, reinterpret_cast<double*>(ptr), (size_t)(count)); \
    } while (0)
#else
#define DCA_ROUND_STAGE(bit, ptr, count) do { } while (0)
#endif

template <typename T>
__global__ void cast_weights_kernel(const double* __restrict__ wd, T* __restrict__ w, int N)
{
    const int n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n < N) w[n] = (T)wd[n];
}

// ======================================================================
// More-Thuente trial-interval update, restated from More & Thuente (1994) with the
// safeguards the reference's library applies (lbfgs.cpp:1128-1295).  Scalars are double.
struct LsPoint { double st, f, d; };

double cubic_min(double u, double fu, double du, double v, double fv, double dv)
{
    const double d = v - u;
    const double theta = (fu - fv) * 3 / d + du + dv;
    const double s = std::max(std::fabs(theta), std::max(std::fabs(du), std::fabs(dv)));
    const double a = theta / s;
    double gamma = s * std::sqrt(a * a - (du / s) * (dv / s));
    if (v < u) gamma = -gamma;
    const double p = gamma - du + theta, q = gamma - du + gamma + dv;
    return u + (p / q) * d;
}
double cubic_min_clamped(double u, double fu, double du, double v, double fv, double dv, double lo, double hi)
{
    const double d = v - u;
    const double theta = (fu - fv) * 3 / d + du + dv;
    const double s = std::max(std::fabs(theta), std::max(std::fabs(du), std::fabs(dv)));
    const double a = theta / s;
    double gamma = s * std::sqrt(std::max(0.0, a * a - (du / s) * (dv / s)));
    if (u < v) gamma = -gamma;
    const double p = gamma - dv + theta, q = gamma - dv + gamma + du;
    const double r = p / q;
    if (r < 0. && gamma != 0.) return v - r * d;
    return a < 0 ? hi : lo;
}
double quad_min_f(double u, double fu, double du, double v, double fv)
{
    const double a = v - u;
    return u + du / ((fu - fv) / a + du) / 2 * a;
}
double quad_min_d(double u, double du, double v, double dv)
{
    const double a = u - v;
    return v + dv / (dv - du) * a;
}

enum {
    LB_OUTOFINTERVAL = -1003, LB_INCORRECT_TMINMAX = -1002, LB_ROUNDING_ERROR = -1001, LB_MINIMUMSTEP = -1000,
    LB_MAXIMUMSTEP = -999, LB_MAXIMUMLINESEARCH = -998, LB_MAXIMUMITERATION = -997, LB_WIDTHTOOSMALL = -996,
    LB_INVALIDPARAMETERS = -995, LB_INCREASEGRADIENT = -994, LB_ALREADY_MINIMIZED = 2
};

int mt_update(LsPoint& best, LsPoint& other, double& t, double ft, double dt, double tmin, double tmax, bool& brackt)
{
    const bool opposite = (dt * (best.d / std::fabs(best.d)) < 0.);
    bool bound;
    double newt;
    if (brackt) {
        if (t <= std::min(best.st, other.st) || std::max(best.st, other.st) <= t) return LB_OUTOFINTERVAL;
        if (0. <= best.d * (t - best.st)) return LB_INCREASEGRADIENT;
        if (tmax < tmin) return LB_INCORRECT_TMINMAX;
    }
    if (best.f < ft) {
        brackt = true; bound = true;
        const double mc = cubic_min(best.st, best.f, best.d, t, ft, dt);
        const double mq = quad_min_f(best.st, best.f, best.d, t, ft);
        newt = (std::fabs(mc - best.st) < std::fabs(mq - best.st)) ? mc : mc + 0.5 * (mq - mc);
    } else if (opposite) {
        brackt = true; bound = false;
        const double mc = cubic_min(best.st, best.f, best.d, t, ft, dt);
        const double mq = quad_min_d(best.st, best.d, t, dt);
        newt = (std::fabs(mc - t) > std::fabs(mq - t)) ? mc : mq;
    } else if (std::fabs(dt) < std::fabs(best.d)) {
        bound = true;
        const double mc = cubic_min_clamped(best.st, best.f, best.d, t, ft, dt, tmin, tmax);
        const double mq = quad_min_d(best.st, best.d, t, dt);
        if (brackt) newt = (std::fabs(t - mc) < std::fabs(t - mq)) ? mc : mq;
        else newt = (std::fabs(t - mc) > std::fabs(t - mq)) ? mc : mq;
    } else {
        bound = false;
        if (brackt) newt = cubic_min(t, ft, dt, other.st, other.f, other.d);
        else newt = (best.st < t) ? tmax : tmin;
    }
    if (best.f < ft) {
        other = LsPoint{t, ft, dt};
    } else {
        if (opposite) other = best;
        best = LsPoint{t, ft, dt};
    }
    newt = std::min(newt, tmax);
    newt = std::max(newt, tmin);
    if (brackt && bound) {
        const double mq = best.st + 0.66 * (other.st - best.st);
        if (best.st < other.st) newt = std::min(newt, mq);
        else newt = std::max(newt, mq);
    }
    t = newt;
    return 0;
}

template <typename T>
struct PlmEngine : PlmEngineBase {
    dca_ctx* ctx;
    int N, L, q, Ls;
    size_t P = 0;
    int Cs = 0;                  // row stride (elements) of W, SR, G
    int Wrows = 0, Grows = 0;
    int Npad = 0;
    double lambda_h = 0, lambda_J = 0;
    int carry_mode = DCA_CARRY_CHUNKED, chunk = 128, warm = 40, halo = 0, add_reg = 1;
    bool configured = false;
    int numScanChunks = 0, numScatChunks = 0;
    static constexpr int kScatWaves = 16;
    int scatSplit = 1, scatChunksPerSplit = 0, scatJW = 2, scatWaves = kScatWavesC;
    int scatBlockChunks = 0;           // float64 mode: tiles per canonical block of sequences (0: plain chains)
    bool scatPerBlock = false;         // ... with one workgroup and one slab of G per block
    // q = 5 in float32: both gather kernels walk site PAIRS on the 25-state combined alphabet (kPairQ; DCA_PLM_PAIRS=0: the
    // per-site blocks, for comparisons).  gUnits = what the kernels' "L" counts: pairs then, sites otherwise.
    bool pairs = false;
    int gUnits = 0;
    int pairJT = 12;                   // site pairs per LDS tile of the logits kernel: of 12 / 11 / 10 the count that pads gUnits least
    int logits_q() const { return pairs ? kPairQ : q; }
    int scatRemCT = 0, scatRemSplit = 0, scatRemChunksPerSplit = 0;     // left-over strips (numCT % 8) in their own, finer split launch

    T *dx = nullptr, *dg = nullptr, *dxp = nullptr, *dgp = nullptr, *dd = nullptr;
    T* dS[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};
    T* dY[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};
    T *dWt = nullptr, *dSR = nullptr, *dR = nullptr, *dG = nullptr, *dw = nullptr;
    LbfgsDev* dLb = nullptr;       // optimiser scalars on the device (two-loop recursion)
    uint16_t* dXL = nullptr;
    uint16_t* dXT2 = nullptr;
    int NT = 0;
    PairIJ* dPairs = nullptr;
    double *dFxPart = nullptr, *dRegPart = nullptr, *dVecPart = nullptr;
    double *dColPart = nullptr, *dColSum = nullptr;      // float64 mode: column sums of R in double-double
    double* dColChunk = nullptr;                         // ... q = 5: per scan chunk, from the softmax kernel
    // Column-strip decomposition (native_mode 4, configure_strips): this rank holds the COLUMNS of sites [cS0, cS1) of W, S, R
    // and G (re-based to column 0; Cs is the window's stride), walks all sequences, and owns the packed parameters
    // [oLo, oHi): the pairs (i, j) whose first site it holds (rank 0 the fields too).  Without it the window is everything.
    bool stripRequested = false, strips = false, stripEmulate = false;
    int sWorld = 1, sRank = 0, cS0 = 0, cS1 = 0, Lloc = 0;
    std::vector<int> siteB;                                 // site boundaries of the ranks (world + 1)
    size_t oLo = 0, oHi = 0;
    int pairBegin = 0, pairEnd = 0;
    T *dGrecv = nullptr, *dXsend = nullptr, *dXrecv = nullptr;
    std::vector<size_t> grecvOff, xsendOff, xrecvOff;      // per peer, elements
    int strip_cs(int r) const { return (int)round_up((size_t)(siteB[r + 1] - siteB[r]) * q, 128); }
    size_t pair_start(int s) const { return (size_t)L * (L - 1) / 2 - (size_t)(L - s) * (L - s - 1) / 2; }     // pairs whose first site is < s
    size_t owned_lo(int r) const { return r == 0 ? 0 : (size_t)L * q + pair_start(siteB[r]) * q * q; }
    size_t owned_hi(int r) const { return (size_t)L * q + pair_start(siteB[r + 1]) * q * q; }
    int nFxPart = 0, nRegPart = 0;
    bool lbfgs_alloc = false;
    bool deferFx = false, fxPending = false;      // fx of the last evaluation still lies in its partial sums (eval_scalars finishes it)
    // vector sharding (dca_plm_set_vector_sharding): this rank's slice [vlo, vlo + vn) of every P-vector;
    // collectives run over Ppad = world * slice elements.  Unsharded: vlo = 0, vn = Ppad = P.
    static constexpr size_t kVecPad = 256;
    size_t vlo = 0, vn = 0, Ppad = 0;
    dca_comm_hook comm = nullptr;
    void* comm_user = nullptr;
    int comm_rank = 0, comm_world = 0;      // of the hook-driven sharding, so that configure() can cut the slices again

    // optimiser state (resumable)
    struct {
        bool begun = false, finished = false;
        int status = 0, k = 1, end = 0, iters = 0, evals = 0, max_iterations = 0, verbose = 0;
        double fx = 0, step = 0, xnorm = 0, gnorm = 0, seconds = 0;
        double last_step = 0;                 // step length the last completed iteration accepted (lbfgs_progress_t's `step`)
        double dginit = 0;                    // g.d of the current search direction
        bool dginit_on_device = false;        // ... still in dScal[kSlotDginit]: read with the next evaluation's scalars
    } o;

    explicit PlmEngine(dca_ctx* c) : ctx(c), N(c->N), L(c->L), q(c->q), Ls(c->Ls) {}

    template <typename U> int dalloc(U** p, size_t n)
    {
        HIP_TRY(dca_dev_malloc(reinterpret_cast<void**>(p), std::max<size_t>(n, 1) * sizeof(U)));
        return DCA_OK;
    }
    void freeall()
    {
        dca_dev_free(dx); dca_dev_free(dg); dca_dev_free(dxp); dca_dev_free(dgp); dca_dev_free(dd);
        for (int i = 0; i < 5; ++i) { dca_dev_free(dS[i]); dca_dev_free(dY[i]); }
        dca_dev_free(dLb); dLb = nullptr; dca_dev_free(dWt); dca_dev_free(dSR); dca_dev_free(dR); dca_dev_free(dG); dca_dev_free(dw); dca_dev_free(dXL); dca_dev_free(dXT2);
        dca_dev_free(dPairs); dca_dev_free(dFxPart); dca_dev_free(dRegPart); dca_dev_free(dVecPart);
        dca_dev_free(dColPart); dca_dev_free(dColSum); dca_dev_free(dColChunk);
        dca_dev_free(dGrecv); dca_dev_free(dXsend); dca_dev_free(dXrecv);
    }
    ~PlmEngine() override { freeall(); }

    int jt() const { return logits_jt(q); }

    // the engine's copy of the weights (dw, made by configure) is stale: everything answers DCA_ERR_STATE until the caller
    // configures again; hooks, native mode and vector sharding are kept (configure re-applies them)
    void weights_changed() override { configured = false; o = decltype(o)(); }

    int configure(double lh, double lJ, int cmode, int chunk_, int warm_, int halo_, int add_reg_) override
    {
        if (q != 21 && q != 5) { dca_set_error("plmDCA kernels are built for q = 21 (protein) and q = 5 (RNA); got %d", q); return DCA_ERR_ARG; }
        if (!ctx->have_weights) { dca_set_error("weights must be computed or set before dca_plm_configure"); return DCA_ERR_STATE; }
        if (halo_ < 0 || halo_ >= N) { dca_set_error("halo out of range"); return DCA_ERR_ARG; }
        if (L > 65535) { dca_set_error("L too large"); return DCA_ERR_ARG; }
        // from here on members are overwritten: an engine that fails below must not keep running with the half-updated window
        // (arrays, siteB and the receive offsets would still be sized for the old one)
        configured = false;
        lambda_h = lh; lambda_J = lJ; carry_mode = cmode; halo = halo_; add_reg = add_reg_;
        // the column window first: the scan's chunking below depends on how many sites this rank walks
        strips = stripRequested && ctx->comm && ctx->comm_world > 1;
        stripRequested = false;
        sWorld = strips ? ctx->comm_world : 1;
        sRank = strips ? ctx->comm_rank : 0;
#ifdef DCA_ROUND_ABLATE
        // ANALYSIS BUILD ONLY: DCA_STRIP_EMULATE=rank,world cuts the column window of that rank WITHOUT a communicator and
        // skips the two exchanges -- the evaluation's results are then wrong, its kernel times are those of one rank of the
        // column-strip decomposition (tools/time_eval.py under the analysis library)
        stripEmulate = false;
        if (const char* e = getenv("DCA_STRIP_EMULATE")) {
            int er = 0, ew = 1;
            if (sscanf(e, "%d,%d", &er, &ew) == 2 && ew > 1 && er >= 0 && er < ew) { strips = true; stripEmulate = true; sRank = er; sWorld = ew; }
        }
#endif
        if (sWorld > kMaxStripRanks || (strips && sWorld > L)) { dca_set_error("column strips: too many ranks for %d sites", L); return DCA_ERR_ARG; }
        if (strips && (halo || hook || comm)) { dca_set_error("column strips take the whole alignment and no hooks"); return DCA_ERR_ARG; }
        siteB.assign(sWorld + 1, 0);
        for (int r = 0; r <= sWorld; ++r) siteB[r] = (int)((long long)L * r / sWorld);
        cS0 = siteB[sRank]; cS1 = siteB[sRank + 1]; Lloc = cS1 - cS0;
        oLo = owned_lo(sRank); oHi = owned_hi(sRank);
        pairBegin = (int)pair_start(cS0); pairEnd = (int)pair_start(cS1);

        // scan chunk: 256 sequences (15 % warm-up rows instead of 31 %) when that still leaves at least one
        // chunk-wave per SIMD and the rows are long (q = 21; config D: 1.20 -> 0.99 ms; with q = 5 the chain
        // latency dominates and 128 stays faster), else 128
        chunk = chunk_ > 0 ? chunk_ : ((q >= 16 && (long long)ceil_div(N - halo_, 256) * ceil_div(Lloc, 64) >= 1024) ? 256 : 128);
        // small alignments: the scan is a chain of one step per sequence and wave, so shorter chunks (more waves, more
        // warm-up rows of a small array) until there is about one chunk-wave per SIMD: config C 0.30 -> 0.16 ms with 32
        if (chunk_ <= 0)
            while (chunk > 32 && (long long)ceil_div(N - halo_, chunk) * ceil_div(Lloc, 64) < 1024) chunk /= 2;
        // warm-up steps of the chunk-parallel scan: 2^-40 of start-up error is far below float rounding; the float64 mode is
        // the parity mode and takes 80, with which the chunked scan is BIT-identical to the serial chain (the start-up
        // error has dropped below the last place of every carried probability; 100 iterations at configs D and E end in the
        // same bits, profiles/r04_sensitivity_*.json)
        warm = warm_ > 0 ? warm_ : (sizeof(T) == 8 ? 80 : 40);
        if (carry_mode == DCA_CARRY_SERIAL) { chunk = N - halo; warm = halo; }
        if (carry_mode == DCA_CARRY_EXACT) warm = 0;
        numScanChunks = ceil_div(N - halo, chunk);
        numScatChunks = ceil_div(N - halo, kNC);

        freeall();
        dx = dg = dxp = dgp = dd = nullptr;
        for (int i = 0; i < 5; ++i) dS[i] = dY[i] = nullptr;
        dWt = dSR = dR = dG = dw = nullptr; dXL = nullptr; dXT2 = nullptr; dPairs = nullptr;
        dFxPart = dRegPart = dVecPart = nullptr;
        dColPart = dColSum = dColChunk = nullptr;
        dGrecv = dXsend = dXrecv = nullptr;
        lbfgs_alloc = false;
        o = decltype(o)();

        P = dca_plm_num_params(L, q);
        const int Lq = L * q;
        const int LqLoc = Lloc * q;
        Cs = (int)round_up(LqLoc, 128);
        {
            const char* pe = getenv("DCA_PLM_PAIRS");
            pairs = q == 5 && sizeof(T) == 4 && !(pe && atoi(pe) == 0);
        }
        gUnits = pairs ? ceil_div(L, 2) : L;
        pairJT = 12;
        if (pairs)
            for (int jt : {11, 10})
                if (ceil_div(gUnits, jt) * jt < ceil_div(gUnits, pairJT) * pairJT) pairJT = jt;      // L = 150: 75 pairs = 7 x 11 (77) rather than 7 x 12 (84)
        const int JT = pairs ? pairJT : logits_jt(q);       // units per logits tile
        Wrows = ceil_div(gUnits, JT) * (pairs ? JT * 2 * q : JT * q) + 128;     // + over-read margin of the last LDS-DMA tile
        scatJW = 2;     // units per wave of the scatter kernel
        const int JG = kScatWavesC * scatJW;
        Grows = ceil_div(gUnits, JG) * JG * (pairs ? 2 * q : q);
        Npad = (int)round_up(N, logits_seq_per_wg(logits_q()));

        DCA_TRY(dalloc(&dx, P + kVecPad)); DCA_TRY(dalloc(&dg, P + kVecPad));
        DCA_TRY(dalloc(&dWt, (size_t)Wrows * Cs));
        DCA_TRY(dalloc(&dSR, (size_t)N * Cs));                  // S: logit sums
        DCA_TRY(dalloc(&dR, (size_t)(N + kNC) * Cs));           // R = w (p - delta); + kNC zero rows: the scatter kernel's last tile reads past row N-1
        HIP_TRY(hipMemsetAsync(dR, 0, (size_t)(N + kNC) * Cs * sizeof(T), ctx->stream));    // pad columns and halo rows stay zero
        {
            // Split of the tile range (every split writes its own slab of G) and the left-over launch.  Strips are dealt to
            // the XCDs in sets of eight (plm_scatter_kernel), the numJG site groups of a strip and split run side by side on
            // one XCD's 32 CUs, and a workgroup costs its tiles + about two for prologue and epilogue.  numCT % 8 left-over
            // strips keep that many XCDs busy for whole extra rounds while the others idle (D: 83 strips = 11 rounds on
            // three XCDs, 10 on five), so they may get their own launch with a finer split that spreads them over all XCDs
            // for a fraction of a round.  Every extra slab costs the fold one more pass over G (about `slabUnits` tile
            // times).  Model: cost = rounds x (tiles per workgroup + 2) [+ the same for the left-over launch] + slabs;
            // candidates up to ~2048 workgroups with >= 12 tiles each.  Measured (tools/time_eval.py, DCA_SCATTER_SPLIT /
            // DCA_SCATTER_REM; scatter + fold, ms): D 1 + left-over 7.28 (split 2 without: 7.74), D/8 1.19 (1.34),
            // C 0.376 (0.458 for the best split without a left-over launch), E split 19-32: 0.90 (51: 0.93).
            const int cw = kRowBytes / (int)sizeof(T);
            const int numCT = ceil_div(Cs, cw), numJGs = ceil_div(gUnits, JG);
            const int fullCT = numCT / kNumXcd * kNumXcd, rem = numCT - fullCT;
            const int s0 = std::max(1, std::min({numScatChunks, ceil_div(2048, numCT * numJGs), std::max(1, numScatChunks / 12)}));
            const int cuPerXcd = 256 / kNumXcd;
            auto rounds = [&](long long wgsPerXcd) { return (double)((wgsPerXcd + cuPerXcd - 1) / cuPerXcd); };
            const double slabUnits = (double)Grows * Cs * sizeof(T) / 4e12 / 4e-6;      // one pass over a slab at ~4 TB/s, in 4 us tile times
            const char* remEnv = getenv("DCA_SCATTER_REM");        // tuning / test knob: 0 never, 1 whenever there are left-over strips
            const char* splitEnv = getenv("DCA_SCATTER_SPLIT");    // tuning knob: the split of the main launch
            double bestCost = 1e300;
            scatSplit = 1; scatChunksPerSplit = numScatChunks; scatRemCT = scatRemSplit = scatRemChunksPerSplit = 0;
            // float64 = parity mode: the oracle's order of summation -- per (site, state, column) the sequences in ascending
            // order inside blocks of kCanonBlock, the block sums added in ascending block order (the test oracle's
            // ORACLE_CANONICAL_BLOCK; round 4: one chain over all N) -- so that the gradient does not depend on the launch
            // geometry.  Two geometries give exactly that order: ONE workgroup per (strip, site group) that adds its
            // finished block to the running sum in G and restarts its chains (plm_scatter_kernel, blockChunks), or one
            // workgroup and one slab PER BLOCK, the slabs summed in ascending order by plm_sum_slabs_kernel.  The second
            // fills the chip where strips x site groups do not (config E: 12 x 5 = 60 workgroups, 13 blocks: scatter 4.03 ->
            // 1.39 ms), the first saves the slab traffic where they do (config D: 2656 workgroups, 4 blocks: 14.4 ms against
            // 16.6 + 0.9 in the fold; round 4's single chain 13.3 -- each of the three read-modify-write passes over G stalls
            // the lock-stepped workgroups for 0.35 ms, which is why the blocks are 16384 and not 4096 sequences).  No separate
            // launch for the left-over strips; a test that forces a split or that launch leaves the canonical order.
            const bool canonical = sizeof(T) == 8 && !splitEnv && !remEnv;
            scatWaves = kScatWavesC;
            scatBlockChunks = 0;
            scatPerBlock = false;
            if (canonical) {
                scatBlockChunks = kCanonBlock / kNC;
                const int nblocks = ceil_div(numScatChunks, scatBlockChunks);
                const char* we = getenv("DCA_SCATTER_WAVES");          // tuning knob (one-workgroup geometry)
                const char* ge = getenv("DCA_SCATTER_CANON");          // tuning / test knob: 1 one workgroup, 2 slab per block
                auto perXcdOf = [&](int waves, int sp) {
                    const int njg = ceil_div(gUnits, waves * scatJW);
                    return fullCT > 0 ? (long long)ceil_div(numCT, kNumXcd) * njg * sp : (long long)ceil_div(numCT * sp, kNumXcd) * njg;
                };
                // one workgroup: 16 waves, or 8 where that does not fill the chip (twice the workgroups; 4 waves measured slower)
                int wavesA = kScatWavesC;
                if (we && (atoi(we) == 16 || atoi(we) == 8 || atoi(we) == 4)) wavesA = atoi(we);
                else if ((long long)numCT * ceil_div(gUnits, kScatWavesC * scatJW) < 192) wavesA = 8;
                // (a tile of an 8-wave workgroup takes 0.85 of a 16-wave one's time: E 3.87 against 4.53 ms on one round each;
                // the (strip, block) pairs of the second geometry are dealt to the XCDs one by one, see launch_eval)
                const double costA = rounds(perXcdOf(wavesA, 1)) * (numScatChunks * (wavesA == 8 ? 0.85 : 1.0) + 2.0 + 0.5 * (nblocks - 1));
                const double costB = rounds((long long)ceil_div(numCT * nblocks, kNumXcd) * numJGs) * (scatBlockChunks + 2.0) + (nblocks - 1) * slabUnits;
                scatPerBlock = nblocks > 1 && (ge ? atoi(ge) == 2 : costB < costA);
                if (scatPerBlock) { scatSplit = nblocks; scatChunksPerSplit = scatBlockChunks; }
                else scatWaves = wavesA;
            }
            for (int sp = 1; sp <= (canonical ? 0 : (splitEnv ? numScatChunks : s0)); ++sp) {
                if (splitEnv && sp != std::max(1, std::min(numScatChunks, atoi(splitEnv)))) continue;
                const int cps = ceil_div(numScatChunks, sp);
                if (ceil_div(numScatChunks, cps) != sp && !splitEnv) continue;               // same as a smaller split
                const int spEff = ceil_div(numScatChunks, cps);
                const double slabs = (spEff - 1) * slabUnits;
                if (!(remEnv && atoi(remEnv) == 1 && fullCT > 0 && rem > 0)) {
                    // fewer than eight strips: the (strip, split) pairs, not the strips, are dealt to the XCDs (launch_eval)
                    const long long perXcd = fullCT > 0 ? (long long)ceil_div(numCT, kNumXcd) * numJGs * spEff
                                                        : (long long)ceil_div(numCT * spEff, kNumXcd) * numJGs;
                    const double cost = rounds(perXcd) * (cps + 2.0) + slabs;
                    if (cost < bestCost) { bestCost = cost; scatSplit = spEff; scatChunksPerSplit = cps; scatRemCT = scatRemSplit = scatRemChunksPerSplit = 0; }
                }
                if (fullCT > 0 && rem > 0 && !(remEnv && atoi(remEnv) == 0)) {
                    int sB = std::max(spEff, 256 / (rem * numJGs));
                    sB = std::max(1, std::min(sB, std::max(1, numScatChunks / 12)));
                    const int cpsB = ceil_div(numScatChunks, sB);
                    sB = ceil_div(numScatChunks, cpsB);
                    const double cost = rounds((long long)(fullCT / kNumXcd) * numJGs * spEff) * (cps + 2.0) +
                                        rounds((long long)ceil_div(rem * sB, kNumXcd) * numJGs) * (cpsB + 2.0) + slabs + 3.0;   // + two more launches
                    if (cost < bestCost) { bestCost = cost; scatSplit = spEff; scatChunksPerSplit = cps; scatRemCT = rem; scatRemSplit = sB; scatRemChunksPerSplit = cpsB; }
                }
            }
        }
        DCA_TRY(dalloc(&dG, (size_t)std::max(scatSplit, scatRemSplit) * Grows * Cs));
        DCA_TRY(dalloc(&dw, N));
        DCA_TRY(dalloc(&dXL, (size_t)ceil_div(gUnits, JT) * JT * Npad));
        NT = numScatChunks * kNC;
        DCA_TRY(dalloc(&dXT2, (size_t)gUnits * NT));
        const size_t npairs = (size_t)L * (L - 1) / 2;
        DCA_TRY(dalloc(&dPairs, npairs));
        nFxPart = ceil_div(Lloc, 64) * ceil_div(numScanChunks, 4) * 4;
        nRegPart = (int)npairs + ceil_div(Lq, 256);
        DCA_TRY(dalloc(&dFxPart, 2 * (size_t)nFxPart));                       // (hi, lo) pairs
        DCA_TRY(dalloc(&dRegPart, 2 * (size_t)(nRegPart + kSumStageBlocks)));      // pairs; + the first-stage sums of the regulariser partials
        DCA_TRY(dalloc(&dVecPart, 2 * 27 * kVecBlocks));      // (hi, lo) pairs
        // with column strips only the owned pairs' (and the window's field blocks') partials are written: the others must read as zero
        HIP_TRY(hipMemsetAsync(dRegPart, 0, 2 * (size_t)(nRegPart + kSumStageBlocks) * sizeof(double), ctx->stream));
        HIP_TRY(hipMemsetAsync(dFxPart, 0, 2 * (size_t)nFxPart * sizeof(double), ctx->stream));
        if (sizeof(T) == 8) {
            DCA_TRY(dalloc(&dColPart, 2 * (size_t)kColSumRowBlocks * Lq));
            if (q == 5) DCA_TRY(dalloc(&dColChunk, 2 * (size_t)numScanChunks * Lq));      // the softmax kernel's per-chunk column sums
            DCA_TRY(dalloc(&dColSum, (size_t)Lq));
        }
        grecvOff.assign(sWorld + 1, 0); xsendOff.assign(sWorld + 1, 0); xrecvOff.assign(sWorld + 1, 0);
        if (strips) {
            const size_t q2 = (size_t)q * q;
            size_t gtot = 0, stot = 0, rtot = 0;
            for (int r = 0; r < sWorld; ++r) {
                grecvOff[r] = gtot; xsendOff[r] = stot; xrecvOff[r] = rtot;
                if (r > sRank) { gtot += (size_t)LqLoc * strip_cs(r); stot += (size_t)Lloc * (siteB[r + 1] - siteB[r]) * q2; }
                if (r < sRank) rtot += (size_t)(siteB[r + 1] - siteB[r]) * Lloc * q2;
            }
            DCA_TRY(dalloc(&dGrecv, gtot)); DCA_TRY(dalloc(&dXsend, stot)); DCA_TRY(dalloc(&dXrecv, rtot));
        }

        HIP_TRY(hipMemsetAsync(dx, 0, (P + kVecPad) * sizeof(T), ctx->stream));
        HIP_TRY(hipMemsetAsync(dg, 0, (P + kVecPad) * sizeof(T), ctx->stream));
        // the exchange scheme (reduce hook, vector-sharding hook, native mode) survives a re-configuration -- a context whose
        // weights changed must be configured again and would otherwise silently fall back to unreduced local sums
        vlo = 0; vn = P; Ppad = P;
        if (strips) {
            native_mode = 4;
            vlo = oLo; vn = oHi - oLo;
        } else if (native_mode == 4) {
            native_mode = 0;
        }
        if (native_mode == 2 || native_mode == 3) {
            if (!ctx->comm) native_mode = 0;
            else DCA_TRY(set_slices(ctx->comm_rank, ctx->comm_world));
        } else if (comm) {
            DCA_TRY(set_slices(comm_rank, comm_world));
        }
        if (native_mode == 1 && !ctx->comm) native_mode = 0;
        HIP_TRY(hipMemsetAsync(dWt, 0, (size_t)Wrows * Cs * sizeof(T), ctx->stream));
        HIP_TRY(hipMemsetAsync(dG, 0, (size_t)std::max(scatSplit, scatRemSplit) * Grows * Cs * sizeof(T), ctx->stream));

        std::vector<PairIJ> hp(npairs);
        {
            size_t k = 0;
            for (int i = 0; i < L - 1; ++i) for (int j = i + 1; j < L; ++j) hp[k++] = PairIJ{(uint16_t)i, (uint16_t)j};
        }
        HIP_TRY(hipMemcpyAsync(dPairs, hp.data(), npairs * sizeof(PairIJ), hipMemcpyHostToDevice, ctx->stream));
        HIP_TRY(hipStreamSynchronize(ctx->stream));

        // weights in T.  1/count is formed in T exactly as the reference does
        // (1.f/count, plmdca_numerics.cpp:669) when the counts are known.
        std::vector<T> hw(N);
        {
            std::vector<double> wd(N);
            HIP_TRY(hipMemcpy(wd.data(), ctx->dWd, (size_t)N * sizeof(double), hipMemcpyDeviceToHost));
            if (ctx->have_counts) {
                std::vector<uint32_t> cnt(N);
                HIP_TRY(hipMemcpy(cnt.data(), ctx->dCounts, (size_t)N * sizeof(uint32_t), hipMemcpyDeviceToHost));
                for (int n = 0; n < N; ++n) hw[n] = (T)1 / (T)cnt[n];
            } else {
                for (int n = 0; n < N; ++n) hw[n] = (T)wd[n];
            }
        }
        HIP_TRY(hipMemcpy(dw, hw.data(), (size_t)N * sizeof(T), hipMemcpyHostToDevice));

        if (pairs) {
            hipLaunchKernelGGL(plm_build_pair_states_kernel, dim3(ceil_div(Npad, 256), ceil_div(gUnits, JT) * JT), dim3(256), 0,
                               ctx->stream, ctx->dX, dXL, N, Npad, L, Ls, 0, 0x2000u);
            hipLaunchKernelGGL(plm_build_pair_states_kernel, dim3(ceil_div(NT, 256), gUnits), dim3(256), 0, ctx->stream,
                               ctx->dX, dXT2, N, NT, L, Ls, halo, 0x9000u);
            HIP_TRY(hipGetLastError());
            HIP_TRY(hipStreamSynchronize(ctx->stream));
        } else {
            hipLaunchKernelGGL(plm_build_logit_states_kernel, dim3(ceil_div(Npad, 256), ceil_div(L, JT) * JT), dim3(256), 0,
                               ctx->stream, ctx->dX, dXL, N, Npad, L, Ls);
            hipLaunchKernelGGL(plm_build_states_kernel, dim3(ceil_div(NT, 256), L), dim3(256), 0, ctx->stream,
                               ctx->dX, dXT2, N, L, Ls, halo, NT);
            HIP_TRY(hipGetLastError());
            HIP_TRY(hipStreamSynchronize(ctx->stream));
        }
        configured = true;
        return DCA_OK;
    }

    int configure_strips(double lh, double lJ, int cmode, int chunk_, int warm_) override
    {
        if (!ctx->comm) { dca_set_error("column strips need a communicator: dca_comm_init first"); return DCA_ERR_STATE; }
        if (o.begun && !o.finished) { dca_set_error("the decomposition cannot change during an optimisation"); return DCA_ERR_STATE; }
        // the caller's hooks go only if the new decomposition stands: a failed call leaves them as they were (the engine
        // itself is then unconfigured -- configure() says so -- and must be configured again either way)
        const auto hook0 = hook; const auto hookUser0 = hook_user; const auto comm0 = comm; const auto commUser0 = comm_user;
        const int commRank0 = comm_rank, commWorld0 = comm_world;
        hook = nullptr; hook_user = nullptr; comm = nullptr; comm_user = nullptr; comm_rank = comm_world = 0;
        stripRequested = true;
        const int rc = configure(lh, lJ, cmode, chunk_, warm_, 0, 1);
        stripRequested = false;
        if (rc != DCA_OK) { hook = hook0; hook_user = hookUser0; comm = comm0; comm_user = commUser0; comm_rank = commRank0; comm_world = commWorld0; strips = false; }
        return rc;
    }

    // PlmDCA::initFieldsAndCouplings (plmdca_numerics.cpp:207-249) in T, host side
    // (L*q values from an N*L pass; not worth a kernel).
    int init_x() override
    {
        if (!configured) { dca_set_error("dca_plm_configure first"); return DCA_ERR_STATE; }
        std::vector<T> hw(N);
        HIP_TRY(hipMemcpy(hw.data(), dw, (size_t)N * sizeof(T), hipMemcpyDeviceToHost));
        T meff = 0;
        for (int n = 0; n < N; ++n) meff += hw[n];
        std::vector<T> h((size_t)L * q, (T)0);
        const uint8_t* X = dca_host_msa(ctx);
        if (!X) return DCA_ERR_HIP;
        for (int n = 0; n < N; ++n)
            for (int i = 0; i < L; ++i) h[(size_t)i * q + X[(size_t)n * L + i]] += hw[n];
        for (int i = 0; i < L; ++i) {
            T* hi = h.data() + (size_t)i * q;
            for (int a = 0; a < q; ++a) hi[a] /= meff;
            for (int a = 0; a < q; ++a) hi[a] = std::log(hi[a] * meff + (T)1);
            T s = 0;
            for (int a = 0; a < q; ++a) s += hi[a];
            const T av = s / (T)q;
            for (int a = 0; a < q; ++a) hi[a] -= av;
        }
        HIP_TRY(hipMemsetAsync(dx, 0, P * sizeof(T), ctx->stream));
        HIP_TRY(hipMemcpyAsync(dx, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice, ctx->stream));
        HIP_TRY(hipStreamSynchronize(ctx->stream));
        return DCA_OK;
    }

    template <typename U> int upload(const U* src, T* dst)
    {
        std::vector<T> tmp(P);
        for (size_t i = 0; i < P; ++i) tmp[i] = (T)src[i];
        HIP_TRY(hipMemcpy(dst, tmp.data(), P * sizeof(T), hipMemcpyHostToDevice));
        return DCA_OK;
    }
    template <typename U> int download(const T* src, U* dst)
    {
        std::vector<T> tmp(P);
        HIP_TRY(hipStreamSynchronize(ctx->stream));
        HIP_TRY(hipMemcpy(tmp.data(), src, P * sizeof(T), hipMemcpyDeviceToHost));
        for (size_t i = 0; i < P; ++i) dst[i] = (U)tmp[i];
        return DCA_OK;
    }
    int set_x(const void* x, int dtype) override
    {
        if (!configured) { dca_set_error("dca_plm_configure first"); return DCA_ERR_STATE; }
        if (dtype == DCA_F32) return upload(static_cast<const float*>(x), dx);
        if (dtype == DCA_F64) return upload(static_cast<const double*>(x), dx);
        return DCA_ERR_ARG;
    }
    int get_x(void* x, int dtype) override
    {
        if (!configured) return DCA_ERR_STATE;
        if (native_mode == 4 && !stripEmulate) DCA_TRY(strip_allgather(dx));       // collective, like get_g: every rank calls it
        if (dtype == DCA_F32) return download(dx, static_cast<float*>(x));
        if (dtype == DCA_F64) return download(dx, static_cast<double*>(x));
        return DCA_ERR_ARG;
    }
    int get_g(void* g, int dtype) override
    {
        if (!configured) return DCA_ERR_STATE;
        DCA_TRY(gather_vector(dg));
        if (dtype == DCA_F32) return download(dg, static_cast<float*>(g));
        if (dtype == DCA_F64) return download(dg, static_cast<double*>(g));
        return DCA_ERR_ARG;
    }

    template <int Q> int launch_eval()
    {
        hipStream_t st = ctx->stream;
        const size_t npairs = (size_t)L * (L - 1) / 2;
        const int Lq = L * q;
        {
            ScopedKernelClock kc(ctx, "plm_expand");
            hipLaunchKernelGGL(plm_expand_kernel<T>, dim3((unsigned)npairs), dim3(256), (size_t)q * q * sizeof(T), st,
                               dx, dWt, dPairs, L, q, Cs, cS0, cS1);
        }
        DCA_ROUND_STAGE(1, dWt, (size_t)Wrows * Cs);
        {
            constexpr int CW = 512 / (int)sizeof(T);
            const int numCT = ceil_div(Cs, CW);
            auto launch = [&](auto kern, int QL) -> int {          // QL: the kernel's alphabet (kPairQ: gUnits site pairs)
                const int numNB = Npad / logits_seq_per_wg(QL);
                const int blocks = numCT * numNB;
                const size_t lds = (size_t)2 * 128 * 512 + (size_t)logits_waves(QL) * 256;   // two tiles + prefetch scratch
                HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
                ScopedKernelClock kc(ctx, "plm_logits");
                hipLaunchKernelGGL(kern, dim3(blocks), dim3(logits_waves(QL) * 64), lds, st, dWt, dXL, dSR, N, Npad, gUnits, Cs, numCT, numNB);
                return DCA_OK;
            };
            bool done = false;
            if constexpr (Q == 5 && sizeof(T) == 4) {
                if (pairs) {
                    if (pairJT == 11) DCA_TRY(launch(plm_logits_kernel<T, kPairQ, 11>, kPairQ));
                    else if (pairJT == 10) DCA_TRY(launch(plm_logits_kernel<T, kPairQ, 10>, kPairQ));
                    else DCA_TRY(launch(plm_logits_kernel<T, kPairQ>, kPairQ));
                    done = true;
                }
            }
            if (!done) DCA_TRY(launch(plm_logits_kernel<T, Q>, Q));
        }
        DCA_ROUND_STAGE(2, dSR, (size_t)N * Cs);
        {
            dim3 grid(ceil_div(Lloc, 64), ceil_div(numScanChunks, 4));
            ScopedKernelClock kc(ctx, "plm_softmax");
            constexpr int softNP = (64 * Q * (int)sizeof(T) + 1023) / 1024;
            const size_t softLds = (size_t)4 * 2 * softNP * 1024;
            HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(plm_softmax_kernel<T, Q>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)softLds));
            // the window's sites: their fields, their alignment column, their columns of S / R
            hipLaunchKernelGGL((plm_softmax_kernel<T, Q>), grid, dim3(256), softLds, st, dSR, dR, dx + (size_t)cS0 * q, ctx->dX + cS0, dw, dFxPart,
                               N, Lloc, Ls, Cs, halo, chunk, warm, carry_mode != DCA_CARRY_EXACT ? 1 : 0, numScanChunks, dColChunk);
        }
        DCA_ROUND_STAGE(4, dR, (size_t)N * Cs);
        {
            constexpr int CW = kRowBytes / (int)sizeof(T);
            const int numCT = ceil_div(Cs, CW);
            const int numJG = ceil_div(gUnits, scatWaves * scatJW);
            const int scatThreads = scatWaves * 64;
            const int mainCT = numCT - scatRemCT;            // strips of the main launch (all of them without a left-over launch)
            const size_t lds = (size_t)2 * kNC * kRowBytes;
            static const bool mergeRemEnv = !(getenv("DCA_SCATTER_MERGE") && atoi(getenv("DCA_SCATTER_MERGE")) == 0);
            const bool mergeRem = mergeRemEnv && !(numCT < kNumXcd || scatPerBlock);
            auto launch = [&](auto kern) -> int {
                HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
                ScopedKernelClock kc(ctx, "plm_scatter");
                if (numCT < kNumXcd || scatPerBlock)        // E: 6 strips would leave two XCDs idle: deal the (strip, split) pairs to the XCDs instead
                    hipLaunchKernelGGL(kern, dim3(kNumXcd * ceil_div(numCT * scatSplit, kNumXcd) * numJG, 1), dim3(scatThreads), lds, st, dR, dXT2, dG,
                                       N, gUnits, Cs, halo, numScatChunks, NT, 0, numCT * scatSplit, scatSplit, numJG, scatChunksPerSplit, (size_t)Grows * Cs, scatBlockChunks, 0x7fffffff, 0, 0, 1, 0);
                else if (scatRemCT && mergeRem) {
                    // main strips and left-over strips in ONE launch (the left-over workgroups behind the main ones of every row)
                    const int remPairs = scatRemCT * scatRemSplit;
                    const int mainX = kNumXcd * ceil_div(mainCT, kNumXcd) * numJG, remX = kNumXcd * ceil_div(remPairs, kNumXcd) * numJG;
                    hipLaunchKernelGGL(kern, dim3(mainX + remX, scatSplit), dim3(scatThreads), lds, st, dR, dXT2, dG,
                                       N, gUnits, Cs, halo, numScatChunks, NT, 0, mainCT, 1, numJG, scatChunksPerSplit, (size_t)Grows * Cs, scatBlockChunks,
                                       mainX, mainCT, remPairs, scatRemSplit, scatRemChunksPerSplit);
                } else
                    hipLaunchKernelGGL(kern, dim3(kNumXcd * ceil_div(mainCT, kNumXcd) * numJG, scatSplit), dim3(scatThreads), lds, st, dR, dXT2, dG,
                                       N, gUnits, Cs, halo, numScatChunks, NT, 0, mainCT, 1, numJG, scatChunksPerSplit, (size_t)Grows * Cs, scatBlockChunks, 0x7fffffff, 0, 0, 1, 0);
                if (scatRemCT) {
                    const int remPairs = scatRemCT * scatRemSplit;
                    if (!mergeRem)
                        hipLaunchKernelGGL(kern, dim3(kNumXcd * ceil_div(remPairs, kNumXcd) * numJG, 1), dim3(scatThreads), lds, st, dR, dXT2, dG,
                                           N, gUnits, Cs, halo, numScatChunks, NT, mainCT, remPairs, scatRemSplit, numJG, scatRemChunksPerSplit, (size_t)Grows * Cs, 0, 0x7fffffff, 0, 0, 1, 0);
                    const int col0 = mainCT * CW, ncols = Cs - col0;
                    hipLaunchKernelGGL(plm_sum_slabs_cols_kernel<T>, dim3((unsigned)(((size_t)Lq * ncols + 255) / 256)), dim3(256), 0, st, dG,
                                       (size_t)Grows * Cs, Cs, col0, ncols, Lq, scatRemSplit, scatSplit);
                }
                return DCA_OK;
            };
            if constexpr (sizeof(T) == 8) {
                if (scatWaves == 8) DCA_TRY(launch(plm_scatter_kernel<T, Q, 2, 8>));
                else if (scatWaves == 4) DCA_TRY(launch(plm_scatter_kernel<T, Q, 2, 4>));
                else DCA_TRY(launch(plm_scatter_kernel<T, Q, 2, 16>));
            } else if constexpr (Q == 5) {
                if (pairs) DCA_TRY(launch(plm_scatter_kernel<T, kPairQ, 2, 16>));
                else DCA_TRY(launch(plm_scatter_kernel<T, Q, 2, 16>));
            } else {
                DCA_TRY(launch(plm_scatter_kernel<T, Q, 2, 16>));
            }
        }
        DCA_ROUND_STAGE(8, dG, (size_t)std::max(scatSplit, scatRemSplit) * Grows * Cs);
        {
            ScopedKernelClock kc(ctx, "plm_fold");
            const int LqLoc = Lloc * q;
            int foldSlabs = scatSplit;
            if (scatSplit > 2 || (strips && scatSplit > 1)) {     // more than two slabs: one streaming pass is cheaper than strided reads in the fold (and rows that travel are sent summed)
                hipLaunchKernelGGL(plm_sum_slabs_kernel<T>, dim3(2048), dim3(256), 0, st, dG, (size_t)Grows * Cs, scatSplit);
                foldSlabs = 1;
            }
            if (dColSum) {
                if (dColChunk)
                    hipLaunchKernelGGL(plm_colsum_chunks_kernel, dim3(ceil_div(LqLoc, 256), kColSumRowBlocks), dim3(256), 0, st, dColChunk, numScanChunks, LqLoc, dColPart);
                else
                    hipLaunchKernelGGL(plm_colsum_parts_kernel<T>, dim3(ceil_div(LqLoc, 64), kColSumRowBlocks), dim3(256), 0, st, dR, N, Cs, LqLoc, dColPart);
                hipLaunchKernelGGL(plm_colsum_final_kernel, dim3(ceil_div(LqLoc, 256)), dim3(256), 0, st, dColPart, kColSumRowBlocks, LqLoc, dColSum);
            }
            // one GPU: the field fold's few workgroups ride behind the pair fold's in ONE launch (same threads, same sums; a launch
            // boundary and a 5 - 11 us kernel less per evaluation); with strips the gradient-table rows travel in between
            static const bool mergeFieldsEnv = !(getenv("DCA_FOLD_MERGE") && atoi(getenv("DCA_FOLD_MERGE")) == 0);
            const int nOwnedPairs = pairEnd - pairBegin;
            const bool mergeFields = mergeFieldsEnv && !strips && nOwnedPairs > 0;
            if (!mergeFields)
            hipLaunchKernelGGL(plm_fold_fields_kernel<T>, dim3(ceil_div(LqLoc, 256)), dim3(256), 0, st, dx + (size_t)cS0 * q, dG, dg + (size_t)cS0 * q,
                               dRegPart + 2 * npairs, LqLoc, q, Cs, (T)lambda_h, add_reg, (size_t)Grows * Cs, foldSlabs, dColSum);
            StripMap sm;
            sm.rank = sRank; sm.world = sWorld; sm.s0 = cS0; sm.s1 = cS1;
            for (int r = 0; r <= sWorld; ++r) sm.site0[r] = siteB[r];
            for (int r = 0; r < sWorld; ++r) { sm.recv[r] = strips && r > sRank ? dGrecv + grecvOff[r] : nullptr; sm.recvCs[r] = strips ? strip_cs(r) : 0; }
            if (strips && !stripEmulate) DCA_TRY(exchange_g());
            const size_t lds = (size_t)kFoldWaves * ((q * q + 3) / 4 * 4) * sizeof(T);
            const int nOwned = pairEnd - pairBegin;
            if (nOwned > 0) {
                const int pairBlocks = ceil_div(nOwned, kFoldWaves);
                FoldFieldsArgs<T> ff{dx + (size_t)cS0 * q, dg + (size_t)cS0 * q, dRegPart + 2 * npairs, LqLoc, (T)lambda_h, dColSum, mergeFields ? pairBlocks : -1};
                hipLaunchKernelGGL(plm_fold_pairs_kernel<T>, dim3((unsigned)(pairBlocks + (mergeFields ? ceil_div(LqLoc, 256) : 0))), dim3(64 * kFoldWaves), lds, st, dx, dG, dg, dPairs,
                                   dRegPart, L, q, Cs, (T)lambda_J, add_reg, (size_t)Grows * Cs, foldSlabs, pairBegin, pairEnd, sm, ff);
            }
        }
        DCA_ROUND_STAGE(16, dg, P);
        // fx = regulariser + data term  -> ctx->dScal[0]
        // (one partial per site pair: summed in two stages, a single workgroup needs 28 us for the 125 000 of config D)
        // (the optimiser on one GPU sums fx inside the two launches of its dot products: eval_scalars)
        fxPending = deferFx;
        if (!deferFx) {
            hipLaunchKernelGGL(dd_sum_chunks_kernel, dim3(kSumStageBlocks), dim3(256), 0, st, dRegPart, nRegPart, dRegPart + 2 * (size_t)nRegPart);
            hipLaunchKernelGGL(dd_sum_final_kernel, dim3(1), dim3(1024), 0, st, dRegPart + 2 * (size_t)nRegPart, kSumStageBlocks, dFxPart, nFxPart, ctx->dScal);
        }
        HIP_TRY(hipGetLastError());
        return DCA_OK;
    }

    // leaves fx in ctx->dScal[0] (device); no host sync unless a reduce hook is set
    // defer_fx: the caller is eval_scalars' (one GPU, no hook: nothing reads fx before the dot products are formed)
    int evaluate_async(bool defer_fx = false)
    {
        if (!configured) { dca_set_error("dca_plm_configure first"); return DCA_ERR_STATE; }
        static const bool fuseFx = !(getenv("DCA_PLM_FUSE_FX") && atoi(getenv("DCA_PLM_FUSE_FX")) == 0);
        deferFx = defer_fx && fuseFx && native_mode == 0 && !hook && !comm && !stripEmulate;
        int rc = (q == 21) ? launch_eval<21>() : launch_eval<5>();
        if (rc != DCA_OK) return rc;
        o.evals += 1;
        if (native_mode == 4) {
            // column strips: both exchanges happened inside launch_eval; fx is summed with the caller's scalars
        } else if (comm || native_mode == 2 || native_mode == 3) {
            // sharded vectors: sum the shards' gradients, keep this rank's slice; fx is summed with the
            // scalars of the caller (eval_scalars / gradient)
            DCA_TRY(do_comm(DCA_COMM_REDUCE_SCATTER, dg, Ppad, (int)sizeof(T) * 8, "reduce-scatter"));
        } else if (native_mode == 1) {
            DCA_TRY(dca_comm_native_reduce(ctx, dg, P, (int)sizeof(T) * 8, ctx->dScal));      // on the stream: no host round trip
        } else if (hook) {
            HIP_TRY(hipStreamSynchronize(ctx->stream));
            if (hook(hook_user, dg, P, (int)sizeof(T) * 8, ctx->dScal) != 0) {
                dca_set_error("reduce hook failed");
                return DCA_ERR_ARG;
            }
        }
        return DCA_OK;
    }

    // sum `count` device doubles ctx->dScal[first..] over the ranks (no-op when vectors are not sharded)
    // one collective of the sharded optimiser: through the native communicator on the stream (no host round trip),
    // or through the caller's hook (the stream is drained first: the hook works outside it)
    int do_comm(int op, void* buf, size_t count, int dtype, const char* what)
    {
        if (native_mode >= 2) return dca_comm_native(ctx, op, buf, count, dtype, native_mode == 3);      // mode 4: only the scalar all-reduce comes here
        HIP_TRY(hipStreamSynchronize(ctx->stream));
        if (comm(comm_user, op, buf, count, dtype) != 0) { dca_set_error("comm hook failed (%s)", what); return DCA_ERR_ARG; }
        return DCA_OK;
    }
    int reduce_scalars(int first, int count)
    {
        if ((!comm && native_mode < 2) || stripEmulate) return DCA_OK;
        return do_comm(DCA_COMM_ALL_REDUCE, ctx->dScal + first, (size_t)count, DCA_F64, "all-reduce");
    }
    // make a P-vector whose slices are valid on their owners valid everywhere
    int gather_vector(T* v)
    {
        if (!comm && native_mode < 2) return DCA_OK;
        if (native_mode == 4) return stripEmulate ? DCA_OK : strip_allgather(v);
        return do_comm(DCA_COMM_ALL_GATHER, v, Ppad, (int)sizeof(T) * 8, "all-gather");
    }
    // after a step: every rank needs x where its evaluation reads it
    int publish_x()
    {
        if (native_mode == 4) return stripEmulate ? DCA_OK : exchange_x();
        return gather_vector(dx);
    }

    // ---------------- column-strip decomposition: the two exchanges of an evaluation and the all-gather of the API calls
    // (grouped point-to-point transfers on the context's stream; no reference counterpart)
    // A, before expand: the parameters a rank's columns need but another rank owns -- the pairs (j, i) with j on a LOWER
    // rank travel up, packed densely; the fields of its sites come from rank 0.
    int exchange_x()
    {
        const int dt = (int)sizeof(T) * 8;
        const size_t q2 = (size_t)q * q;
        for (int r = sRank + 1; r < sWorld; ++r) {
            const int ni = siteB[r + 1] - siteB[r];
            hipLaunchKernelGGL((strip_pairs_copy_kernel<T, true>), dim3((unsigned)(Lloc * ni)), dim3(64), 0, ctx->stream, dx, dXsend + xsendOff[r], L, q, cS0, cS1, siteB[r], siteB[r + 1]);
        }
        DCA_TRY(dca_comm_p2p_begin(ctx));
        int rc = DCA_OK;
        if (sRank == 0) { for (int r = 1; r < sWorld && rc == DCA_OK; ++r) rc = dca_comm_p2p_send(ctx, dx + (size_t)siteB[r] * q, (size_t)(siteB[r + 1] - siteB[r]) * q, dt, r); }
        else rc = dca_comm_p2p_recv(ctx, dx + (size_t)cS0 * q, (size_t)Lloc * q, dt, 0);
        for (int r = sRank + 1; r < sWorld && rc == DCA_OK; ++r) rc = dca_comm_p2p_send(ctx, dXsend + xsendOff[r], (size_t)Lloc * (siteB[r + 1] - siteB[r]) * q2, dt, r);
        for (int r = 0; r < sRank && rc == DCA_OK; ++r) rc = dca_comm_p2p_recv(ctx, dXrecv + xrecvOff[r], (size_t)(siteB[r + 1] - siteB[r]) * Lloc * q2, dt, r);
        const int rc2 = dca_comm_p2p_end(ctx);
        if (rc != DCA_OK) return rc;
        DCA_TRY(rc2);
        for (int r = 0; r < sRank; ++r) {
            const int nj = siteB[r + 1] - siteB[r];
            hipLaunchKernelGGL((strip_pairs_copy_kernel<T, false>), dim3((unsigned)(nj * Lloc)), dim3(64), 0, ctx->stream, dx, dXrecv + xrecvOff[r], L, q, siteB[r], siteB[r + 1], cS0, cS1);
        }
        HIP_TRY(hipGetLastError());
        return DCA_OK;
    }
    // B, between scatter and fold: the rows of G that belong to the sites of a LOWER rank travel down (whole rows of this
    // rank's window: one contiguous block per peer), the field gradients of this rank's sites go to rank 0.
    int exchange_g()
    {
        const int dt = (int)sizeof(T) * 8;
        DCA_TRY(dca_comm_p2p_begin(ctx));
        int rc = DCA_OK;
        for (int r = 0; r < sRank && rc == DCA_OK; ++r)
            rc = dca_comm_p2p_send(ctx, dG + (size_t)siteB[r] * q * Cs, (size_t)(siteB[r + 1] - siteB[r]) * q * Cs, dt, r);
        for (int r = sRank + 1; r < sWorld && rc == DCA_OK; ++r)
            rc = dca_comm_p2p_recv(ctx, dGrecv + grecvOff[r], (size_t)Lloc * q * strip_cs(r), dt, r);
        if (rc == DCA_OK) {
            if (sRank > 0) rc = dca_comm_p2p_send(ctx, dg + (size_t)cS0 * q, (size_t)Lloc * q, dt, 0);
            else for (int r = 1; r < sWorld && rc == DCA_OK; ++r) rc = dca_comm_p2p_recv(ctx, dg + (size_t)siteB[r] * q, (size_t)(siteB[r + 1] - siteB[r]) * q, dt, r);
        }
        const int rc2 = dca_comm_p2p_end(ctx);
        if (rc != DCA_OK) return rc;
        return rc2;
    }
    // every rank's owned range of a P-vector to every other rank, in place (get_x / get_g / scores: not on the hot path)
    int strip_allgather(T* v)
    {
        const int dt = (int)sizeof(T) * 8;
        DCA_TRY(dca_comm_p2p_begin(ctx));
        int rc = DCA_OK;
        for (int k = 1; k < sWorld && rc == DCA_OK; ++k) {
            const int to = (sRank + k) % sWorld, from = (sRank - k + sWorld) % sWorld;
            rc = dca_comm_p2p_send(ctx, v + oLo, oHi - oLo, dt, to);
            if (rc == DCA_OK) rc = dca_comm_p2p_recv(ctx, v + owned_lo(from), owned_hi(from) - owned_lo(from), dt, from);
        }
        const int rc2 = dca_comm_p2p_end(ctx);
        if (rc != DCA_OK) return rc;
        return rc2;
    }
    int set_vector_sharding(int rank, int world, dca_comm_hook h, void* user) override
    {
        if (!configured) { dca_set_error("dca_plm_configure first"); return DCA_ERR_STATE; }
        if (o.begun && !o.finished) { dca_set_error("vector sharding cannot change during an optimisation"); return DCA_ERR_STATE; }
        if (strips) { dca_set_error("configured for column strips: dca_plm_configure again first"); return DCA_ERR_STATE; }
        if (native_mode >= 2) native_mode = 0;
        if (!h || world < 1) { vlo = 0; vn = P; Ppad = P; comm = nullptr; comm_user = nullptr; comm_rank = comm_world = 0; return DCA_OK; }
        DCA_TRY(set_slices(rank, world));
        comm = h; comm_user = user; comm_rank = rank; comm_world = world;
        return DCA_OK;
    }
    int set_slices(int rank, int world)
    {
        if (rank < 0 || rank >= world || world > 64) { dca_set_error("bad rank / world"); return DCA_ERR_ARG; }
        const size_t slice = (P + (size_t)world * 4 - 1) / ((size_t)world * 4) * 4;    // multiple of 4 elements: 16-byte aligned slices
        if (slice * world > P + kVecPad) { dca_set_error("world too large for the vector padding"); return DCA_ERR_ARG; }
        Ppad = slice * world;
        vlo = slice * rank;
        vn = vlo >= P ? 0 : std::min(slice, P - vlo);
        return DCA_OK;
    }
    bool configured_for_comm() const override { return configured; }
    int set_native_comm(int mode) override
    {
        if (!configured) { dca_set_error("dca_plm_configure first"); return DCA_ERR_STATE; }
        if (o.begun && !o.finished) { dca_set_error("the exchange scheme cannot change during an optimisation"); return DCA_ERR_STATE; }
        if (mode != 0 && !ctx->comm) { dca_set_error("no communicator: dca_comm_init first"); return DCA_ERR_STATE; }
        if (mode == 4) { dca_set_error("the column-strip decomposition is set up by dca_plm_configure_strips"); return DCA_ERR_ARG; }
        if (mode < 0 || mode > 3) return DCA_ERR_ARG;
        if (strips) {       // the arrays are cut for a column window: another scheme (or none) needs a new configuration
            if (mode != 0) { dca_set_error("configured for column strips: dca_plm_configure again first"); return DCA_ERR_STATE; }
            configured = false; strips = false; native_mode = 0; o = decltype(o)();
            return DCA_OK;
        }
        if (mode >= 2) DCA_TRY(set_slices(ctx->comm_rank, ctx->comm_world));      // validate before anything is dropped
        else { vlo = 0; vn = P; Ppad = P; }
        comm = nullptr; comm_user = nullptr; comm_rank = comm_world = 0;
        if (mode != 0) { hook = nullptr; hook_user = nullptr; }                   // the native exchange replaces the caller's hook
        native_mode = mode;
        return DCA_OK;
    }

    int read_scalars(int n)
    {
        HIP_TRY(hipMemcpyAsync(ctx->hScal, ctx->dScal, (size_t)n * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
        HIP_TRY(hipStreamSynchronize(ctx->stream));
        return DCA_OK;
    }

    int gradient(double* fx_out) override
    {
        DCA_TRY(evaluate_async());
        DCA_TRY(reduce_scalars(0, 1));
        DCA_TRY(gather_vector(dg));
        DCA_TRY(read_scalars(1));
        if (fx_out) *fx_out = ctx->hScal[0];
        return DCA_OK;
    }

    // ---------------- vector helpers
    void v_neg(T* d, const T* g) { hipLaunchKernelGGL(vec_neg_kernel<T>, dim3(kVecBlocks), dim3(kVecThreads), 0, ctx->stream, d + vlo, g + vlo, vn); }
    void v_axpy(T* y, double a, const T* x) { hipLaunchKernelGGL(vec_axpy_kernel<T>, dim3(kVecBlocks), dim3(kVecThreads), 0, ctx->stream, y + vlo, (T)a, x + vlo, vn); }
    void v_scale(T* y, double a) { hipLaunchKernelGGL(vec_scale_kernel<T>, dim3(kVecBlocks), dim3(kVecThreads), 0, ctx->stream, y + vlo, (T)a, vn); }
    void v_step(T* x, const T* xp, double stp, const T* d) { hipLaunchKernelGGL(vec_step_kernel<T>, dim3(kVecBlocks), dim3(kVecThreads), 0, ctx->stream, x + vlo, xp + vlo, (T)stp, d + vlo, vn); }
    int v_copy(T* dst, const T* src) { HIP_TRY(hipMemcpyAsync(dst, src, P * sizeof(T), hipMemcpyDeviceToDevice, ctx->stream)); return DCA_OK; }
    int v_dot(const T* a, const T* b, double* out)
    {
        hipLaunchKernelGGL(vec_dot_kernel<T>, dim3(kVecBlocks), dim3(kVecThreads), 0, ctx->stream, a, b, P, dVecPart);
        hipLaunchKernelGGL(vec_final_kernel, dim3(1), dim3(256), 0, ctx->stream, dVecPart, kVecBlocks, 1, ctx->dScal + 1);
        HIP_TRY(hipMemcpyAsync(ctx->hScal + 1, ctx->dScal + 1, sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
        HIP_TRY(hipStreamSynchronize(ctx->stream));
        *out = ctx->hScal[1];
        return DCA_OK;
    }
    // after an evaluation: fx (slot 0), g.d, x.x, g.g (slots 1..3) in one round trip
    int eval_scalars(double* fx, double* gd, double* xx, double* gg)
    {
        if (fxPending) {
            double* const fxChunks = dRegPart + 2 * (size_t)nRegPart;
            hipLaunchKernelGGL(vec_dot3_fx_kernel<T>, dim3(kVecBlocks + kSumStageBlocks), dim3(kVecThreads), 0, ctx->stream, dg + vlo, dd + vlo, dx + vlo, vn, dVecPart,
                               dRegPart, nRegPart, fxChunks);
            hipLaunchKernelGGL(vec_final_fx_kernel, dim3(3 + 1), dim3(1024), 0, ctx->stream, dVecPart, kVecBlocks, 3, ctx->dScal + 1,
                               fxChunks, kSumStageBlocks, dFxPart, nFxPart, ctx->dScal);
            fxPending = false;
        } else {
            hipLaunchKernelGGL(vec_dot3_kernel<T>, dim3(kVecBlocks), dim3(kVecThreads), 0, ctx->stream, dg + vlo, dd + vlo, dx + vlo, vn, dVecPart);
            hipLaunchKernelGGL(vec_final_kernel, dim3(3), dim3(256), 0, ctx->stream, dVecPart, kVecBlocks, 3, ctx->dScal + 1);
        }
        DCA_TRY(reduce_scalars(0, 4));     // fx (local data term) and the three partial dot products
        DCA_TRY(read_scalars(kSlotDginit + 1));
        *fx = ctx->hScal[0]; *gd = ctx->hScal[1]; *xx = ctx->hScal[2]; *gg = ctx->hScal[3];
        if (o.dginit_on_device) { o.dginit = ctx->hScal[kSlotDginit]; o.dginit_on_device = false; }
        return DCA_OK;
    }

    int lbfgs_begin(int max_iterations, int verbose) override
    {
        if (!configured) { dca_set_error("dca_plm_configure first"); return DCA_ERR_STATE; }
        if (!lbfgs_alloc) {
            DCA_TRY(dalloc(&dxp, P + kVecPad)); DCA_TRY(dalloc(&dgp, P + kVecPad)); DCA_TRY(dalloc(&dd, P + kVecPad));
            for (int i = 0; i < 5; ++i) { DCA_TRY(dalloc(&dS[i], P + kVecPad)); DCA_TRY(dalloc(&dY[i], P + kVecPad)); }
            HIP_TRY(hipMemsetAsync(dxp, 0, (P + kVecPad) * sizeof(T), ctx->stream));
            HIP_TRY(hipMemsetAsync(dgp, 0, (P + kVecPad) * sizeof(T), ctx->stream));
            HIP_TRY(hipMemsetAsync(dd, 0, (P + kVecPad) * sizeof(T), ctx->stream));
            lbfgs_alloc = true;
        }
        for (int i = 0; i < 5; ++i) {   // unused history slots take part in the Gram kernel as zeros
            HIP_TRY(hipMemsetAsync(dS[i], 0, (P + kVecPad) * sizeof(T), ctx->stream));
            HIP_TRY(hipMemsetAsync(dY[i], 0, (P + kVecPad) * sizeof(T), ctx->stream));
        }
        if (!dLb) HIP_TRY(dca_dev_malloc(reinterpret_cast<void**>(&dLb), sizeof(LbfgsDev)));
        HIP_TRY(hipMemsetAsync(dLb, 0, sizeof(LbfgsDev), ctx->stream));
        o = decltype(o)();
        o.max_iterations = max_iterations;
        o.verbose = verbose;
        auto t0 = std::chrono::steady_clock::now();
        DCA_TRY(evaluate_async(true));
        v_neg(dd, dg);
        double fx, gd, xx, gg;
        DCA_TRY(eval_scalars(&fx, &gd, &xx, &gg));
        o.fx = fx;
        o.xnorm = std::sqrt(xx); o.gnorm = std::sqrt(gg);
        const double xn = std::max(o.xnorm, 1.0);
        o.begun = true;
        if (o.gnorm / xn <= 1e-3) { o.status = LB_ALREADY_MINIMIZED; o.finished = true; }
        o.step = 1.0 / std::sqrt(gg);      // 1/|d| with d = -g   (lbfgs.cpp:459)
        o.dginit = -gg;
        o.seconds += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        return DCA_OK;
    }

    // More-Thuente line search (lbfgs.cpp:815-1004) on device vectors.  Returns the number
    // of evaluations (>0) or a libLBFGS error code; *rc_hip carries runtime failures.
    int line_search(double* stp, double* f, double* xx, double* gg, int* rc_hip)
    {
        const double ftol = 1e-4, gtol = 0.9, xtol = 1e-16, min_step = 1e-20, max_step = 1e20;
        const int max_ls = 5;
        int count = 0, uinfo = 0;
        bool brackt = false, stage1 = true;
        // g.d of the direction: known on the host, or still on its way from the device's two-loop recursion -- then it
        // arrives with the scalars of the first evaluation (nothing before that evaluation depends on it)
        double dginit = o.dginit;
        bool have_dginit = !o.dginit_on_device;
        *rc_hip = 0;
        if (*stp <= 0.) return LB_INVALIDPARAMETERS;
        if (have_dginit && 0 < dginit) return LB_INCREASEGRADIENT;
        const double finit = *f;
        double dgtest = ftol * dginit;
        double width = max_step - min_step, prev_width = 2.0 * width;
        LsPoint bx{0., finit, dginit}, by{0., finit, dginit};
        for (;;) {
            double stmin, stmax;
            if (brackt) { stmin = std::min(bx.st, by.st); stmax = std::max(bx.st, by.st); }
            else { stmin = bx.st; stmax = *stp + 4.0 * (*stp - bx.st); }
            if (*stp < min_step) *stp = min_step;
            if (max_step < *stp) *stp = max_step;
            if ((brackt && ((*stp <= stmin || stmax <= *stp) || max_ls <= count + 1 || uinfo != 0)) ||
                (brackt && (stmax - stmin <= xtol * stmax)))
                *stp = bx.st;
            v_step(dx, dxp, *stp, dd);
            DCA_ROUND_STAGE(32, dx, P);
            if ((*rc_hip = publish_x())) return 0;            // sharded vectors: every rank needs the x its evaluation reads
            if ((*rc_hip = evaluate_async(true))) return 0;
            double dg_;
            if ((*rc_hip = eval_scalars(f, &dg_, xx, gg))) return 0;
            if (!have_dginit) {
                have_dginit = true;
                dginit = o.dginit;
                if (0 < dginit) { o.evals -= 1; *f = finit; return LB_INCREASEGRADIENT; }    // the reference returns before evaluating (lbfgs.cpp:858-861); the caller restores x, g
                dgtest = ftol * dginit;
                bx.d = by.d = dginit;
            }
            const double ftest1 = finit + *stp * dgtest;
            ++count;
            if (brackt && ((*stp <= stmin || stmax <= *stp) || uinfo != 0)) return LB_ROUNDING_ERROR;
            if (*stp == max_step && *f <= ftest1 && dg_ <= dgtest) return LB_MAXIMUMSTEP;
            if (*stp == min_step && (ftest1 < *f || dgtest <= dg_)) return LB_MINIMUMSTEP;
            if (brackt && (stmax - stmin) <= xtol * stmax) return LB_WIDTHTOOSMALL;
            if (max_ls <= count) return LB_MAXIMUMLINESEARCH;
            if (*f <= ftest1 && std::fabs(dg_) <= gtol * (-dginit)) return count;
            if (stage1 && *f <= ftest1 && std::min(ftol, gtol) * dginit <= dg_) stage1 = false;
            if (stage1 && ftest1 < *f && *f <= bx.f) {
                LsPoint mx{bx.st, bx.f - bx.st * dgtest, bx.d - dgtest};
                LsPoint my{by.st, by.f - by.st * dgtest, by.d - dgtest};
                uinfo = mt_update(mx, my, *stp, *f - *stp * dgtest, dg_ - dgtest, stmin, stmax, brackt);
                bx = LsPoint{mx.st, mx.f + mx.st * dgtest, mx.d + dgtest};
                by = LsPoint{my.st, my.f + my.st * dgtest, my.d + dgtest};
            } else {
                uinfo = mt_update(bx, by, *stp, *f, dg_, stmin, stmax, brackt);
            }
            if (brackt) {
                if (0.66 * prev_width <= std::fabs(by.st - bx.st)) *stp = bx.st + 0.5 * (by.st - bx.st);
                prev_width = width;
                width = std::fabs(by.st - bx.st);
            }
        }
    }

    void lbfgs_end() override { o = decltype(o)(); }
    int lbfgs_iterate(int iterations, dca_plm_stats* st) override
    {
        if (!o.begun) { dca_set_error("dca_plm_lbfgs_begin first"); return DCA_ERR_STATE; }
        auto t0 = std::chrono::steady_clock::now();
        constexpr int M = 5;
        for (int it = 0; it < iterations && !o.finished; ++it) {
            // the current point becomes (xp, gp) by swapping buffers: the line search writes
            // x = xp + step*d and the evaluation writes g into the other pair (lbfgs.cpp:465-466)
            std::swap(dx, dxp);
            std::swap(dg, dgp);
            double xx = 0, gg = 0;
            int rc = 0;
            const int ls = line_search(&o.step, &o.fx, &xx, &gg, &rc);
            if (rc) return rc;
            if (ls < 0) {   // revert to the previous point (lbfgs.cpp:478-484)
                std::swap(dx, dxp);
                std::swap(dg, dgp);
                o.status = ls; o.finished = true;
                break;
            }
            o.xnorm = std::sqrt(xx); o.gnorm = std::sqrt(gg);
            o.iters = o.k;
            o.last_step = o.step;
            if (o.verbose) {
                fprintf(stderr, "Iteration %d:\n", o.k);
                fprintf(stderr, "fx = %f, xnorm = %f, gnorm = %f, step = %f\n\n", o.fx, o.xnorm, o.gnorm, o.step);
            }
            const double xn = std::max(o.xnorm, 1.0);
            if (o.gnorm / xn <= 1e-3) { o.status = 0; o.finished = true; break; }
            if (o.max_iterations != 0 && o.max_iterations < o.k + 1) { o.status = LB_MAXIMUMITERATION; o.finished = true; break; }

            // s, y of the accepted step and every dot product the direction needs, in one kernel and ONE
            // round trip for the scalars: dScal[1..2] = y.s, y.y;  dScal[3..27] = the 25 Gram entries
            const int e = o.end;              // slot of the newest pair
            VecPtrs5 ptrs;
            for (int i = 0; i < M; ++i) { ptrs.s[i] = dS[i] + vlo; ptrs.y[i] = dY[i] + vlo; }
            {
                ScopedKernelClock kc(ctx, "lbfgs_vec");
                auto gram = [&](auto slot) {
                    hipLaunchKernelGGL((vec_diff_gram_kernel<T, decltype(slot)::value>), dim3(kVecBlocks), dim3(kVecThreads), 0, ctx->stream, ptrs, dS[e] + vlo, dY[e] + vlo,
                                       dx + vlo, dxp + vlo, dg + vlo, dgp + vlo, vn, dVecPart);
                };
                switch (e) {
                case 0: gram(std::integral_constant<int, 0>()); break;
                case 1: gram(std::integral_constant<int, 1>()); break;
                case 2: gram(std::integral_constant<int, 2>()); break;
                case 3: gram(std::integral_constant<int, 3>()); break;
                default: gram(std::integral_constant<int, 4>()); break;
                }
                hipLaunchKernelGGL(vec_final_kernel, dim3(27), dim3(256), 0, ctx->stream, dVecPart, kVecBlocks, 27, ctx->dScal + 1);
            }
            DCA_TRY(reduce_scalars(1, 27));
            const int bound = (M <= o.k) ? M : o.k;
            ++o.k;
            o.end = (o.end + 1) % M;
            {
                // two-loop recursion on the device (no host round trip), then d = cf.g g + sum cf.s_k s_k + cf.y_k y_k
                ScopedKernelClock kc(ctx, "lbfgs_vec");
                hipLaunchKernelGGL(lbfgs_two_loop_kernel, dim3(1), dim3(64), 0, ctx->stream, ctx->dScal, dLb, e, o.end, bound, gg);
                hipLaunchKernelGGL(vec_compose_kernel<T>, dim3(kVecBlocks), dim3(kVecThreads), 0, ctx->stream, dd + vlo, dg + vlo, ptrs, &dLb->cf, vn);
                DCA_ROUND_STAGE(64, dd, P);
                o.dginit_on_device = true;
            }
            o.step = 1.0;
        }
        HIP_TRY(hipStreamSynchronize(ctx->stream));
        o.seconds += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        if (st) {
            st->status = o.status; st->iterations = o.iters; st->evaluations = o.evals; st->finished = o.finished ? 1 : 0;
            st->fx = o.fx; st->xnorm = o.xnorm; st->gnorm = o.gnorm; st->step = o.iters ? o.last_step : o.step; st->seconds = o.seconds;
        }
        return DCA_OK;
    }

    int scores(int apc, double* out) override
    {
        if (!configured) return DCA_ERR_STATE;
        if (native_mode == 4 && !stripEmulate) DCA_TRY(strip_allgather(dx));
        const size_t npairs = (size_t)L * (L - 1) / 2;
        double* dOut = nullptr;
        HIP_TRY(dca_dev_malloc(reinterpret_cast<void**>(&dOut), npairs * sizeof(double)));
        int rc = dca_fn_scores(ctx, dx, 0, (int)sizeof(T) * 8, L, q, 0, apc, dOut);
        if (rc == DCA_OK) {
            // ctx->stream is non-blocking: the null-stream copy below does not wait for it
            hipError_t e = hipStreamSynchronize(ctx->stream);
            if (e == hipSuccess) e = hipMemcpy(out, dOut, npairs * sizeof(double), hipMemcpyDeviceToHost);
            if (e != hipSuccess) { dca_set_error("copy scores: %s", hipGetErrorString(e)); rc = DCA_ERR_HIP; }
        }
        dca_dev_free(dOut);
        return rc;
    }
    // (q-1)x(q-1) blocks of the current x for selected pairs (compute_params, plmdca.py:345-434)
    int pair_couplings(const int* pairs, int npairs, int shift, double* out) override
    {
        if (!configured) return DCA_ERR_STATE;
        if (native_mode == 4 && !stripEmulate) DCA_TRY(strip_allgather(dx));
        return dca_pair_blocks(ctx, dx, 0, (int)sizeof(T) * 8, L, q, 0, pairs, npairs, shift, out);
    }

    // DI of the current x (plmdca.py:683-790); reg_fi: host, L*q regularised single-site frequencies
    int di_scores(const double* reg_fi, int apc, double* out) override
    {
        if (!configured) return DCA_ERR_STATE;
        if (native_mode == 4 && !stripEmulate) DCA_TRY(strip_allgather(dx));
        const size_t npairs = (size_t)L * (L - 1) / 2;
        double *dOut = nullptr, *dFi = nullptr;
        HIP_TRY(dca_dev_malloc(reinterpret_cast<void**>(&dOut), npairs * sizeof(double)));
        if (dca_dev_malloc(reinterpret_cast<void**>(&dFi), (size_t)L * q * sizeof(double)) != hipSuccess) { dca_dev_free(dOut); return DCA_ERR_NOMEM; }
        int rc = DCA_OK;
        if (hipMemcpy(dFi, reg_fi, (size_t)L * q * sizeof(double), hipMemcpyHostToDevice) != hipSuccess) rc = DCA_ERR_HIP;
        if (rc == DCA_OK) rc = dca_di_scores(ctx, dx, 0, (int)sizeof(T) * 8, dFi, L, q, 0, apc, dOut);
        if (rc == DCA_OK) {
            hipError_t e = hipStreamSynchronize(ctx->stream);
            if (e == hipSuccess) e = hipMemcpy(out, dOut, npairs * sizeof(double), hipMemcpyDeviceToHost);
            if (e != hipSuccess) { dca_set_error("copy DI scores: %s", hipGetErrorString(e)); rc = DCA_ERR_HIP; }
        }
        dca_dev_free(dOut); dca_dev_free(dFi);
        return rc;
    }
};

}  // namespace

PlmEngineBase* dca_make_plm_engine(dca_ctx* ctx)
{
    if (ctx->precision == DCA_F64) return new PlmEngine<double>(ctx);
    return new PlmEngine<float>(ctx);
}
