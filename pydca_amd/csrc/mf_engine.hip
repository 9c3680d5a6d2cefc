// mfDCA on MI355X (all float64): weighted single/pair site counts, pseudocount
// regularisation, correlation matrix, couplings = -inv(C), FN/APC scores.
// Reference: pydca/meanfield_dca/msa_numerics.py:53-342 and meanfield_dca.py:902-988.
//
// Pair counts: Craw[(i,a)][(j,b)] = sum_n w_n [x_ni = a][x_nj = b] is a histogram, not a
// dense GEMM (the one-hot matrix has one non-zero per q entries), so it is built as a
// deterministic gather: for every site i the sequences are bucketed by their state a
// (stable counting sort, once per alignment); a workgroup owns one (i,a) row of Craw,
// walks its bucket in ascending n and lets thread j add w_n into its private LDS
// histogram slot [x_nj][j].  N*L^2 LDS read-modify-writes in total, no atomics, and
// Craw comes out exactly symmetric because both (i,a,j,b) and (j,b,i,a) add the same
// weights in the same order.
#include "dca_internal.h"

namespace {

__host__ __device__ __forceinline__ size_t pair_index(int L, int i, int j)
{
    return (size_t)L * (L - 1) / 2 - (size_t)(L - i) * (L - i - 1) / 2 + (size_t)(j - i - 1);
}

// XT[i][n] = X[n][i]  (64x64 byte tiles through LDS; makes per-site passes coalesced)
__global__ __launch_bounds__(256)
void mf_transpose_kernel(const uint8_t* __restrict__ X, uint8_t* __restrict__ XT, int N, int L, int Ls, int Nt)
{
    __shared__ uint8_t tile[64][65];
    const int n0 = blockIdx.x * 64, i0 = blockIdx.y * 64;
    for (int e = threadIdx.x; e < 64 * 64; e += 256) {
        const int r = e / 64, c = e % 64;                 // r: sequence, c: site
        const int n = n0 + r, i = i0 + c;
        tile[r][c] = (n < N && i < L) ? X[(size_t)n * Ls + i] : (uint8_t)0;
    }
    __syncthreads();
    for (int e = threadIdx.x; e < 64 * 64; e += 256) {
        const int c = e / 64, r = e % 64;
        const int n = n0 + r, i = i0 + c;
        if (i < L && n < Nt) XT[(size_t)i * Nt + n] = tile[r][c];
    }
}

// One workgroup per site: stable counting sort of the sequences by their state at that site
// (ascending n inside a state), dominant state, and the weighted single-site counts
// cnt1[i][b] = sum_{n : x_ni = b} w_n summed in ascending n (deterministic).
// Thread t owns the contiguous segment of sequences [t*seg, (t+1)*seg).
constexpr int kSortThreads = 256;

__global__ __launch_bounds__(kSortThreads)
void mf_site_sort_kernel(const uint8_t* __restrict__ XT, const double* __restrict__ w, uint32_t* __restrict__ perm,
                         int* __restrict__ off, uint8_t* __restrict__ dom, double* __restrict__ cnt1,
                         int N, int Nt, int q)
{
    __shared__ int cnts[32][kSortThreads + 1];
    __shared__ int base[33];
    __shared__ double wsum[32][kSortThreads];      // per-thread weighted counts of its segment
    const int i = blockIdx.x, t = threadIdx.x;
    const int seg = (N + kSortThreads - 1) / kSortThreads;
    const int nb = min(N, t * seg), ne = min(N, (t + 1) * seg);
    const uint8_t* col = XT + (size_t)i * Nt;
    int local[32];
    for (int b = 0; b < q; ++b) { local[b] = 0; wsum[b][t] = 0.0; }
    for (int n = nb; n < ne; ++n) local[col[n]]++;
    for (int b = 0; b < q; ++b) cnts[b][t] = local[b];
    __syncthreads();
    if (t < q) {              // exclusive scan over the threads for state t
        int run = 0;
        for (int k = 0; k < kSortThreads; ++k) { const int c = cnts[t][k]; cnts[t][k] = run; run += c; }
        cnts[t][kSortThreads] = run;
    }
    __syncthreads();
    if (t == 0) {
        int run = 0, best = 0;
        for (int b = 0; b < q; ++b) {
            base[b] = run; run += cnts[b][kSortThreads];
            if (cnts[b][kSortThreads] > cnts[best][kSortThreads]) best = b;
        }
        base[q] = run;
        for (int b = 0; b <= q; ++b) off[i * (q + 1) + b] = base[b];
        dom[i] = (uint8_t)best;
    }
    __syncthreads();
    uint32_t* p = perm + (size_t)i * N;
    for (int b = 0; b < q; ++b) local[b] = base[b] + cnts[b][t];
    for (int n = nb; n < ne; ++n) {
        const int b = col[n];
        p[local[b]++] = (uint32_t)n;
        wsum[b][t] += w[n];                         // ascending n inside the segment
    }
    __syncthreads();
    if (t < q) {              // weighted count of state t: segments in ascending order (deterministic)
        double s = 0.0;
        for (int k = 0; k < kSortThreads; ++k) s += wsum[t][k];
        cnt1[i * q + t] = s;
    }
}

constexpr int kCountThreads = 256;

// *p += v on an LDS slot that only this thread touches, as ONE fire-and-forget ds_add_f64 instead of
// ds_read + v_add_f64 + ds_write: consecutive updates may hit the same slot, so the read-modify-write form is a
// chain of LDS round trips; the LDS unit applies the adds in issue order, so the sum is the same sequence of
// IEEE additions.
__device__ __forceinline__ void lds_add(double* p, double v)
{
    typedef __attribute__((address_space(3))) double* lds_ptr;
    __builtin_amdgcn_ds_atomic_fadd_f64((lds_ptr)p, v);
}

// One workgroup per (site i, state a): the upper-triangle part of row (i,a) of Craw,
// Craw[(i,a)][(j,b)] for j > i.  Rows of the site's dominant state are skipped here and
// completed by mf_complete_kernel from the single-site counts:
// sum_a Craw[(i,a)][(j,b)] = cnt1[j][b].
template <bool PREFETCH>
__global__ __launch_bounds__(kCountThreads)
void mf_counts_kernel(const uint8_t* __restrict__ X, const double* __restrict__ w, const uint32_t* __restrict__ perm,
                      const int* __restrict__ off, const uint8_t* __restrict__ dom, double* __restrict__ Craw,
                      int N, int L, int Ls, int q, int ldc)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char dca_smem[];
    double* hist = reinterpret_cast<double*>(dca_smem);       // [q][kCountThreads]
    const int i = blockIdx.x / q, a = blockIdx.x % q;
    if (a == dom[i] || i == L - 1) return;
    const int k0 = off[i * (q + 1) + a], k1 = off[i * (q + 1) + a + 1];
    const uint32_t* p = perm + (size_t)i * N;
    const int t = threadIdx.x;
#ifndef DCA_COUNTS_U
#define DCA_COUNTS_U 32
#endif
    // List entries fetched per batch.  A batch is two dependent memory round trips (list entry -> weight and alignment
    // byte) followed by fire-and-forget LDS adds, and a wave has one batch in flight: the loop time is the round trips
    // divided by the batch size (6.0 / 5.6 / 4.2 ms at config D for 8 / 16 / 32).
    constexpr int U = DCA_COUNTS_U;
    for (int j0 = i + 1; j0 < L; j0 += kCountThreads) {
        const int j = j0 + t;
        for (int b = 0; b < q; ++b) hist[b * kCountThreads + t] = 0.0;
        if (j < L) {
            const uint8_t* Xj = X + j;
            int k = k0;
            // PREFETCH: the list entries of batch b+1 are requested before the weights / alignment bytes of batch b -- one
            // dependent round trip per batch instead of two.  It pays where the registers are free: E (q = 5, 10 KB of
            // LDS per workgroup) 5.5 -> 4.1 ms; D (q = 21: 43 KB of LDS already limit the CU to 12 waves) 4.2 -> 4.6 ms,
            // so the protein alphabet runs without it.
            uint32_t nn[U];
            if (PREFETCH && k + U <= k1) {
#pragma unroll
                for (int u = 0; u < U; ++u) nn[u] = p[k + u];
            }
            for (; k + U <= k1; k += U) {
                uint32_t n[U];
                double wv[U];
                int bb[U];
#pragma unroll
                for (int u = 0; u < U; ++u) n[u] = PREFETCH ? nn[u] : p[k + u];
                if (PREFETCH && k + 2 * U <= k1) {
#pragma unroll
                    for (int u = 0; u < U; ++u) nn[u] = p[k + U + u];
                }
#pragma unroll
                // (round 5, measured, not kept: the weights in list order beside the lists, written by the sort kernel -- one
                // contiguous run per batch instead of U gathered loads: counts 4.1 -> 3.8 ms at D, but the sort kernel's
                // scattered 8-byte stores cost the same 0.3 ms, and for q = 5 the scalar run's s_waitcnt drains the LDS adds)
                for (int u = 0; u < U; ++u) { wv[u] = w[n[u]]; bb[u] = Xj[(size_t)n[u] * Ls]; }
#pragma unroll
                for (int u = 0; u < U; ++u) lds_add(&hist[bb[u] * kCountThreads + t], wv[u]);     // list order: ascending n
            }
            for (; k < k1; ++k) {
                const uint32_t n = p[k];
                lds_add(&hist[Xj[(size_t)n * Ls] * kCountThreads + t], w[n]);
            }
        }
        // the block's part of row (i,a) is one contiguous run of (sites in the block) x q doubles: store it in that
        // order (thread t's own q values are 8-byte pieces 8q bytes apart -- every store instruction would touch
        // 64 different lines)
        __syncthreads();
        const int nj = min(kCountThreads, L - j0);
        double* dst = Craw + (size_t)(i * q + a) * ldc + (size_t)j0 * q;
        for (int e = t; e < nj * q; e += kCountThreads) dst[e] = hist[(e % q) * kCountThreads + e / q];
        __syncthreads();
    }
}

// Completes Craw: dominant-state rows by complement, diagonal blocks from the single-site
// counts, lower triangle by mirroring.  One workgroup per site pair (i <= j).
__global__ __launch_bounds__(64)
void mf_complete_kernel(double* __restrict__ Craw, const double* __restrict__ cnt1, const uint8_t* __restrict__ dom,
                        int L, int q, int ldc)
{
    const int i = blockIdx.y, j = blockIdx.x;
    if (j < i) return;
    const int t = threadIdx.x;
    if (i == j) {
        for (int e = t; e < q * q; e += 64) {
            const int a = e / q, b = e % q;
            Craw[(size_t)(i * q + a) * ldc + i * q + b] = (a == b) ? cnt1[i * q + a] : 0.0;
        }
        return;
    }
    const int da = dom[i];
    if (t < q) {              // column b = t of the block: dominant row = cnt1[j][b] - sum of the others
        double s = 0.0;
        for (int a = 0; a < q; ++a)
            if (a != da) s += Craw[(size_t)(i * q + a) * ldc + (size_t)j * q + t];
        Craw[(size_t)(i * q + da) * ldc + (size_t)j * q + t] = cnt1[j * q + t] - s;
    }
    __syncthreads();
    for (int e = t; e < q * q; e += 64) {
        const int a = e / q, b = e % q;
        Craw[(size_t)(j * q + b) * ldc + (size_t)i * q + a] = Craw[(size_t)(i * q + a) * ldc + (size_t)j * q + b];
    }
}

// f_i(a) = Craw[(i,a)][(i,a)] / Meff   (compute_single_site_freqs, msa_numerics.py:53-89)
__global__ void mf_fi_kernel(const double* __restrict__ Craw, double* __restrict__ fi, int Lq, int ldc, double meff)
{
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c < Lq) fi[c] = Craw[(size_t)c * ldc + c] / meff;
}

// raw pair frequencies, non-gap states, pair order (msa_numerics.py:182-229)
__global__ void mf_fij_export_kernel(const double* __restrict__ Craw, double* __restrict__ fij, int L, int q, int ldc, double meff)
{
    const int qm = q - 1;
    const int i = blockIdx.y, j = blockIdx.x;
    if (j <= i) return;
    const size_t p = pair_index(L, i, j);
    for (int e = threadIdx.x; e < qm * qm; e += blockDim.x) {
        const int a = e / qm, b = e % qm;
        fij[p * qm * qm + e] = Craw[(size_t)(i * q + a) * ldc + (size_t)j * q + b] / meff;
    }
}

// correlation matrix from raw counts: regularisation (msa_numerics.py:92-125, :231-267)
// fused with construct_corr_mat (:270-318).
// Rows/cols >= n (padding up to np) form an identity block.
// fi = the single-site frequencies mf_fi_kernel has already formed (Craw's diagonal / Meff: the same quotient, taken once
// per column instead of twice per element); QT = q when it is one of the two alphabets, so that the column -> (site,
// state) split divides by a constant (0: any q).  1.4 -> 0.8 ms at D with the reads from one row, -> 0.5 ms with this.
constexpr int kCorrRows = 4;
template <int QT>
__global__ void mf_corr_kernel(const double* __restrict__ Craw, const double* __restrict__ fi, double* __restrict__ C, int L, int qrt,
                               int ldc, int np, double meff, double theta)
{
    const int q = QT ? QT : qrt;
    const int qm = q - 1, n = L * qm;
    const int col = blockIdx.x * blockDim.x + threadIdx.x;
    if (col >= np) return;
    // a thread does kCorrRows consecutive rows of its column: the column's site / state split and regularised frequency are
    // formed once, and the rows' Craw loads are independent and in flight together (0.67 -> 0.45 ms at config D)
    const int j = col / qm, b = col % qm;
    const double thq = theta / (double)q;
    const double fjb = col < n ? thq + (1.0 - theta) * fi[j * q + b] : 0.0;
    const int row0 = blockIdx.y * kCorrRows;
    double fij[kCorrRows];
#pragma unroll
    for (int r = 0; r < kCorrRows; ++r) {
        const int row = row0 + r;
        const int i = row / qm, a = row % qm;
        // Craw is bit-symmetric (mf_complete_kernel mirrors it, sums of shards stay symmetric) and the products below
        // commute, so entry (row, col) is computed from Craw's row `row`: coalesced reads for both triangles
        fij[r] = (row < n && col < n && i != j) ? Craw[(size_t)(i * q + a) * ldc + (size_t)j * q + b] : 0.0;
    }
#pragma unroll
    for (int r = 0; r < kCorrRows; ++r) {
        const int row = row0 + r;
        if (row >= np) break;
        double v;
        if (row >= n || col >= n) {
            v = (row == col) ? 1.0 : 0.0;
        } else {
            const int i = row / qm, a = row % qm;
            const double fia = thq + (1.0 - theta) * fi[i * q + a];
            if (i == j) {
                v = (a == b) ? fia * (1.0 - fia) : -1.0 * fia * fjb;
            } else {
                const double rfij = theta / (double)(q * q) + (1.0 - theta) * (fij[r] / meff);
                v = rfij - fia * fjb;
            }
        }
        C[(size_t)row * np + col] = v;
    }
}

// construct_corr_mat from caller-provided regularised frequencies (stage API)
__global__ void mf_corr_from_freqs_kernel(const double* __restrict__ fi, const double* __restrict__ fij,
                                          double* __restrict__ C, int L, int q)
{
    const int qm = q - 1, n = L * qm;
    const int col = blockIdx.x * blockDim.x + threadIdx.x;
    const int row = blockIdx.y;
    if (col >= n) return;
    int r = row, c = col;
    if (r > c) { const int t = r; r = c; c = t; }
    const int i = r / qm, a = r % qm, j = c / qm, b = c % qm;
    double v;
    if (i == j) {
        const double fia = fi[i * q + a], fib = fi[i * q + b];
        v = (a == b) ? fia * (1.0 - fia) : -1.0 * fia * fib;
    } else {
        v = fij[pair_index(L, i, j) * qm * qm + a * qm + b] - fi[i * q + a] * fi[j * q + b];
    }
    C[(size_t)row * n + col] = v;
}


}  // namespace

struct MfEngine {
    dca_ctx* ctx;
    int N, L, q, Ls, Lq, n, np;
    uint32_t* dPerm = nullptr;
    int* dOff = nullptr;
    uint8_t *dXT = nullptr, *dDom = nullptr;
    double* dCnt1 = nullptr;
    double *dCraw = nullptr, *dFi = nullptr, *dC = nullptr, *dJ = nullptr, *dWork = nullptr;
    bool have_counts = false, have_corr = false, have_J = false;
    bool corr_on_device = false;    // dC holds the correlation matrix (the factorisation of dca_mf_engine_couplings destroys it)
    double theta = 0.0;
    double* dRegFi = nullptr;
    dca_reduce_hook hook = nullptr;   // sequence sharding: sums Craw and Meff over the shards
    void* hook_user = nullptr;
    bool native_reduce = false;       // the same sum through ctx->comm (RCCL on the context's stream)
    // Row window (dca_mf_set_row_window): the context holds the WHOLE alignment and all weights, this rank counts the sequences
    // [winFirst, winFirst + winCount) only and the native reduction sums the raw counts over the ranks -- the decomposition of
    // MeanFieldDCA(devices = ...): everything after the counts (and everything that needs the alignment) stays as on one GPU.
    int winFirst = 0, winCount = -1;  // -1: all rows
    bool counts_global = false;       // the cached counts are the sum over all ranks' windows (= the whole alignment's)
    double meff = 0.0;                // Meff the frequencies are normalised by: ctx->meff (this context's weights) summed over
                                      // the shards when a hook / the native reduction is set; ctx->meff itself stays local
    ~MfEngine() { dca_dev_free(dRegFi); dca_dev_free(dPerm); dca_dev_free(dOff); dca_dev_free(dXT); dca_dev_free(dDom); dca_dev_free(dCnt1); dca_dev_free(dCraw); dca_dev_free(dFi); dca_dev_free(dC); dca_dev_free(dWork); }
};

MfEngine* dca_make_mf_engine(dca_ctx* ctx)
{
    MfEngine* m = new MfEngine();
    m->ctx = ctx; m->N = ctx->N; m->L = ctx->L; m->q = ctx->q; m->Ls = ctx->Ls;
    m->Lq = m->L * m->q;
    m->n = m->L * (m->q - 1);
    m->np = (int)round_up((size_t)m->n, 64);
    return m;
}
void dca_free_mf_engine(MfEngine* m) { delete m; }

static int mf_counts(MfEngine* m)
{
    if (m->have_counts) return DCA_OK;
    dca_ctx* ctx = m->ctx;
    if (m->q > 32) { dca_set_error("q too large"); return DCA_ERR_ARG; }
    const int Nt = (int)round_up((size_t)m->N, 64);
    const bool windowed = m->winCount >= 0;
    const int rowFirst = windowed ? m->winFirst : 0, Nw = windowed ? m->winCount : m->N;      // the rows this context counts
    if (windowed && !m->native_reduce) { dca_set_error("a row window needs the native count reduction (dca_mf_set_native_comm)"); return DCA_ERR_STATE; }
    const uint8_t* Xw = ctx->dX + (size_t)rowFirst * m->Ls;
    const double* Ww = ctx->dWd + rowFirst;
    if (!m->dPerm) {
        HIP_TRY(dca_dev_malloc(reinterpret_cast<void**>(&m->dPerm), (size_t)m->L * m->N * sizeof(uint32_t), false));
        HIP_TRY(dca_dev_malloc(reinterpret_cast<void**>(&m->dOff), (size_t)m->L * (m->q + 1) * sizeof(int)));
        HIP_TRY(dca_dev_malloc(reinterpret_cast<void**>(&m->dCraw), (size_t)m->Lq * m->Lq * sizeof(double), false));
        HIP_TRY(dca_dev_malloc(reinterpret_cast<void**>(&m->dFi), (size_t)m->Lq * sizeof(double)));
        HIP_TRY(dca_dev_malloc(reinterpret_cast<void**>(&m->dXT), (size_t)m->L * Nt, false));
        HIP_TRY(dca_dev_malloc(reinterpret_cast<void**>(&m->dDom), (size_t)m->L));
        HIP_TRY(dca_dev_malloc(reinterpret_cast<void**>(&m->dCnt1), (size_t)m->Lq * sizeof(double)));
    }
    {
        ScopedKernelClock kc(ctx, "mf_sort");
        hipLaunchKernelGGL(mf_transpose_kernel, dim3(std::max(1, ceil_div(Nw, 64)), ceil_div(m->L, 64)), dim3(256), 0, ctx->stream,
                           Xw, m->dXT, Nw, m->L, m->Ls, Nt);
        hipLaunchKernelGGL(mf_site_sort_kernel, dim3(m->L), dim3(kSortThreads), 0, ctx->stream, m->dXT, Ww, m->dPerm,
                           m->dOff, m->dDom, m->dCnt1, Nw, Nt, m->q);
    }
    {
        ScopedKernelClock kc(ctx, "mf_counts");
        const size_t lds = (size_t)m->q * kCountThreads * sizeof(double);
        if (m->q <= 8)
            hipLaunchKernelGGL(mf_counts_kernel<true>, dim3(m->L * m->q), dim3(kCountThreads), lds, ctx->stream, Xw, Ww,
                               m->dPerm, m->dOff, m->dDom, m->dCraw, Nw, m->L, m->Ls, m->q, m->Lq);
        else
            hipLaunchKernelGGL(mf_counts_kernel<false>, dim3(m->L * m->q), dim3(kCountThreads), lds, ctx->stream, Xw, Ww,
                               m->dPerm, m->dOff, m->dDom, m->dCraw, Nw, m->L, m->Ls, m->q, m->Lq);
        hipLaunchKernelGGL(mf_complete_kernel, dim3(m->L, m->L), dim3(64), 0, ctx->stream, m->dCraw, m->dCnt1, m->dDom,
                           m->L, m->q, m->Lq);
    }
    m->meff = ctx->meff;
    if (m->hook || m->native_reduce) {
        // the counts are linear in the sequences: shards hold contiguous blocks of the alignment with the
        // GLOBAL weights, the hook sums the Lq x Lq raw counts and the effective sequence number in place
        HIP_TRY(hipMemcpyAsync(ctx->dScal, &m->meff, sizeof(double), hipMemcpyHostToDevice, ctx->stream));
        if (m->native_reduce) {
            // with a row window every rank holds ALL weights: Meff is already the global one (the very number a single GPU computes)
            DCA_TRY(dca_comm_native_reduce(ctx, m->dCraw, (size_t)m->Lq * m->Lq, DCA_F64, windowed ? nullptr : ctx->dScal));
            m->counts_global = windowed;
        } else {
            HIP_TRY(hipStreamSynchronize(ctx->stream));
            if (m->hook(m->hook_user, m->dCraw, (size_t)m->Lq * m->Lq, DCA_F64, ctx->dScal) != 0) {
                dca_set_error("reduce hook failed");
                return DCA_ERR_ARG;
            }
        }
        HIP_TRY(hipMemcpyAsync(&m->meff, ctx->dScal, sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
        HIP_TRY(hipStreamSynchronize(ctx->stream));      // Meff is a kernel argument below
    }
    hipLaunchKernelGGL(mf_fi_kernel, dim3(ceil_div(m->Lq, 256)), dim3(256), 0, ctx->stream, m->dCraw, m->dFi, m->Lq, m->Lq, m->meff);
    HIP_TRY(hipGetLastError());
    m->have_counts = true;
    return DCA_OK;
}

int dca_mf_engine_site_freqs(MfEngine* m, double* fi_out)
{
    DCA_TRY(mf_counts(m));
    HIP_TRY(hipStreamSynchronize(m->ctx->stream));
    HIP_TRY(hipMemcpy(fi_out, m->dFi, (size_t)m->Lq * sizeof(double), hipMemcpyDeviceToHost));
    return DCA_OK;
}

int dca_mf_engine_pair_freqs(MfEngine* m, double* fij_out)
{
    DCA_TRY(mf_counts(m));
    const int qm = m->q - 1;
    const size_t total = (size_t)m->L * (m->L - 1) / 2 * qm * qm;
    double* dOut = nullptr;
    HIP_TRY(dca_dev_malloc(reinterpret_cast<void**>(&dOut), total * sizeof(double)));
    hipLaunchKernelGGL(mf_fij_export_kernel, dim3(m->L, m->L), dim3(64), 0, m->ctx->stream, m->dCraw, dOut, m->L, m->q, m->Lq, m->meff);
    hipError_t e = hipStreamSynchronize(m->ctx->stream);
    if (e == hipSuccess) e = hipMemcpy(fij_out, dOut, total * sizeof(double), hipMemcpyDeviceToHost);
    dca_dev_free(dOut);
    if (e != hipSuccess) { dca_set_error("pair freqs: %s", hipGetErrorString(e)); return DCA_ERR_HIP; }
    return DCA_OK;
}

static int copy_out_square(MfEngine* m, const double* dSrc, double* out)
{
    // device matrices are np x np; hand back the leading n x n
    HIP_TRY(hipStreamSynchronize(m->ctx->stream));
    HIP_TRY(hipMemcpy2D(out, (size_t)m->n * sizeof(double), dSrc, (size_t)m->np * sizeof(double),
                        (size_t)m->n * sizeof(double), (size_t)m->n, hipMemcpyDeviceToHost));
    return DCA_OK;
}

static int mf_build_corr(MfEngine* m, double theta)
{
    dca_ctx* ctx = m->ctx;
    if (!m->dC) HIP_TRY(dca_dev_malloc(reinterpret_cast<void**>(&m->dC), (size_t)m->np * m->np * sizeof(double), false));
    dim3 grid(ceil_div(m->np, 256), ceil_div(m->np, kCorrRows));
    if (m->q == 21) hipLaunchKernelGGL(mf_corr_kernel<21>, grid, dim3(256), 0, ctx->stream, m->dCraw, m->dFi, m->dC, m->L, m->q, m->Lq, m->np, m->meff, theta);
    else if (m->q == 5) hipLaunchKernelGGL(mf_corr_kernel<5>, grid, dim3(256), 0, ctx->stream, m->dCraw, m->dFi, m->dC, m->L, m->q, m->Lq, m->np, m->meff, theta);
    else hipLaunchKernelGGL(mf_corr_kernel<0>, grid, dim3(256), 0, ctx->stream, m->dCraw, m->dFi, m->dC, m->L, m->q, m->Lq, m->np, m->meff, theta);
    HIP_TRY(hipGetLastError());
    m->corr_on_device = true;
    return DCA_OK;
}

int dca_mf_engine_corr(MfEngine* m, double theta, double* corr_out)
{
    DCA_TRY(mf_counts(m));
    DCA_TRY(mf_build_corr(m, theta));
    m->have_corr = true;
    m->have_J = false;
    m->theta = theta;
    if (corr_out) return copy_out_square(m, m->dC, corr_out);
    return DCA_OK;
}

int dca_mf_engine_couplings(MfEngine* m, double* out)
{
    if (!m->have_corr) { dca_set_error("dca_mf_corr_mat first"); return DCA_ERR_STATE; }
    dca_ctx* ctx = m->ctx;
    const size_t nn = (size_t)m->np * m->np;
    // The factorisation runs in place on the correlation matrix (1.4 ms to rebuild from the counts if it is asked
    // for again) and leaves -inv(C) in the second half of the workspace: no copy in, no copy out, no negation pass.
    if (!m->corr_on_device) DCA_TRY(mf_build_corr(m, m->theta));
    if (!m->dWork) HIP_TRY(dca_dev_malloc(reinterpret_cast<void**>(&m->dWork), 2 * nn * sizeof(double), false));
    int info = 0;
    m->corr_on_device = false;
    m->have_J = false;
    DCA_TRY(dca_spd_inverse_device(ctx, m->dC, m->np, m->dWork, &info, -1.0, &m->dJ));
    if (info != 0) {
        dca_set_error("Singular matrix: the correlation matrix is not positive definite (pivot %d)", info);
        return DCA_ERR_NOT_SPD;
    }
    m->have_J = true;
    if (out) return copy_out_square(m, m->dJ, out);
    return DCA_OK;
}

int dca_mf_engine_scores(MfEngine* m, int apc, double* out)
{
    if (!m->have_J) { dca_set_error("dca_mf_couplings first"); return DCA_ERR_STATE; }
    const size_t npairs = (size_t)m->L * (m->L - 1) / 2;
    double* dOut = nullptr;
    HIP_TRY(dca_dev_malloc(reinterpret_cast<void**>(&dOut), npairs * sizeof(double)));
    int rc = dca_fn_scores(m->ctx, m->dJ, 1, DCA_F64, m->L, m->q, m->np, apc, dOut);
    if (rc == DCA_OK) {
        hipError_t e = hipStreamSynchronize(m->ctx->stream);
        if (e == hipSuccess) e = hipMemcpy(out, dOut, npairs * sizeof(double), hipMemcpyDeviceToHost);
        if (e != hipSuccess) { dca_set_error("scores: %s", hipGetErrorString(e)); rc = DCA_ERR_HIP; }
    }
    dca_dev_free(dOut);
    return rc;
}

// stage API: construct_corr_mat on caller-provided regularised frequencies
extern "C" int dca_mf_corr_from_freqs(dca_ctx* ctx, const double* reg_fi, const double* reg_fij, int L, int q, double* corr_out)
{
    if (!ctx || !reg_fi || !reg_fij || !corr_out || L < 2 || q < 2) { dca_set_error("dca_mf_corr_from_freqs: bad arguments"); return DCA_ERR_ARG; }
    HIP_TRY(hipSetDevice(ctx->device));
    const int qm = q - 1, n = L * qm;
    const size_t nfij = (size_t)L * (L - 1) / 2 * qm * qm;
    double *dFi = nullptr, *dFij = nullptr, *dC = nullptr;
    int rc = DCA_OK;
    if (dca_dev_malloc(reinterpret_cast<void**>(&dFi), (size_t)L * q * sizeof(double)) != hipSuccess ||
        dca_dev_malloc(reinterpret_cast<void**>(&dFij), nfij * sizeof(double)) != hipSuccess ||
        dca_dev_malloc(reinterpret_cast<void**>(&dC), (size_t)n * n * sizeof(double)) != hipSuccess) {
        dca_set_error("out of device memory");
        rc = DCA_ERR_NOMEM;
    }
    if (rc == DCA_OK) {
        hipMemcpy(dFi, reg_fi, (size_t)L * q * sizeof(double), hipMemcpyHostToDevice);
        hipMemcpy(dFij, reg_fij, nfij * sizeof(double), hipMemcpyHostToDevice);
        dim3 grid(ceil_div(n, 256), n);
        hipLaunchKernelGGL(mf_corr_from_freqs_kernel, grid, dim3(256), 0, ctx->stream, dFi, dFij, dC, L, q);
        hipError_t e = hipStreamSynchronize(ctx->stream);
        if (e == hipSuccess) e = hipMemcpy(corr_out, dC, (size_t)n * n * sizeof(double), hipMemcpyDeviceToHost);
        if (e != hipSuccess) { dca_set_error("corr_from_freqs: %s", hipGetErrorString(e)); rc = DCA_ERR_HIP; }
    }
    dca_dev_free(dFi); dca_dev_free(dFij); dca_dev_free(dC);
    return rc;
}

namespace {
// (1 - theta) f + theta / q   (get_reg_single_site_freqs, msa_numerics.py:92-125)
__global__ void mf_regfi_kernel(const double* __restrict__ fi, double* __restrict__ reg, int Lq, int q, double theta)
{
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c < Lq) reg[c] = theta / (double)q + (1.0 - theta) * fi[c];
}
}  // namespace

// DI / DI_APC of the mean-field couplings (meanfield_dca.py:793-899)
int dca_mf_engine_di(MfEngine* m, int apc, double* out)
{
    if (!m->have_J) { dca_set_error("dca_mf_couplings first"); return DCA_ERR_STATE; }
    dca_ctx* ctx = m->ctx;
    if (!m->dRegFi) HIP_TRY(dca_dev_malloc(reinterpret_cast<void**>(&m->dRegFi), (size_t)m->Lq * sizeof(double)));
    hipLaunchKernelGGL(mf_regfi_kernel, dim3(ceil_div(m->Lq, 256)), dim3(256), 0, ctx->stream, m->dFi, m->dRegFi, m->Lq, m->q, m->theta);
    const size_t npairs = (size_t)m->L * (m->L - 1) / 2;
    double* dOut = nullptr;
    HIP_TRY(dca_dev_malloc(reinterpret_cast<void**>(&dOut), npairs * sizeof(double)));
    int rc = dca_di_scores(ctx, m->dJ, 1, DCA_F64, m->dRegFi, m->L, m->q, m->np, apc, dOut);
    if (rc == DCA_OK) {
        hipError_t e = hipStreamSynchronize(ctx->stream);
        if (e == hipSuccess) e = hipMemcpy(out, dOut, npairs * sizeof(double), hipMemcpyDeviceToHost);
        if (e != hipSuccess) { dca_set_error("DI scores: %s", hipGetErrorString(e)); rc = DCA_ERR_HIP; }
    }
    dca_dev_free(dOut);
    return rc;
}

namespace {
// fields_i[a] = log(f_i(a) / f_i(q)) - sum_{j != i} sum_b J[(i,a)][(j,b)] f_j(b)   (compute_fields,
// meanfield_dca.py:588-633).  One workgroup per row (i,a) of J; the diagonal block is skipped.
__global__ __launch_bounds__(256)
void mf_fields_kernel(const double* __restrict__ J, const double* __restrict__ regfi, int L, int q, int ld, double* __restrict__ out)
{
    __shared__ double red[256];
    const int qm = q - 1;
    const int r = blockIdx.x, i = r / qm, a = r % qm;
    double s = 0.0;
    for (int c = threadIdx.x; c < L * qm; c += 256) {
        const int j = c / qm, b = c % qm;
        if (j != i) s += J[(size_t)r * ld + c] * regfi[j * q + b];
    }
    red[threadIdx.x] = s;
    __syncthreads();
    for (int st = 128; st > 0; st >>= 1) {
        if ((int)threadIdx.x < st) red[threadIdx.x] += red[threadIdx.x + st];
        __syncthreads();
    }
    if (threadIdx.x == 0) out[r] = log(regfi[i * q + a] / regfi[i * q + qm]) - red[0];
}
}  // namespace

int dca_mf_engine_fields(MfEngine* m, double* out)
{
    if (!m->have_J) { dca_set_error("dca_mf_couplings first"); return DCA_ERR_STATE; }
    dca_ctx* ctx = m->ctx;
    if (!m->dRegFi) HIP_TRY(dca_dev_malloc(reinterpret_cast<void**>(&m->dRegFi), (size_t)m->Lq * sizeof(double)));
    hipLaunchKernelGGL(mf_regfi_kernel, dim3(ceil_div(m->Lq, 256)), dim3(256), 0, ctx->stream, m->dFi, m->dRegFi, m->Lq, m->q, m->theta);
    const int n = m->L * (m->q - 1);
    double* dOut = nullptr;
    HIP_TRY(dca_dev_malloc(reinterpret_cast<void**>(&dOut), (size_t)n * sizeof(double)));
    hipLaunchKernelGGL(mf_fields_kernel, dim3(n), dim3(256), 0, ctx->stream, m->dJ, m->dRegFi, m->L, m->q, m->np, dOut);
    hipError_t e = hipStreamSynchronize(ctx->stream);
    if (e == hipSuccess) e = hipMemcpy(out, dOut, (size_t)n * sizeof(double), hipMemcpyDeviceToHost);
    dca_dev_free(dOut);
    if (e != hipSuccess) { dca_set_error("fields: %s", hipGetErrorString(e)); return DCA_ERR_HIP; }
    return DCA_OK;
}

int dca_mf_engine_pair_couplings(MfEngine* m, const int* pairs, int npairs, int shift, double* out)
{
    if (!m->have_J) { dca_set_error("dca_mf_couplings first"); return DCA_ERR_STATE; }
    return dca_pair_blocks(m->ctx, m->dJ, 1, DCA_F64, m->L, m->q, m->np, pairs, npairs, shift, out);
}

// Counts, frequencies, correlation matrix and couplings cached so far were summed under the previous exchange scheme (or the
// previous weights): they are recomputed by the next query instead of answering for a single shard.
void dca_mf_engine_invalidate(MfEngine* m) { m->have_counts = m->have_corr = m->have_J = m->corr_on_device = false; m->counts_global = false; }
int dca_mf_engine_set_row_window(MfEngine* m, int first, int count)
{
    if (count < 0) { first = 0; count = -1; }
    else if (first < 0 || (long long)first + count > m->N) { dca_set_error("row window outside the alignment"); return DCA_ERR_ARG; }
    if (first != m->winFirst || count != m->winCount) dca_mf_engine_invalidate(m);
    m->winFirst = first; m->winCount = count;
    return DCA_OK;
}
void dca_mf_engine_set_native(MfEngine* m, bool on)
{
    if (!on && m->native_reduce && m->have_counts && m->counts_global) {
        // the communicator goes away AFTER the windows were summed: what is cached are the whole alignment's counts, which is what
        // this context -- it holds the whole alignment -- would count alone; keep them, drop the window
        m->native_reduce = false; m->winFirst = 0; m->winCount = -1;
        return;
    }
    if (on != m->native_reduce || (on && m->hook)) dca_mf_engine_invalidate(m);
    m->native_reduce = on;
    if (on) { m->hook = nullptr; m->hook_user = nullptr; }
}
void dca_mf_engine_set_hook(MfEngine* m, dca_reduce_hook hook, void* user)
{
    m->hook = hook;
    m->hook_user = user;
    m->native_reduce = false;
    dca_mf_engine_invalidate(m);
}
