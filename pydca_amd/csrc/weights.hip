// Sequence weights on MI355X: count_n = #{m : ident(n,m) >= T}, w_n = 1/count_n.
// Reference: PlmDCA::computeSeqsWeight (pydca/plmdca/plmdca_numerics.cpp:611-671, float
// compare) and compute_sequences_weight (pydca/meanfield_dca/msa_numerics.py:13-50, double
// compare).  The reference's floating-point test  ident/L > seqid  is monotone in the
// integer ident, so it is evaluated once on the host, in the reference's precision, for
// ident = 0..L and turned into an integer threshold T: the kernel is integer-exact.
//
// Integer/byte work, VALU-bound (N^2 L byte compares against N L bytes of input):
// 64x64 sequence tiles (upper triangle of tile pairs only), rows staged in LDS as dwords
// (4 sites), a 4x4 register block per thread, mismatches counted with xor / add 0x7f7f7f7f / and 0x80808080 / popcount
// (states are < 32 so bytes never carry).
#include "dca_internal.h"

namespace {

constexpr int kTile = 64;      // sequences per tile side
constexpr int kKD = 32;        // dwords (128 sites) per LDS stage
constexpr int kLdsStride = kKD + 1;

__global__ __launch_bounds__(256)
void weights_count_kernel(const uint8_t* __restrict__ X, uint32_t* __restrict__ counts, int N, int L, int Ls, int thresh)
{
    __shared__ uint32_t sA[kTile * kLdsStride];
    __shared__ uint32_t sB[kTile * kLdsStride];
    __shared__ unsigned sCol[kTile];
    // identity is symmetric: only tile pairs with column tile >= row tile are computed; an
    // off-diagonal tile also credits its columns' sequences (rows of the mirrored tile)
    if (blockIdx.x < blockIdx.y) return;
    const bool offDiag = blockIdx.x > blockIdx.y;
    const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
    const int rowBase = blockIdx.y * kTile, colBase = blockIdx.x * kTile;
    if (threadIdx.x < kTile) sCol[threadIdx.x] = 0;
    const int dwords = Ls / 4;
    unsigned mism[4][4];
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int c = 0; c < 4; ++c) mism[r][c] = 0;

    for (int k0 = 0; k0 < dwords; k0 += kKD) {
        __syncthreads();
        for (int t = threadIdx.x; t < kTile * kKD; t += 256) {
            const int r = t / kKD, k = t % kKD;
            uint32_t a = 0, b = 0;
            if (k0 + k < dwords) {
                if (rowBase + r < N) a = reinterpret_cast<const uint32_t*>(X + (size_t)(rowBase + r) * Ls)[k0 + k];
                if (colBase + r < N) b = reinterpret_cast<const uint32_t*>(X + (size_t)(colBase + r) * Ls)[k0 + k];
            }
            sA[r * kLdsStride + k] = a;
            sB[r * kLdsStride + k] = b;
        }
        __syncthreads();
#pragma unroll 4
        for (int k = 0; k < kKD; ++k) {
            uint32_t a[4], b[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) a[r] = sA[(ty + 16 * r) * kLdsStride + k];
#pragma unroll
            for (int c = 0; c < 4; ++c) b[c] = sB[(tx + 16 * c) * kLdsStride + k];
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int c = 0; c < 4; ++c)
                    mism[r][c] += __popc(((a[r] ^ b[c]) + 0x7f7f7f7fu) & 0x80808080u);
        }
    }
    // ident = L - mismatches (padding bytes are 0 in every row and never mismatch)
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
    if (offDiag) {
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

int dca_weights_compute(dca_ctx* ctx, double seqid, int compare_precision)
{
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
        dim3 grid(ceil_div(N, kTile), ceil_div(N, kTile));
        hipLaunchKernelGGL(weights_count_kernel, grid, dim3(256), 0, ctx->stream, ctx->dX, ctx->dCounts, N, L, ctx->Ls, thresh);
    }
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
