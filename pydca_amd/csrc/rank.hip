// Ranking of a score vector on the device: the sorted(..., reverse=True) of compute_sorted_FN /
// _APC / DI (meanfield_dca.py:941, plmdca.py:479) as a stable descending radix sort of
// (score, pair index) pairs -- equal scores keep ascending pair order, exactly what Python's stable
// sort does on the reference's pair-ordered list.  rocPRIM device radix sort (plain library sort;
// 125 k keys at config D, ~0.1 ms against 7.5 ms for numpy's argsort on the host).
#include <cstring>

#include <rocprim/device/device_radix_sort.hpp>

#include "dca_internal.h"

namespace {
__global__ void iota_kernel(int32_t* v, int n)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) v[i] = i;
}
}  // namespace

int dca_scores_order_device(dca_ctx* ctx, const double* dScores, int n, int32_t* order_out)
{
    if (n <= 0) return DCA_OK;
    double* dKeysOut = nullptr;
    int32_t *dIdx = nullptr, *dIdxOut = nullptr;
    void* dTemp = nullptr;
    size_t tempBytes = 0;
    int rc = DCA_OK;
    hipError_t e = dca_dev_malloc(reinterpret_cast<void**>(&dKeysOut), (size_t)n * sizeof(double));
    if (e == hipSuccess) e = dca_dev_malloc(reinterpret_cast<void**>(&dIdx), (size_t)n * sizeof(int32_t));
    if (e == hipSuccess) e = dca_dev_malloc(reinterpret_cast<void**>(&dIdxOut), (size_t)n * sizeof(int32_t));
    if (e == hipSuccess) {
        hipLaunchKernelGGL(iota_kernel, dim3(ceil_div(n, 256)), dim3(256), 0, ctx->stream, dIdx, n);
        e = rocprim::radix_sort_pairs_desc(nullptr, tempBytes, dScores, dKeysOut, dIdx, dIdxOut, (size_t)n, 0, 64, ctx->stream);
    }
    if (e == hipSuccess) e = dca_dev_malloc(&dTemp, std::max<size_t>(tempBytes, 16));
    if (e == hipSuccess) e = rocprim::radix_sort_pairs_desc(dTemp, tempBytes, dScores, dKeysOut, dIdx, dIdxOut, (size_t)n, 0, 64, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    if (e == hipSuccess) e = hipMemcpy(order_out, dIdxOut, (size_t)n * sizeof(int32_t), hipMemcpyDeviceToHost);
    if (e != hipSuccess) { dca_set_error("ranking scores: %s", hipGetErrorString(e)); rc = DCA_ERR_HIP; }
    dca_dev_free(dKeysOut); dca_dev_free(dIdx); dca_dev_free(dIdxOut); dca_dev_free(dTemp);
    return rc;
}
