// Frobenius-norm (+ average-product-corrected) scores of coupling blocks.
// Reference: PlmDCA.get_couplings_no_gap_state / compute_sorted_FN / compute_sorted_FN_APC
// (pydca/plmdca/plmdca.py:246-268, :437-524) and MeanFieldDCA.compute_sorted_FN[_APC]
// (pydca/meanfield_dca/meanfield_dca.py:902-988).  The gap row/column is dropped, the
// (q-1)x(q-1) block is double-centred, FN = sqrt(sum J'^2); APC subtracts av_i*av_j/av.
// HBM-bound (each coupling is read once); float64 arithmetic for both paths.
#include "dca_internal.h"

namespace {

__host__ __device__ __forceinline__ size_t pair_index(int L, int i, int j)
{
    return (size_t)L * (L - 1) / 2 - (size_t)(L - i) * (L - i - 1) / 2 + (size_t)(j - i - 1);
}

// one 64-lane workgroup per site pair; block values go through LDS
template <typename S>
__global__ __launch_bounds__(64)
void fn_kernel(const S* __restrict__ src, int kind, int L, int q, int ld, double* __restrict__ fn)
{
    __shared__ double blk[20 * 20];
    __shared__ double rowm[20], colm[20];
    __shared__ double tot;
    const int qm = q - 1;
    // decode (i,j) from blockIdx: rows of the upper triangle
    const size_t p = blockIdx.x;
    int i = 0;
    {
        // largest i with pair_index(L,i,i+1) <= p  (solve the quadratic, then fix up)
        const double Ld = (double)L;
        double t = (2.0 * Ld - 1.0 - sqrt((2.0 * Ld - 1.0) * (2.0 * Ld - 1.0) - 8.0 * (double)p)) / 2.0;
        i = (int)t;
        if (i < 0) i = 0;
        if (i > L - 2) i = L - 2;
        while (i > 0 && pair_index(L, i, i + 1) > p) --i;
        while (i < L - 2 && pair_index(L, i + 1, i + 2) <= p) ++i;
    }
    const int j = (int)(p - pair_index(L, i, i + 1)) + i + 1;
    const int t = threadIdx.x;
    for (int e = t; e < qm * qm; e += 64) {
        const int a = e / qm, b = e % qm;
        double v;
        if (kind == 0) v = (double)src[(size_t)L * q + p * (size_t)q * q + (size_t)a * q + b];
        else v = (double)src[(size_t)(i * qm + a) * ld + (size_t)j * qm + b];
        blk[e] = v;
    }
    __syncthreads();
    if (t < qm) {
        double s = 0;
        for (int b = 0; b < qm; ++b) s += blk[t * qm + b];
        rowm[t] = s / qm;           // mean over b (axis=1)
        double c = 0;
        for (int a = 0; a < qm; ++a) c += blk[a * qm + t];
        colm[t] = c / qm;           // mean over a (axis=0)
    }
    __syncthreads();
    if (t == 0) {
        double s = 0;
        for (int e = 0; e < qm * qm; ++e) s += blk[e];
        tot = s / (double)(qm * qm);
    }
    __syncthreads();
    double acc = 0;
    for (int e = t; e < qm * qm; e += 64) {
        const int a = e / qm, b = e % qm;
        const double v = blk[e] - colm[b] - rowm[a] + tot;
        acc += v * v;
    }
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off);
    if (t == 0) fn[p] = sqrt(acc);
}

// av[i] = sum_{j != i} FN_ij / (L-1)
__global__ __launch_bounds__(256)
void apc_site_kernel(const double* __restrict__ fn, int L, double* __restrict__ av)
{
    __shared__ double red[256];
    const int i = blockIdx.x;
    double s = 0;
    for (int j = threadIdx.x; j < L; j += 256) {
        if (j == i) continue;
        const int a = j < i ? j : i, b = j < i ? i : j;
        s += fn[pair_index(L, a, b)];
    }
    red[threadIdx.x] = s;
    __syncthreads();
    for (int st = 128; st > 0; st >>= 1) {
        if ((int)threadIdx.x < st) red[threadIdx.x] += red[threadIdx.x + st];
        __syncthreads();
    }
    if (threadIdx.x == 0) av[i] = red[0] / (double)(L - 1);
}

__global__ __launch_bounds__(256)
void apc_mean_kernel(const double* __restrict__ av, int L, double* __restrict__ out)
{
    __shared__ double red[256];
    double s = 0;
    for (int i = threadIdx.x; i < L; i += 256) s += av[i];
    red[threadIdx.x] = s;
    __syncthreads();
    for (int st = 128; st > 0; st >>= 1) {
        if ((int)threadIdx.x < st) red[threadIdx.x] += red[threadIdx.x + st];
        __syncthreads();
    }
    if (threadIdx.x == 0) out[0] = red[0] / (double)L;
}

// FN_ij - av_i * (av_j / av)    (same association as the reference expression)
__global__ void apc_apply_kernel(double* __restrict__ fn, const double* __restrict__ av, const double* __restrict__ avAll, int L)
{
    const int i = blockIdx.x;
    for (int j = i + 1 + threadIdx.x; j < L; j += blockDim.x) {
        const size_t p = pair_index(L, i, j);
        fn[p] = fn[p] - av[i] * (av[j] / avAll[0]);
    }
}

// Direct information of one site pair (meanfield_dca/msa_numerics.py:378-533 and the plmDCA twin
// plmdca/msa_numerics.py:156-311): two-site model fields by the reference's fixed-point iteration
// (both fields updated from the old ones, stop when the largest change is <= 1e-4), direct
// probability P_dir = exp(J_ij) h_i h_j / Z with the gap row/column of J set to 0, and
// DI = sum over non-gap (a,b) of (P_dir + eps) log((P_dir + eps) / (f_i f_j + eps)), eps = 1e-20.
template <typename S>
__global__ __launch_bounds__(64)
void di_kernel(const S* __restrict__ src, int kind, const double* __restrict__ regfi, int L, int q, int ld,
               double* __restrict__ di, double* __restrict__ fields, const double* __restrict__ fields_in = nullptr)
{
    __shared__ double E[21 * 21];
    __shared__ double fi[21], fj[21], hi[21], hj[21], ni[21], nj[21];
    __shared__ double change;
    const int qm = q - 1;
    const size_t p = blockIdx.x;
    int i = 0;
    {
        const double Ld = (double)L;
        double t = (2.0 * Ld - 1.0 - sqrt((2.0 * Ld - 1.0) * (2.0 * Ld - 1.0) - 8.0 * (double)p)) / 2.0;
        i = (int)t;
        if (i < 0) i = 0;
        if (i > L - 2) i = L - 2;
        while (i > 0 && pair_index(L, i, i + 1) > p) --i;
        while (i < L - 2 && pair_index(L, i + 1, i + 2) <= p) ++i;
    }
    const int j = (int)(p - pair_index(L, i, i + 1)) + i + 1;
    const int t = threadIdx.x;
    for (int e = t; e < q * q; e += 64) {
        const int a = e / q, b = e % q;
        double v = 0.0;
        if (a < qm && b < qm) {
            if (kind == 0) v = (double)src[(size_t)L * q + p * (size_t)q * q + (size_t)a * q + b];
            else if (kind == 1) v = (double)src[(size_t)(i * qm + a) * ld + (size_t)j * qm + b];
            else v = (double)src[p * (size_t)qm * qm + (size_t)a * qm + b];      // gap-stripped blocks, pair order
        }
        E[e] = exp(v);
    }
    if (t < q) { fi[t] = regfi[i * q + t]; fj[t] = regfi[j * q + t]; hi[t] = hj[t] = 1.0 / (double)q; }
    if (fields_in && t < q) {      // the caller's two-site model fields (compute_direct_info's fields_ij argument)
        hi[t] = fields_in[(p * 2 + 0) * q + t];
        hj[t] = fields_in[(p * 2 + 1) * q + t];
    }
    __syncthreads();
    for (int iter = 0; iter < (fields_in ? 0 : 100000); ++iter) {
        if (t < q) {
            double x = 0.0;
            for (int b = 0; b < q; ++b) x += E[t * q + b] * hj[b];
            ni[t] = fi[t] / x;
        } else if (t >= 32 && t < 32 + q) {
            const int b = t - 32;
            double x = 0.0;
            for (int a = 0; a < q; ++a) x += E[a * q + b] * hi[a];
            nj[b] = fj[b] / x;
        }
        __syncthreads();
        if (t == 0) {
            double si = 0.0, sj = 0.0;
            for (int a = 0; a < q; ++a) { si += ni[a]; sj += nj[a]; }
            double mx = 0.0;
            for (int a = 0; a < q; ++a) {
                const double vi = ni[a] / si, vj = nj[a] / sj;
                mx = fmax(mx, fmax(fabs(vi - hi[a]), fabs(vj - hj[a])));
                hi[a] = vi; hj[a] = vj;
            }
            change = mx;
        }
        __syncthreads();
        if (!(change > 1.0e-4)) break;
    }
    if (fields && t < q) {     // two-site model fields, layout [pair][0 = site i, 1 = site j][q]
        fields[(p * 2 + 0) * q + t] = hi[t];
        fields[(p * 2 + 1) * q + t] = hj[t];
    }
    // direct probability and DI
    double zpart = 0.0;
    for (int e = t; e < q * q; e += 64) zpart += E[e] * hi[e / q] * hj[e % q];
    for (int off = 32; off > 0; off >>= 1) zpart += __shfl_down(zpart, off);
    const double Z = __shfl(zpart, 0);
    double acc = 0.0;
    for (int e = t; e < qm * qm; e += 64) {
        const int a = e / qm, b = e % qm;
        const double pd = E[a * q + b] * hi[a] * hj[b] / Z + 1.0e-20;
        acc += pd * log(pd / (fi[a] * fj[b] + 1.0e-20));
    }
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off);
    if (t == 0) di[p] = acc;
}

// Coupling blocks of selected site pairs, optionally shifted to the zero-sum gauge
// (shift_couplings, meanfield_dca.py:636-658 / plmdca.py:320-342): J - rowmean - colmean + mean.
// One 64-lane workgroup per requested pair; out[p][(q-1)*(q-1)] doubles.
template <typename S>
__global__ __launch_bounds__(64)
void pair_blocks_kernel(const S* __restrict__ src, int kind, const int* __restrict__ pairs, int L, int q, int ld, int shift,
                        double* __restrict__ out)
{
    __shared__ double blk[20 * 20];
    __shared__ double rowm[20], colm[20];
    __shared__ double tot;
    const int qm = q - 1;
    const int i = pairs[2 * blockIdx.x], j = pairs[2 * blockIdx.x + 1];
    const size_t p = pair_index(L, i, j);
    const int t = threadIdx.x;
    for (int e = t; e < qm * qm; e += 64) {
        const int a = e / qm, b = e % qm;
        double v;
        if (kind == 0) v = (double)src[(size_t)L * q + p * (size_t)q * q + (size_t)a * q + b];
        else v = (double)src[(size_t)(i * qm + a) * ld + (size_t)j * qm + b];
        blk[e] = v;
    }
    __syncthreads();
    if (shift) {
        if (t < qm) {
            double s = 0, c = 0;
            for (int b = 0; b < qm; ++b) s += blk[t * qm + b];
            for (int a = 0; a < qm; ++a) c += blk[a * qm + t];
            rowm[t] = s / qm;
            colm[t] = c / qm;
        }
        if (t == 63) {
            double s = 0;
            for (int e = 0; e < qm * qm; ++e) s += blk[e];
            tot = s / (double)(qm * qm);
        }
        __syncthreads();
    }
    for (int e = t; e < qm * qm; e += 64) {
        const int a = e / qm, b = e % qm;
        out[(size_t)blockIdx.x * qm * qm + e] = shift ? blk[e] - rowm[a] - colm[b] + tot : blk[e];
    }
}

}  // namespace

int dca_fn_scores(dca_ctx* ctx, const void* src, int src_kind, int dtype, int L, int q, int ld, int apc, double* dOut)
{
    if (q - 1 > 20) { dca_set_error("q too large for the scoring kernel"); return DCA_ERR_ARG; }
    const size_t npairs = (size_t)L * (L - 1) / 2;
    ScopedKernelClock kc(ctx, "scores");
    if (dtype == DCA_F32)
        hipLaunchKernelGGL(fn_kernel<float>, dim3((unsigned)npairs), dim3(64), 0, ctx->stream, static_cast<const float*>(src), src_kind, L, q, ld, dOut);
    else
        hipLaunchKernelGGL(fn_kernel<double>, dim3((unsigned)npairs), dim3(64), 0, ctx->stream, static_cast<const double*>(src), src_kind, L, q, ld, dOut);
    if (apc) {
        double* dAv = nullptr;
        HIP_TRY(dca_dev_malloc(reinterpret_cast<void**>(&dAv), (size_t)(L + 1) * sizeof(double)));
        hipLaunchKernelGGL(apc_site_kernel, dim3(L), dim3(256), 0, ctx->stream, dOut, L, dAv);
        hipLaunchKernelGGL(apc_mean_kernel, dim3(1), dim3(256), 0, ctx->stream, dAv, L, dAv + L);
        hipLaunchKernelGGL(apc_apply_kernel, dim3(L - 1), dim3(256), 0, ctx->stream, dOut, dAv, dAv + L, L);
        hipError_t e = hipStreamSynchronize(ctx->stream);
        dca_dev_free(dAv);
        if (e != hipSuccess) { dca_set_error("apc: %s", hipGetErrorString(e)); return DCA_ERR_HIP; }
    }
    HIP_TRY(hipGetLastError());
    return dca_remember_scores(ctx, dOut, (int)npairs);
}

// DI / DI_APC of coupling blocks; same source conventions as dca_fn_scores.  dRegFi: device, L*q.
int dca_di_scores(dca_ctx* ctx, const void* src, int src_kind, int dtype, const double* dRegFi, int L, int q, int ld,
                  int apc, double* dOut)
{
    if (q > 21) { dca_set_error("q too large for the DI kernel"); return DCA_ERR_ARG; }
    const size_t npairs = (size_t)L * (L - 1) / 2;
    ScopedKernelClock kc(ctx, "scores");
    if (dtype == DCA_F32)
        hipLaunchKernelGGL(di_kernel<float>, dim3((unsigned)npairs), dim3(64), 0, ctx->stream, static_cast<const float*>(src), src_kind, dRegFi, L, q, ld, dOut, static_cast<double*>(nullptr), static_cast<const double*>(nullptr));
    else
        hipLaunchKernelGGL(di_kernel<double>, dim3((unsigned)npairs), dim3(64), 0, ctx->stream, static_cast<const double*>(src), src_kind, dRegFi, L, q, ld, dOut, static_cast<double*>(nullptr), static_cast<const double*>(nullptr));
    if (apc) {
        double* dAv = nullptr;
        HIP_TRY(dca_dev_malloc(reinterpret_cast<void**>(&dAv), (size_t)(L + 1) * sizeof(double)));
        hipLaunchKernelGGL(apc_site_kernel, dim3(L), dim3(256), 0, ctx->stream, dOut, L, dAv);
        hipLaunchKernelGGL(apc_mean_kernel, dim3(1), dim3(256), 0, ctx->stream, dAv, L, dAv + L);
        hipLaunchKernelGGL(apc_apply_kernel, dim3(L - 1), dim3(256), 0, ctx->stream, dOut, dAv, dAv + L, L);
        hipError_t e = hipStreamSynchronize(ctx->stream);
        dca_dev_free(dAv);
        if (e != hipSuccess) { dca_set_error("apc: %s", hipGetErrorString(e)); return DCA_ERR_HIP; }
    }
    HIP_TRY(hipGetLastError());
    return dca_remember_scores(ctx, dOut, (int)npairs);
}

// pairs: host array of 2*npairs ints (i < j); out: host, npairs*(q-1)^2 doubles
int dca_pair_blocks(dca_ctx* ctx, const void* src, int src_kind, int dtype, int L, int q, int ld, const int* pairs, int npairs,
                    int shift, double* out)
{
    if (npairs <= 0) return DCA_OK;
    if (q > 21) { dca_set_error("q too large"); return DCA_ERR_ARG; }
    for (int k = 0; k < npairs; ++k)
        if (pairs[2 * k] < 0 || pairs[2 * k] >= pairs[2 * k + 1] || pairs[2 * k + 1] >= L) { dca_set_error("site pair %d out of order or range", k); return DCA_ERR_ARG; }
    const size_t per = (size_t)(q - 1) * (q - 1);
    int* dPairs = nullptr;
    double* dOut = nullptr;
    HIP_TRY(dca_dev_malloc(reinterpret_cast<void**>(&dPairs), (size_t)npairs * 2 * sizeof(int)));
    if (dca_dev_malloc(reinterpret_cast<void**>(&dOut), (size_t)npairs * per * sizeof(double)) != hipSuccess) { dca_dev_free(dPairs); return DCA_ERR_NOMEM; }
    hipError_t e = hipMemcpyAsync(dPairs, pairs, (size_t)npairs * 2 * sizeof(int), hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess) {
        if (dtype == DCA_F32)
            hipLaunchKernelGGL(pair_blocks_kernel<float>, dim3(npairs), dim3(64), 0, ctx->stream, static_cast<const float*>(src), src_kind, dPairs, L, q, ld, shift, dOut);
        else
            hipLaunchKernelGGL(pair_blocks_kernel<double>, dim3(npairs), dim3(64), 0, ctx->stream, static_cast<const double*>(src), src_kind, dPairs, L, q, ld, shift, dOut);
        e = hipStreamSynchronize(ctx->stream);
    }
    if (e == hipSuccess) e = hipMemcpy(out, dOut, (size_t)npairs * per * sizeof(double), hipMemcpyDeviceToHost);
    dca_dev_free(dPairs); dca_dev_free(dOut);
    if (e != hipSuccess) { dca_set_error("pair blocks: %s", hipGetErrorString(e)); return DCA_ERR_HIP; }
    return DCA_OK;
}

// DI and two-site model fields from caller-provided HOST arrays (the module-level functions of the
// reference: compute_two_site_model_fields / compute_direct_info, meanfield_dca/msa_numerics.py:378-533
// with the n x n couplings matrix, plmdca/msa_numerics.py:156-311 with the gap-stripped 1-D array).
int dca_di_from_arrays_impl(dca_ctx* ctx, const double* couplings, int layout, const double* reg_fi, int L, int q,
                            double* fields_out, double* di_out, const double* fields_in)
{
    if (q > 21 || q < 2 || L < 2) { dca_set_error("dca_di_from_arrays: bad L / q"); return DCA_ERR_ARG; }
    const int qm = q - 1, n = L * qm;
    const size_t npairs = (size_t)L * (L - 1) / 2;
    const size_t nc = layout == 1 ? (size_t)n * n : npairs * qm * qm;
    double *dC = nullptr, *dF = nullptr, *dDi = nullptr, *dFields = nullptr, *dFieldsIn = nullptr;
    hipError_t e = dca_dev_malloc(reinterpret_cast<void**>(&dC), nc * sizeof(double));
    if (e == hipSuccess && fields_in) e = dca_dev_malloc(reinterpret_cast<void**>(&dFieldsIn), npairs * 2 * q * sizeof(double));
    if (e == hipSuccess && fields_in) e = hipMemcpy(dFieldsIn, fields_in, npairs * 2 * q * sizeof(double), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = dca_dev_malloc(reinterpret_cast<void**>(&dF), (size_t)L * q * sizeof(double));
    if (e == hipSuccess) e = dca_dev_malloc(reinterpret_cast<void**>(&dDi), npairs * sizeof(double));
    if (e == hipSuccess && fields_out) e = dca_dev_malloc(reinterpret_cast<void**>(&dFields), npairs * 2 * q * sizeof(double));
    if (e == hipSuccess) e = hipMemcpy(dC, couplings, nc * sizeof(double), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(dF, reg_fi, (size_t)L * q * sizeof(double), hipMemcpyHostToDevice);
    if (e == hipSuccess) {
        hipLaunchKernelGGL(di_kernel<double>, dim3((unsigned)npairs), dim3(64), 0, ctx->stream, dC, layout == 1 ? 1 : 2, dF, L, q, n, dDi, dFields, dFieldsIn);
        e = hipStreamSynchronize(ctx->stream);
    }
    if (e == hipSuccess && di_out) e = hipMemcpy(di_out, dDi, npairs * sizeof(double), hipMemcpyDeviceToHost);
    if (e == hipSuccess && fields_out) e = hipMemcpy(fields_out, dFields, npairs * 2 * q * sizeof(double), hipMemcpyDeviceToHost);
    dca_dev_free(dC); dca_dev_free(dF); dca_dev_free(dDi); dca_dev_free(dFields); dca_dev_free(dFieldsIn);
    if (e != hipSuccess) { dca_set_error("dca_di_from_arrays: %s", hipGetErrorString(e)); return DCA_ERR_HIP; }
    return DCA_OK;
}
