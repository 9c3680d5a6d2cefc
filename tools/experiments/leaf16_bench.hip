// Round 6: the 16-column-step leaf alone (back-to-back launches) and the in-wave 16 x 16 factor + invert alone.
// hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -Wno-unused-result -I../../include [-DDCA_LEAF16_ABLATE=k] -o leaf16_bench leaf16_bench.hip
#include <unistd.h>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "../../pydca_amd/csrc/cholinv.hip"
void dca_set_error(const char* fmt, ...) { va_list ap; va_start(ap, fmt); vfprintf(stderr, fmt, ap); va_end(ap); }
void dca_flush_clocks(dca_ctx*) {}
hipError_t dca_dev_malloc(void** p, size_t b, bool) { return hipMalloc(p, b); }
hipError_t dca_dev_free(void* p) { return hipFree(p); }
__global__ __launch_bounds__(64) void factor16_loop_kernel(const double* S, double* out, int iters, int* info)
{
    __shared__ double sb[256], db[256];
    for (int e = threadIdx.x; e < 256; e += 64) sb[e] = S[e];
    __syncthreads();
    for (int it = 0; it < iters; ++it) {
        factor16(sb, db, threadIdx.x, info, 0);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (threadIdx.x < 16) sb[threadIdx.x * 17] += db[threadIdx.x * 17] * 1e-30;     // keeps the iterations dependent
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    for (int e = threadIdx.x; e < 256; e += 64) out[e] = db[e];
}
int main()
{
    const int n = 256, reps = 200;
    std::vector<double> h((size_t)n * n);
    for (int i = 0; i < n; ++i) for (int j = 0; j < n; ++j) h[(size_t)i * n + j] = (i == j ? 2.0 : 0.0) + 0.5 / (1.0 + abs(i - j));
    double* d; hipMalloc(&d, sizeof(double) * n * n * reps);
    for (int r = 0; r < reps; ++r) hipMemcpy(d + (size_t)r * n * n, h.data(), sizeof(double) * n * n, hipMemcpyHostToDevice);
    int* info; hipMalloc(&info, 4); hipMemset(info, 0, 4);
    hipStream_t st; hipStreamCreate(&st);
    gemm_kernels_prepare(0);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    auto timeit = [&](const char* name, auto launch) {
        hipEventRecord(e0, st);
        for (int r = 0; r < reps; ++r) launch(d + (size_t)r * n * n);
        hipEventRecord(e1, st); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("%-28s %7.2f us per launch (ablate %d)\n", name, ms * 1e3 / reps, DCA_LEAF16_ABLATE);
    };
    timeit("leaf16<64>", [&](double* m) { hipLaunchKernelGGL(cholinv_leaf16_small_kernel<64>, dim3(1), dim3(512), leaf16_lds_bytes<64>(), st, m, n, 0, info); });
    timeit("leaf16<128>", [&](double* m) { hipLaunchKernelGGL(cholinv_leaf16_small_kernel<128>, dim3(1), dim3(512), leaf16_lds_bytes<128>(), st, m, n, 0, info); });
    timeit("leaf16<192>", [&](double* m) { hipLaunchKernelGGL(cholinv_leaf16_kernel<192>, dim3(1), dim3(512), leaf16_lds_bytes<192>(), st, m, n, 0, info); });
    timeit("leaf16<256>", [&](double* m) { hipLaunchKernelGGL(cholinv_leaf16_kernel<256>, dim3(1), dim3(512), leaf16_lds_bytes<256>(), st, m, n, 0, info); });
    for (int r = 0; r < reps; ++r) hipMemcpy(d + (size_t)r * n * n, h.data(), sizeof(double) * n * n, hipMemcpyHostToDevice);
    timeit("old mfma leaf<128>", [&](double* m) { hipLaunchKernelGGL(cholinv_leaf_mfma_kernel<128>, dim3(1), dim3(DCA_LEAF_WAVES128 * 64), 0, st, m, n, 0, info); });
    if (DCA_LEAF16_ABLATE == 0) {
        double* out; hipMalloc(&out, 2048);
        const int iters = 2000;
        hipEventRecord(e0, st);
        hipLaunchKernelGGL(factor16_loop_kernel, dim3(1), dim3(64), 0, st, d, out, iters, info);
        hipEventRecord(e1, st); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("factor16 alone: %.2f us per 16 x 16 tile\n", ms * 1e3 / iters);
    }
    int hi; hipMemcpy(&hi, info, 4, hipMemcpyDeviceToHost); printf("info %d\n", hi);
    return 0;
}
