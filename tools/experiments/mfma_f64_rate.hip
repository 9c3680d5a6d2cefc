// Raw issue rate of v_mfma_f64_16x16x4_f64: NACC independent accumulators per wave, operands in registers, 1..4 waves per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef double double4_t __attribute__((ext_vector_type(4)));
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
template <int NACC>
__global__ __launch_bounds__(1024) void k(double* out, int iters, double a0, double b0)
{
    double4_t acc[NACC];
    for (int i = 0; i < NACC; ++i) acc[i] = (double4_t){0, 0, 0, 0};
    double a = a0 + threadIdx.x, b = b0 - threadIdx.x;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
    }
    double s = 0;
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int NACC> void run(int waves)
{
    double* d; CHECK(hipMalloc(&d, 256 * 1024 * 8));
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    const int iters = 4000;
    float best = 1e9;
    for (int rep = 0; rep < 3; ++rep) {
        CHECK(hipEventRecord(e0));
        hipLaunchKernelGGL(k<NACC>, dim3(256), dim3(waves * 64), 0, 0, d, iters, 1.0, 2.0);
        CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
        float ms; CHECK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
    }
    const double flops = 256.0 * waves * iters * NACC * 2048.0;
    printf("NACC %d waves/CU %2d: %.3f ms  %.1f TFLOP/s  (%.1f clk per MFMA per SIMD @2.4GHz)\n", NACC, waves, best, flops / best / 1e9,
           best * 1e-3 * 2.4e9 / (iters * NACC * (waves / 4.0)));
    CHECK(hipFree(d));
}
int main() { for (int w : {4, 8, 16}) { run<1>(w); run<2>(w); run<4>(w); run<8>(w); } return 0; }
