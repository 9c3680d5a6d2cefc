// Dependent-issue latency of v_fma_f64 for ONE workgroup alone on the GPU (the situation of the Cholesky leaf):
// a chain of dependent FMAs, 4 and 16 independent chains, and a barrier + LDS round trip loop.
// hipcc --offload-arch=gfx950 -O3 -o f64_latency f64_latency.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
template <int CH>
__global__ void chain_kernel(double* out, int iters, double a, double b)
{
    double v[CH];
#pragma unroll
    for (int c = 0; c < CH; ++c) v[c] = threadIdx.x * 1e-3 + c;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int c = 0; c < CH; ++c) v[c] = __builtin_fma(v[c], a, b);
    }
    double s = 0;
#pragma unroll
    for (int c = 0; c < CH; ++c) s += v[c];
    out[threadIdx.x] = s;
}
__global__ void barrier_kernel(double* out, int iters)
{
    __shared__ double buf[256];
    double v = threadIdx.x;
    for (int i = 0; i < iters; ++i) {
        buf[threadIdx.x] = v;
        __syncthreads();
        v += buf[(threadIdx.x + 17) & 255];
        __syncthreads();
    }
    out[threadIdx.x] = v;
}
__global__ void heater_kernel(double* out, int iters)
{
    double v[8];
#pragma unroll
    for (int c = 0; c < 8; ++c) v[c] = threadIdx.x * 1e-3 + c;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int c = 0; c < 8; ++c) v[c] = __builtin_fma(v[c], 1.0000001, 1e-9);
    }
    double s = 0;
#pragma unroll
    for (int c = 0; c < 8; ++c) s += v[c];
    if (s == 12345.0) out[0] = s;
}
int main(int argc, char** argv)
{
    // argv[1] = number of heater workgroups kept busy on a second stream while the measurements run (0: GPU otherwise idle)
    const int heaters = argc > 1 ? atoi(argv[1]) : 0;
    hipStream_t hs; hipStreamCreateWithFlags(&hs, hipStreamNonBlocking);
    if (heaters > 0) { double* hd; hipMalloc(&hd, 64); hipLaunchKernelGGL(heater_kernel, dim3(heaters), dim3(256), 0, hs, hd, 60000000); printf("heater: %d workgroups\n", heaters); }
    double* d; CHECK(hipMalloc(&d, 4096));
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    const int iters = 20000;
    hipStream_t ms_; hipStreamCreateWithFlags(&ms_, hipStreamNonBlocking);
    auto run = [&](const char* name, auto launch, double opsPerIter) {
        float best = 1e9;
        for (int r = 0; r < 5; ++r) {
            hipEventRecord(e0, ms_); launch(); hipEventRecord(e1, ms_); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
        }
        printf("%-44s %8.1f ns per iteration, %6.2f ns per op\n", name, best * 1e6 / iters, best * 1e6 / iters / opsPerIter);
    };
    run("1 dependent fma chain, 1 wave", [&] { hipLaunchKernelGGL(chain_kernel<1>, dim3(1), dim3(64), 0, ms_, d, iters, 1.0000001, 1e-9); }, 1);
    run("4 independent chains, 1 wave", [&] { hipLaunchKernelGGL(chain_kernel<4>, dim3(1), dim3(64), 0, ms_, d, iters, 1.0000001, 1e-9); }, 4);
    run("16 independent chains, 1 wave", [&] { hipLaunchKernelGGL(chain_kernel<16>, dim3(1), dim3(64), 0, ms_, d, iters, 1.0000001, 1e-9); }, 16);
    run("16 independent chains, 4 waves (1 per SIMD)", [&] { hipLaunchKernelGGL(chain_kernel<16>, dim3(1), dim3(256), 0, ms_, d, iters, 1.0000001, 1e-9); }, 16);
    run("LDS write + barrier + LDS read + barrier", [&] { hipLaunchKernelGGL(barrier_kernel, dim3(1), dim3(256), 0, ms_, d, iters); }, 1);
    return 0;
}
