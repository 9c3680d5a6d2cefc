// Round 6: what one host round trip of the optimiser costs -- (a) tiny kernel, hipMemcpyAsync of 80 bytes to pinned memory,
// hipStreamSynchronize, next kernel; (b) the scalars written by a one-wave kernel into mapped host memory with a sequence word the host
// polls.  Reported: time per round trip over 2000 trips (the kernels themselves are ~2 us).
// hipcc --offload-arch=gfx950 -O3 -o readback_bench readback_bench.hip
#include <hip/hip_runtime.h>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdint>
__global__ void work_kernel(double* s, int it) { if (threadIdx.x < 10) s[threadIdx.x] = it + threadIdx.x; }
__global__ void publish_kernel(const double* s, int n, double* pub, unsigned long long* flag, unsigned long long seq)
{
    if ((int)threadIdx.x < n) __hip_atomic_store(&pub[threadIdx.x], s[threadIdx.x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __threadfence_system();
    if (threadIdx.x == 0) __hip_atomic_store(flag, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}
int main()
{
    hipStream_t st; hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
    double* d; hipMalloc(&d, 64 * 8);
    double* h; hipHostMalloc(&h, 64 * 8, hipHostMallocDefault);
    double* pub; hipHostMalloc(&pub, 64 * 8, hipHostMallocMapped | hipHostMallocCoherent);
    unsigned long long* flag = reinterpret_cast<unsigned long long*>(pub + 32);
    *flag = 0;
    const int N = 2000;
    for (int mode = 0; mode < 3; ++mode) {
        for (int rep = 0; rep < 2; ++rep) {
            auto t0 = std::chrono::steady_clock::now();
            double chk = 0;
            for (int it = 1; it <= N; ++it) {
                hipLaunchKernelGGL(work_kernel, dim3(1), dim3(64), 0, st, d, it);
                if (mode == 0) {
                    hipMemcpyAsync(h, d, 80, hipMemcpyDeviceToHost, st);
                    hipStreamSynchronize(st);
                    chk += h[3];
                } else if (mode == 1) {
                    const unsigned long long seq = (unsigned long long)rep * N + it + (mode << 20);
                    hipLaunchKernelGGL(publish_kernel, dim3(1), dim3(64), 0, st, d, 10, pub, flag, seq);
                    while (__atomic_load_n(flag, __ATOMIC_ACQUIRE) != seq) { __builtin_ia32_pause(); }
                    chk += pub[3];
                } else {
                    hipStreamSynchronize(st);        // no copy: the wait alone
                }
            }
            hipStreamSynchronize(st);
            const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / N;
            printf("%-44s %7.2f us per trip (check %.0f)\n", mode == 0 ? "memcpyAsync + streamSynchronize" : mode == 1 ? "publish kernel + poll of mapped host memory" : "kernel + streamSynchronize only", us, chk);
        }
    }
    return 0;
}
