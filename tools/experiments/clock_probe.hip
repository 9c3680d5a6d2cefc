// Shader clock seen by a kernel (s_memtime ticks per 100 MHz wall tick): one workgroup alone, then right after a
// chip-filling kernel, then one workgroup again.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
__global__ void spin(double* out, long long wall_ticks, int slot)
{
    const long long w0 = wall_clock64(), c0 = clock64();
    double a = threadIdx.x;
    while (wall_clock64() - w0 < wall_ticks) { for (int i = 0; i < 64; ++i) a = a * 1.0000001 + 0.5; }
    const long long w1 = wall_clock64(), c1 = clock64();
    if (threadIdx.x == 0 && blockIdx.x == 0) { out[2 * slot] = (double)(c1 - c0) / (double)(w1 - w0) * 100.0; out[2 * slot + 1] = a; }
}
int main()
{
    double* d; CHECK(hipMalloc(&d, 64 * sizeof(double)));
    int slot = 0;
    auto run = [&](int blocks, int ms) { hipLaunchKernelGGL(spin, dim3(blocks), dim3(512), 0, 0, d, (long long)ms * 100000, slot++); };
    run(1, 2); run(1, 2); run(1, 5);             // one workgroup alone
    run(2048, 5); run(2048, 5);                  // the whole chip busy (VALU f64)
    run(1, 1); run(1, 1); run(1, 2); run(1, 5);  // alone again
    run(2048, 2); run(2048, 2); run(2048, 2);    // busy again: how fast does it come back
    CHECK(hipDeviceSynchronize());
    double h[64]; CHECK(hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost));
    const char* what[] = {"1 WG 2ms", "1 WG 2ms", "1 WG 5ms", "2048 WG 5ms", "2048 WG 5ms", "1 WG 1ms", "1 WG 1ms", "1 WG 2ms", "1 WG 5ms", "2048 WG 2ms", "2048 WG 2ms", "2048 WG 2ms"};
    for (int i = 0; i < slot; ++i) printf("%-12s shader clock %.0f MHz\n", what[i], h[2 * i]);
    return 0;
}
