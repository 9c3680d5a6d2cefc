// Basic issue rates on gfx950 with 4 waves per SIMD (16-wave workgroup, one per CU):
// v_pk_add_f32, v_add_f32, v_add_f64, SALU s_lshr, in straight-line blocks of 256 independent ops.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
#ifndef THREADS
#define THREADS 1024
#endif
#define R4(x) x x x x
#define R16(x) R4(x) R4(x) R4(x) R4(x)
#define R64(x) R16(x) R16(x) R16(x) R16(x)
template <int V>
__global__ __launch_bounds__(1024) void k(float* out, long long* cyc, int iters)
{
    float a0 = threadIdx.x, a1 = 1, a2 = 2, a3 = 3, a4 = 4, a5 = 5, a6 = 6, a7 = 7;
    double d0 = 1, d1 = 2, d2 = 3, d3 = 4;
    typedef float v2f __attribute__((ext_vector_type(2)));
    v2f p0 = {a0, 1}, p1 = {2, 3}, p2 = {4, 5}, p3 = {6, 7}, inc = {1, 1};
    int s0 = 0;
    long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
        if (V == 0) asm volatile(R64("v_pk_add_f32 %0, %0, %4\n v_pk_add_f32 %1, %1, %4\n v_pk_add_f32 %2, %2, %4\n v_pk_add_f32 %3, %3, %4\n") : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(inc));
        if (V == 1) asm volatile(R64("v_add_f32 %0, %0, %4\n v_add_f32 %1, %1, %4\n v_add_f32 %2, %2, %4\n v_add_f32 %3, %3, %4\n") : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(a7));
        if (V == 2) asm volatile(R64("v_add_f64 %0, %0, %4\n v_add_f64 %1, %1, %4\n v_add_f64 %2, %2, %4\n v_add_f64 %3, %3, %4\n") : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3) : "v"(d3));
        if (V == 3) asm volatile(R64("v_add_f32_e64 %0, %0, %4\n v_add_f32_e64 %1, %1, %4\n v_add_f32_e64 %2, %2, %4\n v_add_f32_e64 %3, %3, %4\n") : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(a7));
        if (V == 4) asm volatile(R64("s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n"));
        if (V == 5) asm volatile(R64("v_add_f32 %0, %0, %4\n s_nop 0\n v_add_f32 %2, %2, %4\n s_nop 0\n") : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(a7));
    }
    long long t1 = __builtin_readcyclecounter();
    out[blockIdx.x * 1024 + threadIdx.x] = a0 + a1 + a2 + a3 + p0[0] + p1[1] + p2[0] + p3[1] + (float)(d0 + d1 + d2 + d3) + s0;
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}
template <int V> void run(const char* name, float* dO, long long* dC)
{
    const int iters = 200, blocks = 256;
    hipEvent_t a, b; CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
    float best = 1e9;
    for (int r = 0; r < 3; ++r) {
        CHECK(hipEventRecord(a));
        hipLaunchKernelGGL(k<V>, dim3(blocks), dim3(THREADS), 0, 0, dO, dC, iters);
        CHECK(hipEventRecord(b)); CHECK(hipEventSynchronize(b));
        float ms; CHECK(hipEventElapsedTime(&ms, a, b)); if (ms < best) best = ms;
    }
    long long c; CHECK(hipMemcpy(&c, dC, 8, hipMemcpyDeviceToHost));
    const double ops = (double)iters * 256 * (THREADS / 64);   // wave-instructions per CU (V=4: pairs)
    printf("%-22s %.3f ms  %.2f ns per wave-instr per CU -> %.2f clk @2.4GHz; s_memtime ticks %lld (%.1f MHz)\n", name, best, best * 1e6 / ops, best * 1e-3 * 2.4e9 / ops, c, c / (best * 1e3));
}
int main(int argc, char** argv)
{
    const int which = argc > 1 ? atoi(argv[1]) : 0;
    float* dO; long long* dC;
    CHECK(hipMalloc(&dO, 256 * 1024 * 4)); CHECK(hipMalloc(&dC, 8));
    if (which == 0) run<0>("v_pk_add_f32", dO, dC);
    if (which == 1) run<1>("v_add_f32", dO, dC);
    if (which == 2) run<2>("v_add_f64", dO, dC);
    if (which == 3) run<3>("v_add_f32_e64 (8 B)", dO, dC);
    if (which == 4) run<4>("s_nop 0 (4 B)", dO, dC);
    if (which == 5) run<5>("v_add_f32 + s_nop", dO, dC);
    fflush(stdout);
    return 0;
}
