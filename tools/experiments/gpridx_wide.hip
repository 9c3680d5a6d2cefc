// Timing experiment: 16-byte-lane gather block (gen_wide.py), 8 waves per workgroup, 256 VGPRs.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <cstdint>
#include "gather_wide.inc"
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
constexpr int ROWS = 64, ROWB = 1024, WAVES = 8;
__global__ __launch_bounds__(WAVES * 64)
void bench_kernel(const float* __restrict__ tile, const uint16_t* __restrict__ states, float* __restrict__ out, int iters)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int t = tid; t < ROWS * ROWB / 4; t += WAVES * 64) reinterpret_cast<float*>(smem)[t] = tile[t];
    __syncthreads();
    const uint32_t vbase = (uint32_t)(uintptr_t)smem + lane * 16;
    const uint32_t* sp = reinterpret_cast<const uint32_t*>(states) + (size_t)(blockIdx.x * WAVES + wave) * 2 * 64;
    const uint32_t st0 = sp[lane], st1 = sp[64 + lane];
    for (int it = 0; it < iters; ++it) GATHER_WIDE(vbase, st0, st1);
    out[(size_t)blockIdx.x * WAVES * 64 + tid] = (float)iters;
}
int main(int argc, char** argv)
{
    const int blocks = argc > 1 ? atoi(argv[1]) : 256, iters = argc > 2 ? atoi(argv[2]) : 400;
    std::vector<float> tile(ROWS * 256, 0.25f);
    std::vector<uint16_t> st((size_t)blocks * WAVES * 2 * 128);
    srand(1);
    for (auto& s : st) s = (uint16_t)(0x9000 | (4 * (rand() % 21)));
    float *dT, *dO; uint16_t* dS;
    CHECK(hipMalloc(&dT, tile.size() * 4)); CHECK(hipMalloc(&dS, st.size() * 2)); CHECK(hipMalloc(&dO, (size_t)blocks * WAVES * 64 * 4));
    CHECK(hipMemcpy(dT, tile.data(), tile.size() * 4, hipMemcpyHostToDevice));
    CHECK(hipMemcpy(dS, st.data(), st.size() * 2, hipMemcpyHostToDevice));
    CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(bench_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, ROWS * ROWB));
    hipEvent_t a, b; CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
    float best = 1e9;
    for (int rep = 0; rep < 4; ++rep) {
        CHECK(hipEventRecord(a));
        hipLaunchKernelGGL(bench_kernel, dim3(blocks), dim3(WAVES * 64), ROWS * ROWB, 0, dT, dS, dO, iters);
        CHECK(hipEventRecord(b));
        CHECK(hipEventSynchronize(b));
        float ms; CHECK(hipEventElapsedTime(&ms, a, b));
        if (ms < best) best = ms;
    }
    const double units512 = (double)blocks * WAVES * 2 * iters * ROWS * 2;
    printf("wide (16 B lanes, 8 waves, 256 VGPRs): blocks %d: %.3f ms  %.2f clk per 512-byte unit per CU @2.4GHz\n", blocks, best, best * 1e-3 * 2.4e9 / (units512 / 256));
    fflush(stdout);
    return 0;
}
