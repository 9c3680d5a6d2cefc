// Micro-benchmark of the product gather block (two sites per wave, 128 VGPRs, one workgroup per CU)
// and of ablated variants (gen_variants.py) to see which unit bounds it.  Timing only.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <cstdint>
typedef float v32f __attribute__((ext_vector_type(32)));
typedef float v8f __attribute__((ext_vector_type(8)));
typedef float v2f __attribute__((ext_vector_type(2)));
#include "gather_variant.inc"
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
#ifndef NWAVES
#define NWAVES 16
#endif
constexpr int ROWS = 128, ROWB = 512, WAVES = NWAVES;

__global__ __launch_bounds__(1024)
void bench_kernel(const float* __restrict__ tile, const uint16_t* __restrict__ states, float* __restrict__ out, int iters)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    for (int t = tid; t < ROWS * ROWB / 4; t += WAVES * 64) reinterpret_cast<float*>(smem)[t] = tile[t];
    __syncthreads();
    v32f a0, a1; v8f b0, b1; v2f c0, c1;
    for (int i = 0; i < 32; ++i) a0[i] = a1[i] = 0.f;
    for (int i = 0; i < 8; ++i) b0[i] = b1[i] = 0.f;
    c0[0] = c0[1] = c1[0] = c1[1] = 0.f;
    const uint32_t vbase = (uint32_t)(uintptr_t)smem + lane * 8;
    const uint32_t* sp = reinterpret_cast<const uint32_t*>(states) + (size_t)(blockIdx.x * WAVES + wave) * 2 * 64;
    const uint32_t st0 = sp[lane], st1 = sp[64 + lane];
#ifdef USE_SMEM
    const uint32_t* sp0 = sp;
    const uint32_t* sp1 = sp + 64;
    for (int it = 0; it < iters; ++it) GATHER_BLOCK_SMEM(vbase, sp0, sp1, a0, b0, c0, a1, b1, c1);
    (void)st0; (void)st1;
#else
    for (int it = 0; it < iters; ++it) GATHER_BLOCK(vbase, st0, st1, a0, b0, c0, a1, b1, c1);
#endif
    float s = 0;
    for (int i = 0; i < 32; ++i) s += a0[i] + a1[i];
    for (int i = 0; i < 8; ++i) s += b0[i] + b1[i];
    s += c0[0] + c0[1] + c1[0] + c1[1];
    out[(size_t)blockIdx.x * WAVES * 64 + tid] = s;
}

int main(int argc, char** argv)
{
    const int blocks = argc > 1 ? atoi(argv[1]) : 256, iters = argc > 2 ? atoi(argv[2]) : 400;
    std::vector<float> tile(ROWS * 128, 0.25f);
    std::vector<uint16_t> st((size_t)blocks * WAVES * 2 * 128);
    srand(1);
    for (auto& s : st) s = (uint16_t)(0x9000 | (2 * (rand() % 21)));
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
    const double rowsites = (double)blocks * WAVES * 2 * iters * ROWS;
    printf("%s waves %d: blocks %d: %.3f ms  %.2f clk per (row,site) per CU @2.4GHz\n", VARIANT, WAVES, blocks, best, best * 1e-3 * 2.4e9 / (rowsites / 256));
    return 0;
}
