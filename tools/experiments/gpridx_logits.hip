// Timing of the logits inner block (gen_logits_variants.py): one LDS tile of 6 sites x 21 rows, 32
// sequences per wave, state words from global memory by scalar loads (a different 64-byte line for
// every wave, site and iteration, so the scalar cache does not help).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <cstdint>
#include "logits_variant.inc"
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
#ifndef NBSEQ
#define NBSEQ 32
#endif
#ifndef NWAVES
#define NWAVES 16
#endif
constexpr int ROWS = 128, ROWB = 512, WAVES = NWAVES, JT = 6;
__global__ __launch_bounds__(WAVES * 64)
void bench_kernel(const float* __restrict__ tile, const uint16_t* __restrict__ states, float* __restrict__ out, int iters, int strideBytes)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    for (int t = tid; t < ROWS * ROWB / 4; t += WAVES * 64) reinterpret_cast<float*>(smem)[t] = tile[t];
    __syncthreads();
    const uint32_t vbase = (uint32_t)(uintptr_t)smem + lane * 8;
    const uint16_t* sp = states + ((size_t)blockIdx.x * WAVES + wave) * NBSEQ;
#ifdef PREFETCH
    // as in the product kernel: one vector load per wave touches every 64-byte line of the NEXT tile's state words
    constexpr int LPS = (2 * NBSEQ + 63) / 64 + 1;
    const int pfSite = min(lane / LPS, JT - 1);
    const size_t pfOff = (size_t)pfSite * strideBytes + min((lane % LPS) * 64, NBSEQ * 2 - 4);
    uint32_t sink = 0;
#endif
    for (int it = 0; it < iters; ++it) {
#ifdef PREFETCH
        if (it + 1 < iters)
            __builtin_amdgcn_global_load_lds(
                (const __attribute__((address_space(1))) void*)(reinterpret_cast<const unsigned char*>(sp) + (size_t)JT * strideBytes + pfOff),
                (__attribute__((address_space(3))) void*)(smem + ROWS * ROWB + wave * 256), 4, 0, 0);
#endif
        LOGITS_BLOCK(vbase, sp, strideBytes);
#ifndef STATES_CACHED
        sp += (size_t)JT * strideBytes / 2;
#endif
    }
#ifdef PREFETCH
    out[(size_t)blockIdx.x * WAVES * 64 + tid] = (float)(iters + (sink == 0x12345u));
#else
    out[(size_t)blockIdx.x * WAVES * 64 + tid] = (float)iters;
#endif
}
int main(int argc, char** argv)
{
    const int blocks = argc > 1 ? atoi(argv[1]) : 256, iters = argc > 2 ? atoi(argv[2]) : 84;
    const int seqs = blocks * WAVES * NBSEQ;
    const int strideBytes = seqs * 2;
    std::vector<float> tile(ROWS * 128, 0.25f);
    std::vector<uint16_t> st((size_t)seqs * JT * iters + 4096);
    srand(1);
    for (auto& s : st) s = (uint16_t)(0x2000 | (2 * (rand() % 21)));
    float *dT, *dO; uint16_t* dS;
    CHECK(hipMalloc(&dT, tile.size() * 4)); CHECK(hipMalloc(&dS, st.size() * 2)); CHECK(hipMalloc(&dO, (size_t)blocks * WAVES * 64 * 4));
    CHECK(hipMemcpy(dT, tile.data(), tile.size() * 4, hipMemcpyHostToDevice));
    CHECK(hipMemcpy(dS, st.data(), st.size() * 2, hipMemcpyHostToDevice));
    CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(bench_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, ROWS * ROWB + WAVES * 256));
    hipEvent_t a, b; CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
    float best = 1e9;
    for (int rep = 0; rep < 4; ++rep) {
        CHECK(hipEventRecord(a));
        hipLaunchKernelGGL(bench_kernel, dim3(blocks), dim3(WAVES * 64), ROWS * ROWB + WAVES * 256, 0, dT, dS, dO, iters, strideBytes);
        CHECK(hipEventRecord(b));
        CHECK(hipEventSynchronize(b));
        float ms; CHECK(hipEventElapsedTime(&ms, a, b));
        if (ms < best) best = ms;
    }
    const double units = (double)blocks * WAVES * NBSEQ * JT * iters;
    printf("%s waves %d: blocks %d: %.3f ms  %.2f clk per (sequence,site) per CU @2.4GHz\n", VARIANT, WAVES, blocks, best, best * 1e-3 * 2.4e9 / (units / 256));
    fflush(stdout);
    return 0;
}
