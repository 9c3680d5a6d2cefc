// Timing of the scatter gather block with JW sites per wave (gen_scatter_var.py): LDS tile resident, no staging, no
// barrier; state words streamed from global memory by scalar loads (distinct per wave and tile).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <cstdint>
#include "scatter_variant.inc"
#ifndef SC_QSTATES
#define SC_QSTATES 21
#endif
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
#ifndef NWAVES
#define NWAVES 8
#endif
constexpr int ROWS = 128, ROWB = 512, WAVES = NWAVES;
__global__ __launch_bounds__(WAVES * 64)
void bench_kernel(const float* __restrict__ tile, const uint32_t* __restrict__ states, float* __restrict__ out, int iters)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    for (int t = tid; t < ROWS * ROWB / 4; t += WAVES * 64) reinterpret_cast<float*>(smem)[t] = tile[t];
    __syncthreads();
    const uint32_t vbase = (uint32_t)(uintptr_t)smem + lane * 8;
#ifdef SHARED_STATES      // every workgroup reads the same stream: L2 / scalar-cache hits, as the strips of one site group in the product
    const uint32_t* sp = states + (size_t)wave * (size_t)(iters + 1) * SC_STREAM_WORDS_PER_TILE;
#else
    const uint32_t* sp = states + ((size_t)blockIdx.x * WAVES + wave) * (size_t)(iters + 1) * SC_STREAM_WORDS_PER_TILE;
#endif
    float res;
    int it = iters;
#ifdef WITH_DMA      // the product's staging traffic: every wave copies its share of 64 KiB per tile into a second LDS buffer
    const uint32_t voff = lane * 16;
    SCATTER_BLOCK(vbase, sp, it, res, voff, tile);
#else
    SCATTER_BLOCK(vbase, sp, it, res);
#endif
    out[(size_t)blockIdx.x * WAVES * 64 + tid] = res;
}
int main(int argc, char** argv)
{
    const int blocks = argc > 1 ? atoi(argv[1]) : 256, iters = argc > 2 ? atoi(argv[2]) : 100;
    std::vector<float> tile(ROWS * 128, 0.25f);
    std::vector<uint32_t> st((size_t)blocks * WAVES * (iters + 1) * SC_STREAM_WORDS_PER_TILE + 64);
    srand(1);
    for (auto& s : st) s = (0x9000u | (2u * (rand() % SC_QSTATES))) | ((0x9000u | (2u * (rand() % SC_QSTATES))) << 16);
    float *dT, *dO; uint32_t* dS;
    CHECK(hipMalloc(&dT, tile.size() * 4)); CHECK(hipMalloc(&dS, st.size() * 4)); CHECK(hipMalloc(&dO, (size_t)blocks * WAVES * 64 * 4));
    CHECK(hipMemcpy(dT, tile.data(), tile.size() * 4, hipMemcpyHostToDevice));
    CHECK(hipMemcpy(dS, st.data(), st.size() * 4, hipMemcpyHostToDevice));
    CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(bench_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 132 * 1024));
    hipEvent_t a, b; CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
    float best = 1e9;
    for (int rep = 0; rep < 4; ++rep) {
        CHECK(hipEventRecord(a));
        hipLaunchKernelGGL(bench_kernel, dim3(blocks), dim3(WAVES * 64), 132 * 1024, 0, dT, dS, dO, iters);
        CHECK(hipEventRecord(b));
        CHECK(hipEventSynchronize(b));
        float ms; CHECK(hipEventElapsedTime(&ms, a, b));
        if (ms < best) best = ms;
    }
    CHECK(hipGetLastError());
    const double units = (double)blocks * WAVES * SC_JW * iters * ROWS;
    printf("%s waves %d jw %d: %.3f ms  %.2f clk per (row,site) per CU @2.4GHz\n", VARIANT, WAVES, SC_JW, best, best * 1e-3 * 2.4e9 / (units / 256));
    return 0;
}
