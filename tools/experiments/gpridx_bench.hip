// Micro-benchmark + semantic check of the VGPR-index-mode gather (see gen_gather_asm.py).
// hipcc --offload-arch=gfx950 -O3 -o gpridx_bench gpridx_bench.hip && ./gpridx_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <cstdint>

typedef float v32f __attribute__((ext_vector_type(32)));
typedef float v8f __attribute__((ext_vector_type(8)));
typedef float v2f __attribute__((ext_vector_type(2)));

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

constexpr int Q = 21, ROWS = 128, ROWB = 512, WAVES = 16;

// states2: [numSites][iters][128] bytes (2*state); tile: 128 x 128 floats; out: [numSites][Q][128]
__global__ __launch_bounds__(WAVES * 64, WAVES / 2)
void bench_kernel(const float* __restrict__ tile, const uint8_t* __restrict__ states2, float* __restrict__ out, int iters)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int t = tid; t < ROWS * ROWB / 4; t += WAVES * 64) reinterpret_cast<float*>(smem)[t] = tile[t];
    __syncthreads();
    const int site = blockIdx.x * WAVES + wave;
    v32f accA; v8f accB; v2f accC;
    for (int i = 0; i < 32; ++i) accA[i] = 0.f;
    for (int i = 0; i < 8; ++i) accB[i] = 0.f;
    accC[0] = accC[1] = 0.f;
    const uint32_t vbase = (uint32_t)(uintptr_t)smem + lane * 8;   // LDS byte address (group segment offset)
    const uint8_t* sp = states2 + (size_t)site * iters * ROWS;
    uint32_t st = (lane < 32) ? reinterpret_cast<const uint32_t*>(sp)[lane] : 0u;
    for (int it = 0; it < iters; ++it) {
        const int nx = (it + 1 < iters) ? it + 1 : it;
        const uint32_t stn = (lane < 32) ? reinterpret_cast<const uint32_t*>(sp + (size_t)nx * ROWS)[lane] : 0u;
        asm volatile(
#include "gather_q21_jw1_f32.inc"
            : "+{v[22:53]}"(accA), "+{v[54:61]}"(accB), "+{v[62:63]}"(accC)
            : [vbase] "v"(vbase), [st0] "v"(st)
            : "memory", "v6", "v7", "v8", "v9", "v10", "v11", "v12", "v13", "v14", "v15", "v16", "v17", "v18", "v19", "v20", "v21",
              "s40", "s41", "s42", "s43", "s44", "s45", "s46", "s47", "s48", "s49", "s50", "s51", "s52", "s53", "s54", "s55", "s56", "s57");
        st = stn;
    }
    float* o = out + (size_t)site * Q * 128 + lane * 2;
    for (int b = 0; b < 16; ++b) { o[b * 128] = accA[2 * b]; o[b * 128 + 1] = accA[2 * b + 1]; }
    for (int b = 0; b < 4; ++b) { o[(16 + b) * 128] = accB[2 * b]; o[(16 + b) * 128 + 1] = accB[2 * b + 1]; }
    o[20 * 128] = accC[0]; o[20 * 128 + 1] = accC[1];
}

int main(int argc, char** argv)
{
    const int blocks = argc > 1 ? atoi(argv[1]) : 512, iters = argc > 2 ? atoi(argv[2]) : 400;
    const int sites = blocks * WAVES;
    std::vector<float> tile(ROWS * 128);
    srand(1);
    for (auto& v : tile) v = (float)(rand() % 2001 - 1000) / 1024.f;
    std::vector<uint8_t> st((size_t)sites * iters * ROWS);
    for (auto& s : st) s = (uint8_t)(2 * (rand() % Q));
    float *dT, *dO; uint8_t* dS;
    CHECK(hipMalloc(&dT, tile.size() * 4)); CHECK(hipMalloc(&dS, st.size())); CHECK(hipMalloc(&dO, (size_t)sites * Q * 128 * 4));
    CHECK(hipMemcpy(dT, tile.data(), tile.size() * 4, hipMemcpyHostToDevice));
    CHECK(hipMemcpy(dS, st.data(), st.size(), hipMemcpyHostToDevice));
    CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(bench_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, ROWS * ROWB));
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(a);
        hipLaunchKernelGGL(bench_kernel, dim3(blocks), dim3(WAVES * 64), ROWS * ROWB, 0, dT, dS, dO, iters);
        hipEventRecord(b);
        CHECK(hipEventSynchronize(b));
        float ms; hipEventElapsedTime(&ms, a, b);
        const double units = (double)sites * iters;                 // (tile, site) units
        printf("blocks %d iters %d: %.3f ms, %.1f ns per (tile,site) per CU-slot, LDS %.1f TB/s, %.2f clk/row/CU @2.4GHz\n", blocks, iters, ms,
               ms * 1e6 / units * 256, units * ROWS * ROWB / (ms * 1e-3) / 1e12, ms * 1e-3 * 2.4e9 / (units * ROWS / 256));
    }
    std::vector<float> out((size_t)sites * Q * 128);
    CHECK(hipMemcpy(out.data(), dO, out.size() * 4, hipMemcpyDeviceToHost));
    // check a few sites exactly (same summation order)
    int bad = 0;
    for (int s : {0, 1, 17, sites - 1}) {
        std::vector<float> ref(Q * 128, 0.f);
        for (int it = 0; it < iters; ++it)
            for (int r = 0; r < ROWS; ++r) {
                const int b = st[((size_t)s * iters + it) * ROWS + r] / 2;
                for (int c = 0; c < 128; ++c) ref[b * 128 + c] += tile[r * 128 + c];
            }
        for (int k = 0; k < Q * 128; ++k) if (ref[k] != out[(size_t)s * Q * 128 + k]) { if (bad < 5) printf("mismatch site %d k %d: %g vs %g\n", s, k, out[(size_t)s * Q * 128 + k], ref[k]); ++bad; }
    }
    printf(bad ? "FAILED (%d mismatches)\n" : "OK: bit-exact vs sequential CPU sums (%d)\n", bad);
    return bad != 0;
}
