// Pure add-chain micro-benchmark (gen_chain.py): per-wave cost of  M0 write -> index-mode v_pk_add_f32  and its
// scaling with 1..4 waves per SIMD; one workgroup per CU (the LDS request keeps a second one out).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <cstdint>
#include "chain_variants.inc"
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

#define DEFINE_KERNEL(name, MACRO) \
__global__ __launch_bounds__(1024) void k_##name(const uint32_t* __restrict__ states, float* __restrict__ out, int iters) \
{ \
    extern __shared__ unsigned char smem[]; \
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6); \
    const uint32_t* sp = states + ((size_t)blockIdx.x * 16 + wave) * 32; \
    float res; \
    MACRO(sp, iters, res); \
    out[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = res + (float)smem[0]; \
}
CHAIN_FOR_EACH(DEFINE_KERNEL)

typedef void (*kern_t)(const uint32_t*, float*, int);
struct Entry { const char* name; kern_t fn; };
#define ENTRY(name, MACRO) { #name, k_##name },
static Entry entries[] = { CHAIN_FOR_EACH(ENTRY) };

int main(int argc, char** argv)
{
    const int iters = argc > 1 ? atoi(argv[1]) : 2000;
    const int blocks = 256;
    std::vector<uint32_t> st((size_t)blocks * 16 * 32);
    srand(1);
    for (auto& s : st) s = (0x2000u | (2u * (rand() % 21))) | ((0x2000u | (2u * (rand() % 21))) << 16);
    uint32_t* dS; float* dO;
    CHECK(hipMalloc(&dS, st.size() * 4)); CHECK(hipMalloc(&dO, (size_t)blocks * 1024 * 4));
    CHECK(hipMemcpy(dS, st.data(), st.size() * 4, hipMemcpyHostToDevice));
    hipEvent_t a, b; CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
    const size_t lds = 96 * 1024;
    for (auto& ent : entries) {
        CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(ent.fn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        for (int waves : {4, 8, 12, 16}) {
            float best = 1e9;
            for (int rep = 0; rep < 3; ++rep) {
                CHECK(hipEventRecord(a));
                hipLaunchKernelGGL(ent.fn, dim3(blocks), dim3(waves * 64), lds, 0, dS, dO, iters);
                CHECK(hipEventRecord(b));
                CHECK(hipEventSynchronize(b));
                float ms; CHECK(hipEventElapsedTime(&ms, a, b));
                if (ms < best) best = ms;
            }
            CHECK(hipGetLastError());
            const double clk = best * 1e-3 * 2.4e9;
            const double units_wave = (double)CHAIN_NU * iters;
            printf("%-12s %2d waves/CU: %.3f ms  %.2f clk/unit/wave  %.2f clk/unit/SIMD  %.3f clk/unit/CU (@2.4GHz)\n", ent.name, waves, best,
                   clk / units_wave, clk / (units_wave * waves / 4), clk / (units_wave * waves));
        }
    }
    fflush(stdout);
    return 0;
}
