// Stand-alone timing of the 64 x 64 leaf of csrc/cholinv.hip (the file is included: product kernel), alone on the
// GPU and back to back, with parts switched off (-DDCA_LEAF_ABLATE=1 no diagonal factorisation, 2 no panel
// products, 3 no updates) to see what its time is made of.
// hipcc --offload-arch=gfx950 -O3 -I../../include -I../../pydca_amd/csrc [-DDCA_LEAF_ABLATE=k] -o leaf_bench leaf_bench.hip
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "../../pydca_amd/csrc/cholinv.hip"
void dca_set_error(const char* fmt, ...) { va_list ap; va_start(ap, fmt); vfprintf(stderr, fmt, ap); va_end(ap); }
void dca_flush_clocks(dca_ctx*) {}
hipError_t dca_dev_malloc(void** p, size_t b, bool) { return hipMalloc(p, b); }
hipError_t dca_dev_free(void* p) { return hipFree(p); }
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
int main()
{
#ifndef LEAF_NB
#define LEAF_NB 64
#endif
    const int n = LEAF_NB, ld = 10048, reps = 200;
    std::vector<double> h((size_t)n * ld, 0.0);
    for (int i = 0; i < n; ++i)
        for (int j = 0; j <= i; ++j) h[(size_t)i * ld + j] = (i == j) ? 4.0 + 0.01 * i : 0.3 / (1.0 + i - j);
    double* dA; int* dInfo;
    CHECK(hipMalloc(&dA, h.size() * 8 * 2)); CHECK(hipMalloc(&dInfo, 4)); CHECK(hipMemset(dInfo, 0, 4));
    hipStream_t st; CHECK(hipStreamCreate(&st));
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    float best = 1e9;
    for (int outer = 0; outer < 5; ++outer) {
        CHECK(hipMemcpy(dA, h.data(), h.size() * 8, hipMemcpyHostToDevice));
        CHECK(hipEventRecord(e0, st));
        // the inverse of the inverse factor is not SPD input, so alternate between two copies is pointless: re-run on
        // the same (overwritten) block -- timing only, values are whatever they become
        for (int r = 0; r < reps; ++r) 
#ifdef LEAF_MFMA
            hipLaunchKernelGGL(cholinv_leaf_mfma_kernel<LEAF_NB>, dim3(1), dim3(LEAF_NB == 128 ? 512 : 320), 0, st, dA, ld, 0, dInfo);
#else
            hipLaunchKernelGGL(cholinv_leaf_kernel<LEAF_NB>, dim3(1), dim3((LEAF_NB / 4) * (LEAF_NB / 4)), 0, st, dA, ld, 0, dInfo);
#endif

        CHECK(hipEventRecord(e1, st));
        CHECK(hipEventSynchronize(e1));
        float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < best) best = ms;
    }
    printf("NB=%d ablate=%d  %.2f us per leaf (back to back, %d launches)\n", LEAF_NB, DCA_LEAF_ABLATE, best * 1e3 / reps, reps);
#ifdef DCA_LEAF_TRACE
    // phase stamps of thread DCA_LEAF_TRACE (100 MHz wall clock), last launch: 0 step start, 1 after its phase 1, 2 after
    // barrier A, 3 after its phase 2, 4 after barrier B, 5 after its phase 3
    unsigned long long tr[32 * 8];
    CHECK(hipMemcpyFromSymbol(tr, HIP_SYMBOL(g_leaf_trace), sizeof(tr)));
    printf("thread %d: step  phase1  waitA  phase2  waitB  phase3 [mfma leaf: a, B1, c, B2, chain, d]  (ns)\n", DCA_LEAF_TRACE);
    for (int kb = 0; kb < LEAF_NB / 4; ++kb) {
        const unsigned long long* r = tr + kb * 8;
        printf("   %2d  %6llu %6llu %6llu %6llu %6llu %6llu\n", kb, (r[1] - r[0]) * 10, (r[2] - r[1]) * 10, (r[3] - r[2]) * 10, (r[4] - r[3]) * 10, (r[5] - r[4]) * 10, (r[6] - r[5]) * 10);
    }
    printf("total loop %llu ns\n", (tr[(LEAF_NB / 4 - 1) * 8 + 5] - tr[0]) * 10);
#endif
    return 0;
}
