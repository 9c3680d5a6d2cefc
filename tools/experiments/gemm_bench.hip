// Stand-alone timing of the f64 NT GEMM of csrc/cholinv.hip (the file is included, so the kernel
// under test is the product kernel): plain and triangular-operand products at the sizes of the inverse.
// hipcc --offload-arch=gfx950 -O3 -I../../include -I../../pydca_amd/csrc -o gemm_bench gemm_bench.hip
#include <unistd.h>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include "../../pydca_amd/csrc/cholinv.hip"
void dca_set_error(const char* fmt, ...) { va_list ap; va_start(ap, fmt); vfprintf(stderr, fmt, ap); va_end(ap); }
void dca_flush_clocks(dca_ctx*) {}
hipError_t dca_dev_malloc(void** p, size_t b, bool) { return hipMalloc(p, b); }
hipError_t dca_dev_free(void* p) { return hipFree(p); }
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
__global__ void fill_kernel(double* p, size_t n, unsigned seed)
{
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        unsigned h = (unsigned)i * 2654435761u + seed * 40503u; h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
        p[i] = (double)(h & 0xffffff) / 16777216.0 - 0.5;
    }
}
__global__ void tiny_kernel(double* p, int spin)
{
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < spin / 10) { }
    if (p == nullptr) p[0] = 0;
}

int main(int argc, char** argv)
{
    const int n = argc > 1 ? atoi(argv[1]) : 5056;
    dca_ctx ctx;
    CHECK(hipStreamCreate(&ctx.stream));
    if (gemm_kernels_prepare(0) != DCA_OK) return 1;      // dynamic-LDS attributes of the product kernels (per device)
    double *A, *B, *C;
    CHECK(hipMalloc(&A, (size_t)n * n * 8)); CHECK(hipMalloc(&B, (size_t)n * n * 8)); CHECK(hipMalloc(&C, (size_t)n * n * 8));
    // non-trivial operand values: an all-zero GEMM draws less power and clocks higher than the real thing
    hipLaunchKernelGGL(fill_kernel, dim3(1024), dim3(256), 0, ctx.stream, A, (size_t)n * n, 1u);
    hipLaunchKernelGGL(fill_kernel, dim3(1024), dim3(256), 0, ctx.stream, B, (size_t)n * n, 2u);
    CHECK(hipStreamSynchronize(ctx.stream));
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    double* Cmirror = nullptr;
    struct Case { const char* name; int maskA, maskB, lower; double flops; int mirror = 0; } cases[] = {
        {"plain NT", MASK_NONE, MASK_NONE, 0, 2.0 * n * (double)n * n},
        {"B lower-triangular (TRSM-like)", MASK_NONE, MASK_LOWER, 0, 1.0 * n * (double)n * n},
        {"SYRK lower tiles", MASK_NONE, MASK_NONE, 1, 1.0 * n * (double)n * n},
        {"X^T X (both upper, lower tiles)", MASK_UPPER, MASK_UPPER, 1, 2.0 / 6.0 * n * (double)n * n * 1.0},
        {"X^T X + mirrored upper part", MASK_UPPER, MASK_UPPER, 1, 2.0 / 6.0 * n * (double)n * n * 1.0, 1},
    };
    for (auto& c : cases) {
        float best = 1e9;
        for (int rep = 0; rep < 3; ++rep) {
            CHECK(hipEventRecord(e0, ctx.stream));
            launch_gemm(&ctx, GemmArgs{A, n, c.maskA, B, n, c.maskB, C, n, c.mirror ? C : nullptr, c.mirror ? n : 0, n, n, n, c.mirror ? -1.0 : 1.0, 0.0, c.lower});
            CHECK(hipEventRecord(e1, ctx.stream));
            CHECK(hipEventSynchronize(e1));
            float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
            if (ms < best) best = ms;
        }
        printf("n=%d %-34s %.3f ms  %.1f TFLOP/s (useful flops)\n", n, c.name, best, c.flops / (best * 1e-3) / 1e12);
    }
    // the final product of the inverse in its own situation: both operands the SAME matrix, one launch, after a stretch of
    // tiny single-workgroup kernels (the tail of the recursion) or after idling
    auto once = [&](const char* what, double* Bop, int tiny, int sleep_us) {
        if (sleep_us) usleep(sleep_us);
        for (int i = 0; i < tiny; ++i) hipLaunchKernelGGL(tiny_kernel, dim3(1), dim3(64), 0, ctx.stream, C, 4000);
        CHECK(hipEventRecord(e0, ctx.stream));
        launch_gemm(&ctx, GemmArgs{A, n, MASK_UPPER, Bop, n, MASK_UPPER, C, n, C, n, n, n, n, -1.0, 0.0, 1});
        CHECK(hipEventRecord(e1, ctx.stream));
        CHECK(hipEventSynchronize(e1));
        float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
        printf("n=%d X^T X once, %-44s %.3f ms\n", n, what, ms);
    };
    once("B != A, back to back", B, 0, 0);
    once("B != A, back to back", B, 0, 0);
    once("B == A, back to back", A, 0, 0);
    once("B == A, back to back", A, 0, 0);
    once("B == A, after 5 ms host sleep", A, 0, 5000);
    once("B == A, after 100 ms host sleep", A, 0, 100000);
    once("B == A, after 200 tiny kernels (~1 ms)", A, 200, 0);
    once("B == A, after 1000 tiny kernels (~5 ms)", A, 1000, 0);
    once("B == A, back to back", A, 0, 0);
    fflush(stdout);
    return 0;
}
