// Round 5: the deep update of the blocked inverse (M x 512 output, deep k) alone on the GPU -- the 128 x 128-tile kernel as one
// workgroup per tile (what the bulk stream launches) against the stream-K form, and the split-k form.
// hipcc --offload-arch=gfx950 -O3 -Wno-unused-result -o streamk_bench streamk_bench.hip && ./streamk_bench
#include <unistd.h>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include "../../pydca_amd/csrc/cholinv.hip"
void dca_set_error(const char* fmt, ...) { va_list ap; va_start(ap, fmt); vfprintf(stderr, fmt, ap); va_end(ap); }
void dca_flush_clocks(dca_ctx*) {}
hipError_t dca_dev_malloc(void** p, size_t b, bool) { return hipMalloc(p, b); }
hipError_t dca_dev_free(void* p) { return hipFree(p); }
__global__ void fill_kernel(double* p, size_t n, unsigned seed)
{
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        unsigned h = (unsigned)i * 2654435761u + seed * 40503u; h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
        p[i] = (double)(h & 0xffffff) / 16777216.0 - 0.5;
    }
}
int main()
{
    const int n = 10048;
    hipStream_t st; hipStreamCreate(&st);
    if (gemm_kernels_prepare(0) != DCA_OK) return 1;
    double *L, *C, *P;
    hipMalloc(&L, (size_t)n * n * 8); hipMalloc(&C, (size_t)n * n * 8); hipMalloc(&P, (size_t)n * n * 4);
    hipLaunchKernelGGL(fill_kernel, dim3(1024), dim3(256), 0, st, L, (size_t)n * n, 1u);
    hipStreamSynchronize(st);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    struct Shape { int M, K; } shapes[] = {{8000, 1024}, {6464, 3072}, {5440, 4096}, {3392, 6144}, {1344, 8192}, {9024, 512}, {5000, 512}};
    for (auto sh : shapes) {
        const double flop = 2.0 * sh.M * 512.0 * sh.K;
        for (int mode = 0; mode < 4; ++mode) {
            float best = 1e9;
            for (int rep = 0; rep < 4; ++rep) {
                hipEventRecord(e0, st);
                if (mode == 0) launch_gemm_capped(st, GemmArgs{L, n, MASK_NONE, L + (size_t)5000 * n, n, MASK_NONE, C, n, nullptr, 0, sh.M, 512, sh.K, -1.0, 1.0, 0}, 248);
                else if (mode == 1) launch_gemm_splitk_capped(st, L, n, L + (size_t)5000 * n, n, C, n, sh.M, 512, sh.K, 2, P, 2000);
                else launch_gemm_streamk(st, L, n, L + (size_t)5000 * n, n, MASK_NONE, C, n, sh.M, 512, sh.K, -1.0, 1.0, P, (size_t)n * n / 2, mode == 2 ? 248 : 256);
                hipEventRecord(e1, st);
                hipEventSynchronize(e1);
                float ms; hipEventElapsedTime(&ms, e0, e1);
                if (ms < best) best = ms;
            }
            const char* names[] = {"one workgroup per tile, cap 248", "split-k 2, uncapped", "stream-K 248", "stream-K 256"};
            printf("M=%5d K=%5d  %-34s %7.1f us  %5.1f TF\n", sh.M, sh.K, names[mode], best * 1e3, flop / (best * 1e-3) / 1e12);
        }
    }
    return 0;
}
