// Round 6: the block sweep's update kernel alone (one REST launch of a middle panel, one PRIO launch) by operand stages and
// workgroup cap.   hipcc --offload-arch=gfx950 -O3 -std=c++17 -Wno-unused-result -I../../include [-DDCA_SWEEP_ABLATE=k] -o sweep_bench sweep_bench.hip
//   ./sweep_bench n w
#include <unistd.h>
#include <cmath>
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
        p[i] = ((double)(h & 0xffffff) / 16777216.0 - 0.5) * 1e-3;
    }
}
int main(int argc, char** argv)
{
    const int n = argc > 1 ? atoi(argv[1]) : 10048, w = argc > 2 ? atoi(argv[2]) : 512;
    hipStream_t st; hipStreamCreate(&st);
    if (sweep_kernels_prepare(0) != DCA_OK) return 1;
    double *M, *W;
    hipMalloc(&M, (size_t)n * n * 8); hipMalloc(&W, (size_t)n * w * 8);
    hipLaunchKernelGGL(fill_kernel, dim3(1024), dim3(256), 0, st, M, (size_t)n * n, 1u);
    hipLaunchKernelGGL(fill_kernel, dim3(1024), dim3(256), 0, st, W, (size_t)n * w, 2u);
    hipStreamSynchronize(st);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int nt = (n + 127) / 128;
    const int c = (nt / 2) * 128 / w * w;                   // a middle panel
    const int c1 = c + w, w1 = w, c2 = c1 + w1, w2 = w;
    int* ctr; hipMalloc(&ctr, 64);
    SweepArgs g{W, w, M + c, n, M, n, n, c, w, nt, c / 128, (c1 + w1) / 128 - c / 128, SWEEP_REST, c1 / 128, w1 / 128, c2 / 128, w2 / 128, 0, ctr};
    const int nR = nt - g.skipN;
    for (int mode = 0; mode < 2; ++mode) {
        g.mode = mode == 0 ? SWEEP_REST : SWEEP_PRIO;
        const int bands = (nR + 3) / 4;
        g.nTiles = mode == 0 ? 8 * bands * bands + 2 * bands : g.prN * nR + g.dgN * (g.dgN + 1) / 2;
        const double tiles = mode == 0 ? nR * (nR + 1) / 2.0 - g.dgN * (g.dgN + 1) / 2.0 : g.nTiles;
        const double flop = tiles * 2.0 * 128 * 128 * w;
        for (int perCu = 1; perCu <= 2; ++perCu)
        for (int stages = 2; stages <= (perCu == 2 ? 2 : 4); stages += 2)
            for (int cap : {248, 256, 496, 512, 100000}) {
                if (perCu == 1 && cap > 256 && cap < 100000) continue;
                float best = 1e9;
                const int G = std::min(cap, (g.nTiles + 7) / 8 * 8);
                for (int rep = 0; rep < 4; ++rep) {
                    hipMemsetAsync(ctr, 0, 64, st);
                    hipEventRecord(e0, st);
                    sweep_update_launch(st, G, stages, perCu, g);
                    hipEventRecord(e1, st);
                    hipEventSynchronize(e1);
                    float ms; hipEventElapsedTime(&ms, e0, e1);
                    if (ms < best) best = ms;
                }
                printf("n=%d w=%d %s tiles=%.0f perCu=%d stages=%d G=%5d  %8.1f us  %5.1f TF  (%.1f us per round)\n", n, w, mode == 0 ? "REST" : "PRIO", tiles, perCu, stages, G,
                       best * 1e3, flop / (best * 1e-3) / 1e12, best * 1e3 / ceil(tiles / G));
            }
    }
    if (hipGetLastError() != hipSuccess) printf("launch error\n");
    return 0;
}
