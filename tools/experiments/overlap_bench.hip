// Does a chain of single-workgroup leaves (stream A) run undisturbed next to a background product whose launches are
// sized below the chip (stream B)?  Product kernels of csrc/cholinv.hip, included as they are.
// hipcc --offload-arch=gfx950 -O3 -I../../include -I../../pydca_amd/csrc -o overlap_bench overlap_bench.hip
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
int main(int argc, char** argv)
{
    const int ld = 10048, leaves = 200;
    std::vector<double> h((size_t)128 * ld, 0.0);
    for (int i = 0; i < 128; ++i)
        for (int j = 0; j <= i; ++j) h[(size_t)i * ld + j] = (i == j) ? 4.0 + 0.01 * i : 0.3 / (1.0 + i - j);
    double *dA, *dB, *dC; int* dInfo;
    const size_t big = (size_t)4096 * 4096;
    CHECK(hipMalloc(&dA, h.size() * 8)); CHECK(hipMalloc(&dInfo, 4)); CHECK(hipMemset(dInfo, 0, 4));
    CHECK(hipMalloc(&dB, big * 8)); CHECK(hipMalloc(&dC, big * 8));
    CHECK(hipMemset(dB, 0, big * 8)); CHECK(hipMemset(dC, 0, big * 8));
    CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_nt_f64_dma_kernel<4>), hipFuncAttributeMaxDynamicSharedMemorySize, 4 * 128 * 16 * 8));
    hipStream_t sa, sb;
    int lo, hi; CHECK(hipDeviceGetStreamPriorityRange(&lo, &hi));
    CHECK(hipStreamCreateWithPriority(&sa, hipStreamNonBlocking, hi));     // chain: highest priority
    CHECK(hipStreamCreateWithPriority(&sb, hipStreamNonBlocking, lo));
    hipEvent_t a0, a1, b0, b1; CHECK(hipEventCreate(&a0)); CHECK(hipEventCreate(&a1)); CHECK(hipEventCreate(&b0)); CHECK(hipEventCreate(&b1));
    auto run_leaves = [&]() {
        CHECK(hipMemcpy(dA, h.data(), h.size() * 8, hipMemcpyHostToDevice));
        CHECK(hipEventRecord(a0, sa));
        for (int r = 0; r < leaves; ++r) hipLaunchKernelGGL(cholinv_leaf_mfma_kernel<128>, dim3(1), dim3(512), 0, sa, dA, ld, 0, dInfo);
        CHECK(hipEventRecord(a1, sa));
    };
    run_leaves(); CHECK(hipDeviceSynchronize());
    float ms; 
    run_leaves(); CHECK(hipEventSynchronize(a1)); CHECK(hipEventElapsedTime(&ms, a0, a1));
    printf("leaves alone: %.2f us per leaf\n", ms * 1e3 / leaves);
    const int grids[][2] = {{8, 16}, {12, 16}, {15, 16}, {15, 17}, {16, 16}, {16, 24}, {16, 32}, {32, 32}};
    for (int KC : {256, 1024}) {
        for (auto& gr : grids) {
            const int gx = gr[0], gy = gr[1];
            GemmArgs g{dB, 4096, MASK_NONE, dB + 2048, 4096, MASK_NONE, dC, 4096, nullptr, 0, gy * 128, gx * 128, KC, 1.0, 1.0, 0, WALK_ROWS};
            const int launches = (int)(40e9 / (2.0 * gx * gy * 128.0 * 128.0 * KC)) + 1;     // ~ enough work to cover the leaves
            // bg alone
            CHECK(hipEventRecord(b0, sb));
            for (int l = 0; l < launches; ++l) hipLaunchKernelGGL(gemm_nt_f64_dma_kernel<4>, dim3(gx, gy), dim3(256), (size_t)4 * 128 * 16 * 8, sb, g);
            CHECK(hipEventRecord(b1, sb)); CHECK(hipEventSynchronize(b1));
            float msb; CHECK(hipEventElapsedTime(&msb, b0, b1));
            const double flop = 2.0 * gx * gy * 128.0 * 128.0 * KC * launches;
            // together
            CHECK(hipMemcpy(dA, h.data(), h.size() * 8, hipMemcpyHostToDevice));
            CHECK(hipEventRecord(b0, sb));
            for (int l = 0; l < launches; ++l) hipLaunchKernelGGL(gemm_nt_f64_dma_kernel<4>, dim3(gx, gy), dim3(256), (size_t)4 * 128 * 16 * 8, sb, g);
            CHECK(hipEventRecord(b1, sb));
            CHECK(hipEventRecord(a0, sa));
            for (int r = 0; r < leaves; ++r) hipLaunchKernelGGL(cholinv_leaf_mfma_kernel<128>, dim3(1), dim3(512), 0, sa, dA, ld, 0, dInfo);
            CHECK(hipEventRecord(a1, sa));
            CHECK(hipDeviceSynchronize());
            float msa, msb2; CHECK(hipEventElapsedTime(&msa, a0, a1)); CHECK(hipEventElapsedTime(&msb2, b0, b1));
            printf("KC %4d grid %2d x %2d = %4d WGs x %3d launches: bg alone %.2f ms (%.1f TF); together: leaves %.2f us each, bg %.2f ms (%.1f TF)\n",
                   KC, gx, gy, gx * gy, launches, msb, flop / msb / 1e9, msa * 1e3 / leaves, msb2, flop / msb2 / 1e9);
        }
    }
    return 0;
}
