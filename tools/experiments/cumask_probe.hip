// Which CUs does a stream created with hipExtStreamCreateWithCUMask use on an 8-XCD part?  Every workgroup records
// (XCC_ID, HW_ID: SE / CU) and spins ~20 us so that the grid spreads over everything it is allowed to use.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <vector>
#include <map>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
__global__ void probe(uint32_t* out, int spin)
{
    uint32_t hw, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < spin) { }
    if (threadIdx.x == 0) { out[2 * blockIdx.x] = hw; out[2 * blockIdx.x + 1] = xcc; }
}
int main(int argc, char** argv)
{
    const int words = 8;
    std::vector<uint32_t> mask(words, 0);
    const int keep = argc > 1 ? atoi(argv[1]) : 240;          // bits 0 .. keep-1 set
    for (int i = 0; i < keep; ++i) mask[i / 32] |= 1u << (i % 32);
    hipStream_t s;
    CHECK(hipExtStreamCreateWithCUMask(&s, words, mask.data()));
    const int blocks = 4096;
    uint32_t* d; CHECK(hipMalloc(&d, blocks * 8));
    hipLaunchKernelGGL(probe, dim3(blocks), dim3(256), 0, s, d, 2000);
    CHECK(hipStreamSynchronize(s));
    std::vector<uint32_t> h(blocks * 2);
    CHECK(hipMemcpy(h.data(), d, blocks * 8, hipMemcpyDeviceToHost));
    std::map<uint32_t, std::map<uint32_t, int>> used;      // xcc -> (se, cu) -> count
    for (int b = 0; b < blocks; ++b) {
        const uint32_t hw = h[2 * b], xcc = h[2 * b + 1] & 0xf;
        const uint32_t cu = (hw >> 8) & 0xf, sh = (hw >> 12) & 0x1, se = (hw >> 13) & 0x7;
        used[xcc][(se << 8) | (sh << 4) | cu]++;
    }
    int total = 0;
    for (auto& x : used) { printf("xcc %u: %zu distinct (se,sh,cu)\n", x.first, x.second.size()); total += (int)x.second.size(); }
    printf("mask bits set %d -> %d distinct CUs used\n", keep, total);
    return 0;
}
