// Round 5 microbenchmark: what the LDS sustains for the pair-count histogram's update pattern (mf_counts_kernel):
// every lane owns a column of q = 21 double slots (slot = b * 256 + t) and adds a weight into slot b of its column.
//   mode 0: ds_add_f64 (fire and forget)          mode 1: ds_read_b64 + v_add_f64 + ds_write_b64 on ONE histogram
//   NH > 1: the same on several independent histograms, interleaved (several chains in flight)
//   mode 3 / 4: ds_add_u64 / ds_add_u32 (integer atomics: what a fixed-point accumulation would issue)
// Prints wave-instructions' lane-updates per clock and CU at the sustained clock (s_memtime ticks at 100 MHz).
//   hipcc --offload-arch=gfx950 -O3 -o lds_atomic_rate lds_atomic_rate.hip && ./lds_atomic_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

template <int MODE, int NH>
__global__ __launch_bounds__(256) void k(double* out, int iters, unsigned seed)
{
    extern __shared__ double hist[];          // NH x [21][256]
    const int t = threadIdx.x;
    for (int e = t; e < NH * 21 * 256; e += 256) hist[e] = 0.0;
    __syncthreads();
    unsigned s = seed * 2654435761u + t * 40503u + blockIdx.x;
    typedef __attribute__((address_space(3))) double* lds_ptr;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 32; ++u) {
            // cheap indices (about three VALU instructions per update, so that the LDS and not the index arithmetic is measured):
            // one LCG step per four updates, a 4-bit field each (16 of the 21 slots)
            if (u % 4 == 0) s = s * 1664525u + 1013904223u;
            const int b = (s >> (8 * (u % 4) + 4)) & 15;
            const double w = 1.0;
            double* p = &hist[((u % NH) * 21 + b) * 256 + t];
            if (MODE == 0) __builtin_amdgcn_ds_atomic_fadd_f64((lds_ptr)p, w);
            else if (MODE == 3) atomicAdd(reinterpret_cast<unsigned long long*>(p), 1ull << 20);     // ds_add_u64
            else if (MODE == 4) atomicAdd(reinterpret_cast<unsigned*>(p), 1u);                       // ds_add_u32
            else *p = *p + w;
        }
    }
    __syncthreads();
    double a = 0;
    for (int e = t; e < NH * 21 * 256; e += 256) a += hist[e];
    out[blockIdx.x * 256 + t] = a;
}

template <int MODE, int NH>
void run(const char* name, int wgPerCu)
{
    const int iters = 400, wgs = 256 * wgPerCu;
    double* out;
    hipMalloc(&out, (size_t)wgs * 256 * 8);
    const size_t lds = (size_t)NH * 21 * 256 * 8;
    hipFuncSetAttribute(reinterpret_cast<const void*>(k<MODE, NH>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k<MODE, NH>), dim3(wgs), dim3(256), lds, 0, out, 10, 1u);
    hipDeviceSynchronize();
    float best = 1e30f;
    for (int r = 0; r < 3; ++r) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((k<MODE, NH>), dim3(wgs), dim3(256), lds, 0, out, iters, 2u + r);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    const double updates = (double)wgs * 256 * iters * 32;
    printf("%-44s %d WG/CU (%zu KB LDS): %.3f ms, %.2f T updates/s, %.2f lane-updates per clk and CU at 2.1 GHz\n", name, wgPerCu, lds / 1024, best,
           updates / best / 1e9, updates / (best * 1e-3) / 256 / 2.1e9);
    hipFree(out);
}

int main()
{
    run<0, 1>("ds_add_f64, one histogram", 3);
    run<0, 1>("ds_add_f64, one histogram", 1);
    run<1, 1>("read + add + write, one histogram", 3);
    run<1, 1>("read + add + write, one histogram", 1);
    run<1, 2>("read + add + write, two histograms", 1);
    run<1, 3>("read + add + write, three histograms", 1);
    run<0, 3>("ds_add_f64, three histograms", 1);
    run<3, 1>("ds_add_u64, one histogram", 3);
    run<3, 1>("ds_add_u64, one histogram", 1);
    run<4, 1>("ds_add_u32, one histogram", 3);
    return 0;
}
