#!/usr/bin/env python3
"""Do LDS -> register row reads and packed adds of the SAME SIMD overlap?  Generates and times kernels whose waves repeat
{K ds_read_b64 of 512-byte rows into one register set, M independent v_pk_add_f32 reading the OTHER set, one lgkmcnt(0)}:
if the two overlapped, an iteration would cost max(reads, adds); the gather kernels' timings (logits
t = 3.15 ms + 295 ms / sequences per wave) say it costs their SUM.  Prints SIMD clocks per iteration at 2 and 4 waves per SIMD.
    python lds_valu_overlap.py            (builds /tmp/lds_valu_overlap with hipcc, runs it; needs an MI355X)"""
import os
import subprocess

CASES = [(0, 96), (21, 0), (21, 96), (10, 96), (21, 48), (8, 16), (0, 16), (8, 0)]
src = ['#include <hip/hip_runtime.h>', '#include <cstdio>', '#include <vector>']
for K, M in CASES:
    body = []
    for half in range(2):
        rd, ad = (40, 100) if half == 0 else (100, 40)          # read into one set of row registers, add from the other
        for k in range(K):
            body.append("ds_read_b64 v[%d:%d], %%[vb] offset:%d" % (rd + 2 * k, rd + 2 * k + 1, (half * 21 + k) * 512))
        for m in range(M):
            a = 160 + 2 * (m % 48)
            body.append("v_pk_add_f32 v[%d:%d], v[%d:%d], v[%d:%d]" % (a, a + 1, a, a + 1, ad + 2 * (m % 21), ad + 2 * (m % 21) + 1))
        body.append("s_waitcnt lgkmcnt(0)")
    clob = ", ".join('"v%d"' % r for r in range(40, 256))
    src.append("__global__ __launch_bounds__(512) void k_%d_%d(float* out, int iters) {" % (K, M))
    src.append("  extern __shared__ char smem[]; unsigned vb = (unsigned)(uintptr_t)smem + (threadIdx.x & 63) * 8;")
    src.append("  for (int i = threadIdx.x; i < 16384; i += blockDim.x) reinterpret_cast<float*>(smem)[i] = 1.0f; __syncthreads();")
    src.append("  for (int it = 0; it < iters; ++it) asm volatile(\"" + "\\n\\t".join(body) + "\" :: [vb] \"v\"(vb) : \"memory\", " + clob + ");")
    src.append("  if (iters < 0) out[threadIdx.x] = 0.f; }")
src.append("""
template <typename F> double run(F kern, int threads, int iters) {
  float* d; hipMalloc(&d, 4096);
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
  hipLaunchKernelGGL(kern, dim3(256), dim3(threads), 65536, 0, d, 100);
  hipEventRecord(a); hipLaunchKernelGGL(kern, dim3(256), dim3(threads), 65536, 0, d, iters); hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b); hipFree(d); return ms; }
int main() { const int iters = 20000; const double ghz = 2.4;""")
for K, M in CASES:
    src.append('  for (int t : {256, 512}) { double ms = run(k_%d_%d, t, iters); double clk = ms * 1e-3 * ghz * 1e9 / (2.0 * iters) / (t / 256);' % (K, M))
    src.append('    printf("K=%%2d reads M=%%2d adds, %%d waves/SIMD: %%7.1f clk@2.4GHz per (K reads + M adds) per wave  -> sum model %%d, max model %%d\\n", %d, %d, t / 256, clk, %d, %d); }' % (K, M, int(8 * K + 4.4 * M), int(max(8 * K, 4.4 * M))))
src.append("  return 0; }")
open("/tmp/lds_valu_overlap.hip", "w").write("\n".join(src))
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O2", "-w", "-o", "/tmp/lds_valu_overlap", "/tmp/lds_valu_overlap.hip"])
if os.path.exists("/dev/kfd"):
    subprocess.check_call(["/tmp/lds_valu_overlap"])
