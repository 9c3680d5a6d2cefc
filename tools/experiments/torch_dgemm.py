#!/usr/bin/env python3
"""What the vendor library reaches for a plain f64 GEMM on this GPU (reference point for cholinv.hip's kernel;
torch.mm -> rocBLAS/hipBLASLt).  Not used by the product."""
import time
import torch
for n in (5056, 10048):
    a = torch.randn(n, n, dtype=torch.float64, device="cuda")
    b = torch.randn(n, n, dtype=torch.float64, device="cuda")
    for _ in range(2):
        c = a @ b.T
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    reps = 5
    for _ in range(reps):
        c = a @ b.T
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    print("n=%d  %.2f ms  %.1f TFLOP/s" % (n, dt * 1e3, 2 * n ** 3 / dt / 1e12))
