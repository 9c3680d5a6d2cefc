#!/usr/bin/env python3
"""Round 6: where the block sweep's result differs from LAPACK -- worst relative error per 64 x 64 block."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from pydca_amd import _lib
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1088
rng = np.random.default_rng(n)
B = rng.standard_normal((n, n + 8))
A = B @ B.T / n + 0.5 * np.diag(rng.random(n) + 0.5)
ctx = _lib.Context(0, _lib.DCA_F64)
inv = ctx.spd_inverse(A)
ctx.close()
ref = np.linalg.inv(A)
scale = np.abs(ref).max()
nb = (n + 63) // 64
print("n =", n, "overall", np.linalg.norm(inv - ref) / np.linalg.norm(ref))
for i in range(nb):
    row = []
    for j in range(nb):
        e = np.abs(inv[64 * i:64 * i + 64, 64 * j:64 * j + 64] - ref[64 * i:64 * i + 64, 64 * j:64 * j + 64]).max() / scale
        row.append("." if e < 1e-12 else "x" if e < 1e-3 else "X")
    print("".join(row))
