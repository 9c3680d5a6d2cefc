#!/usr/bin/env python3
"""Time the SPD inverse (csrc/cholinv.hip) alone on a random SPD matrix: device time of the whole inverse, of the
recursion / factorisation part and of the final X^T X, from the library's own HIP events.  Environment knobs of
cholinv.hip (DCA_CHOLINV_*) are read once per process, so one process per variant.

    python tools/time_inv.py --n 10048 --reps 4 [--check]"""
import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pydca_amd import _lib  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, default=10048)
ap.add_argument("--reps", type=int, default=4)
ap.add_argument("--check", action="store_true", help="compare with numpy.linalg.inv")
ap.add_argument("--tag", default="")
a = ap.parse_args()
rng = np.random.default_rng(a.n)
B = rng.standard_normal((a.n, a.n + 8))
A = B @ B.T / a.n + 0.5 * np.diag(rng.random(a.n) + 0.5)
del B
ctx = _lib.Context(0, _lib.DCA_F64)
ctx.set_profiling(True)
best = None
for rep in range(a.reps):
    ctx.reset_kernel_times()
    t0 = time.perf_counter()
    inv = ctx.spd_inverse(A)
    dt = time.perf_counter() - t0
    t = {k: ctx.kernel_time(k)[0] for k in ("mf_inverse", "mf_inverse_recursion", "mf_inverse_xtx")}
    if best is None or t["mf_inverse"] < best["mf_inverse"]:
        best = t
    print("%s n=%d rep %d inverse %.2f ms (factor+trtri %.2f, xtx %.2f)  %.1f TF  host call %.0f ms" % (
        a.tag, a.n, rep, t["mf_inverse"], t["mf_inverse_recursion"], t["mf_inverse_xtx"], a.n ** 3 / t["mf_inverse"] / 1e9, dt * 1e3), flush=True)
print("%s n=%d BEST inverse %.2f ms (factor+trtri %.2f, xtx %.2f)  %.1f TF = %.3f of 78.6" % (
    a.tag, a.n, best["mf_inverse"], best["mf_inverse_recursion"], best["mf_inverse_xtx"], a.n ** 3 / best["mf_inverse"] / 1e9,
    a.n ** 3 / best["mf_inverse"] / 1e9 / 78.6), flush=True)
if a.check:
    ref = np.linalg.inv(A)
    print("%s rel. error vs LAPACK %.2e, symmetric %s" % (a.tag, np.linalg.norm(inv - ref) / np.linalg.norm(ref), np.array_equal(inv, inv.T)))
ctx.close()
