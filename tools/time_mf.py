#!/usr/bin/env python3
"""Time the mfDCA compute_fn chain (encoded MSA on host -> ranked FN_APC) and its stages."""
import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pydca_amd import _lib  # noqa: E402
from tools.gen_msa import dedup, generate  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--L", type=int, default=500)
ap.add_argument("--N", type=int, default=50000)
ap.add_argument("--q", type=int, default=21)
ap.add_argument("--seed", type=int, default=12346)
ap.add_argument("--reps", type=int, default=2)
a = ap.parse_args()
X = dedup(generate(a.L, a.N, a.q, a.seed))
for rep in range(a.reps):
    ctx = _lib.Context(0, _lib.DCA_F64)
    t0 = time.perf_counter()
    ctx.set_msa(X, a.q)
    t1 = time.perf_counter()
    ctx.set_profiling(True)
    ctx.compute_weights(0.8, _lib.DCA_F64)
    t2 = time.perf_counter()
    scores = ctx.mf_run(0.5, True)
    t3 = time.perf_counter()
    order = ctx.scores_order()
    dt = time.perf_counter() - t0
    print("rep", rep, "total %.1f ms" % (dt * 1e3), {k: round(ctx.kernel_time(k)[0], 2) for k in ("weights", "mf_counts", "mf_inverse", "mf_inverse_recursion", "mf_inverse_xtx", "scores")},
          "pairs/s %.0f" % (a.L * (a.L - 1) / 2 / dt),
          "host ms: set_msa %.1f weights %.1f mf_run %.1f order %.1f" % ((t1 - t0) * 1e3, (t2 - t1) * 1e3, (t3 - t2) * 1e3, (t0 + dt - t3) * 1e3))
    ctx.close()
