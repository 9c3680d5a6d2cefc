#!/usr/bin/env python3
"""Time the kernels of one plmDCA objective/gradient evaluation (no optimiser)."""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pydca_amd import _lib  # noqa: E402
from tools.gen_msa import dedup, generate  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--L", type=int, default=500)
ap.add_argument("--N", type=int, default=50000)
ap.add_argument("--q", type=int, default=21)
ap.add_argument("--seed", type=int, default=12346)
ap.add_argument("--reps", type=int, default=5)
ap.add_argument("--precision", type=int, default=32)
ap.add_argument("--iters", type=int, default=0)
a = ap.parse_args()
X = dedup(generate(a.L, a.N, a.q, a.seed))
ctx = _lib.Context(0, a.precision)
ctx.set_msa(X, a.q)
ctx.compute_weights(0.8, _lib.DCA_F32)
ctx.plm_configure(1.0, 50.0, chunk=int(os.environ.get('DCA_CHUNK', '0')))
ctx.plm_init_x()
if a.iters:
    ctx.plm_lbfgs_begin(1000)
    st = ctx.plm_lbfgs_iterate(a.iters)
    print('lbfgs', st.iterations, st.status, st.fx)
ctx.plm_gradient()
ctx.set_profiling(True)
ctx.reset_kernel_times()
for _ in range(a.reps):
    fx = ctx.plm_gradient()
out = {k: round(ctx.kernel_time(k)[0] / a.reps, 3) for k in ("plm_expand", "plm_logits", "plm_softmax", "plm_scatter", "plm_fold")}
print(out, "fx=%.6g" % fx)
