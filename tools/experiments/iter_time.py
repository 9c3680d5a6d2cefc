#!/usr/bin/env python3
"""Wall time per L-BFGS iteration at config D with and without the per-kernel HIP events (what do they cost?)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from pydca_amd import _lib  # noqa: E402
from tools.gen_msa import dedup, generate  # noqa: E402

X = dedup(generate(500, 50000, 21, 12346))
ctx = _lib.Context(0, _lib.DCA_F32)
ctx.set_msa(X, 21)
ctx.compute_weights(0.8, _lib.DCA_F32)
ctx.plm_configure(1.0, 50.0)
ctx.plm_init_x()
ctx.plm_lbfgs_begin(2000)
ctx.plm_lbfgs_iterate(5)
for prof in (True, False, True, False):
    ctx.set_profiling(prof)
    ctx.reset_kernel_times()
    t0 = time.perf_counter()
    st = ctx.plm_lbfgs_iterate(30)
    dt = time.perf_counter() - t0
    ksum = sum(ctx.kernel_time(k)[0] for k in ("plm_expand", "plm_logits", "plm_softmax", "plm_scatter", "plm_fold", "lbfgs_vec")) / 30 if prof else float("nan")
    print("events=%s  %.3f ms per iteration (kernel events sum %.3f)" % (prof, dt / 30 * 1e3, ksum))
