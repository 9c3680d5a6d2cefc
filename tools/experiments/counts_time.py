#!/usr/bin/env python3
"""mf_counts stage time at config D, whatever the numerical result (for ablated builds of the counts kernel)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from pydca_amd import _lib  # noqa: E402
from tools.gen_msa import dedup, generate  # noqa: E402

X = dedup(generate(500, 50000, 21, 12346))
for rep in range(3):
    ctx = _lib.Context(0, _lib.DCA_F64)
    ctx.set_msa(X, 21)
    ctx.set_profiling(True)
    ctx.compute_weights(0.8, _lib.DCA_F64)
    try:
        ctx.mf_corr_mat(0.5, want=False)
    except Exception as exc:
        print("error", exc)
    print("rep", rep, {k: round(ctx.kernel_time(k)[0], 3) for k in ("mf_sort", "mf_counts")})
    ctx.close()
