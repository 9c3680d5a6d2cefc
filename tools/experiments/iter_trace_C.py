#!/usr/bin/env python3
"""A few L-BFGS iterations at config C (or DCA_ITER_CFG=E) without profiling events, for rocprofv3 --kernel-trace + gap_trace.py."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from pydca_amd import _lib  # noqa: E402
from tools.gen_msa import dedup, generate  # noqa: E402

cfg = {"C": (200, 10000, 21, 12345, 1.0, 50.0), "E": (150, 200000, 5, 12347, 29.8, 29.8)}[os.environ.get("DCA_ITER_CFG", "C")]
X = dedup(generate(*cfg[:4]))
ctx = _lib.Context(0, _lib.DCA_F32)
ctx.set_msa(X, cfg[2])
ctx.compute_weights(0.8, _lib.DCA_F32)
ctx.plm_configure(cfg[4], cfg[5])
ctx.plm_init_x()
ctx.plm_lbfgs_begin(2000)
ctx.plm_lbfgs_iterate(30)
