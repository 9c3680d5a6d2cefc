#!/usr/bin/env python3
"""float32 gradient error at config C against the float64 oracle as a function of the scatter kernel's tile-range split
(DCA_SCATTER_SPLIT / DCA_SCATTER_REM): how much of it is the length of the float32 accumulation chains."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import perturbed, rel_err  # noqa: E402
from oracle import plm as oracle_plm  # noqa: E402
from pydca_amd import _lib as L_  # noqa: E402
from tools.gen_msa import SEEDS, dedup, generate  # noqa: E402

X = dedup(generate(200, 10000, 21, SEEDS["C"]))
w = oracle_plm.weights(X, 0.8, np.float32)
x = perturbed(oracle_plm.init_x(X, w, 21), 200, 21)
fx_o, g_o = oracle_plm.gradient(X, w.astype(np.float64), 21, 1.0, 50.0, x.astype(np.float64), carry=True)
for rem in ("0", "1"):
    for split in ("1", "2", "4", "6"):
        os.environ["DCA_SCATTER_SPLIT"], os.environ["DCA_SCATTER_REM"] = split, rem
        ctx = L_.Context(0, L_.DCA_F32)
        ctx.set_msa(X, 21)
        ctx.compute_weights(0.8, L_.DCA_F32)
        ctx.plm_configure(1.0, 50.0)
        ctx.plm_set_x(x)
        fx = ctx.plm_gradient()
        g = ctx.plm_get_g(np.float64)
        d = np.abs(g - g_o)
        print("rem %s split %s: rel_err %.3e  max abs %.3e  fx rel %.2e" % (rem, split, rel_err(g, g_o), d.max(), abs(fx - fx_o) / abs(fx_o)), flush=True)
        ctx.close()
