#!/usr/bin/env python3
"""float64 device path against the float64 oracle at FIXED x: how many gradient elements are equal to the last bit, and how
far apart the others are (the two exp() implementations differ in the last place; every sum is formed in the same order
or order-independently).  Analysis tool (uses oracle/).   python tests/analysis/f64_bits.py [C] [D] [E]"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import golden, perturbed  # noqa: E402
from oracle import plm as oplm  # noqa: E402
from pydca_amd import _lib  # noqa: E402
from tools.gen_msa import SEEDS, dedup, generate  # noqa: E402

FULL = {"C": (200, 10000, 21, 1.0, 50.0), "D": (500, 50000, 21, 1.0, 50.0), "E": (150, 200000, 5, 29.8, 29.8)}
cases = []
for tag, lh, lJ in (("plm_toy_rna", 1.8, 1.8), ("plm_toy_protein", 1.0, 5.0), ("plm_rf71", 1.0, 20.0)):
    G = golden(tag)
    cases.append((tag, G["X"], int(G["q"]), lh, lJ))
for cfg in sys.argv[1:]:
    L, N, q, lh, lJ = FULL[cfg]
    cases.append(("config_" + cfg, dedup(generate(L, N, q, SEEDS[cfg])), q, lh, lJ))
out = {}
for tag, X, q, lh, lJ in cases:
    L = X.shape[1]
    ctx = _lib.Context(0, _lib.DCA_F64)
    ctx.set_msa(X, q)
    ctx.compute_weights(0.8, _lib.DCA_F64)
    w = ctx.weights()
    x0 = oplm.init_x(X, w, q)
    for name, x in (("x0", x0), ("x1", perturbed(x0, L, q))):
        for mode, mname in ((_lib.CARRY_CHUNKED, "chunked"), (_lib.CARRY_SERIAL, "serial"), (_lib.CARRY_EXACT, "exact")):
            if mname != "chunked" and X.shape[0] > 20000:
                continue
            ctx.plm_configure(lh, lJ, mode)
            ctx.plm_set_x(x)
            fx = ctx.plm_gradient()
            g = ctx.plm_get_g(np.float64)
            fx_o, g_o = oplm.gradient(X, w, q, lh, lJ, x, carry=(mode != _lib.CARRY_EXACT))
            neq = g != g_o
            r = {"fx_equal": bool(fx == fx_o), "fx_rel": abs(fx - fx_o) / abs(fx_o), "g_unequal_fraction": float(neq.mean()),
                 "g_rel_err_norm": float(np.linalg.norm(g - g_o) / np.linalg.norm(g_o)),
                 "g_max_ulps": float(np.max(np.abs(g - g_o) / np.maximum(np.spacing(np.abs(g_o)), 1e-300))) if neq.any() else 0.0,
                 "h_unequal": int(neq[:L * q].sum()), "J_unequal": int(neq[L * q:].sum())}
            out["%s %s %s" % (tag, name, mname)] = r
            print(tag, name, mname, json.dumps(r), flush=True)
    ctx.close()
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump(out, open(os.path.join(ROOT, "gpurun_out", "f64_bits.json"), "w"), indent=1)
