#!/usr/bin/env python3
"""Config E, float64, ONE evaluation at x0 and at a perturbed point: chunked scan (warm-up 40 / 80, chunks 256 / 128 / 64) and the
serial chain against each other and the oracle.  (analysis tool: uses oracle/)"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import perturbed  # noqa: E402
from oracle import plm as oplm  # noqa: E402
from pydca_amd import _lib  # noqa: E402
from tools.gen_msa import SEEDS, dedup, generate  # noqa: E402

L, N, q, lh, lJ = 150, 200000, 5, 29.8, 29.8
X = dedup(generate(L, N, q, SEEDS["E"]))
res = {}
for name, mode, chunk, warm in (("chunked40", 1, 0, 40), ("chunked80", 1, 0, 80), ("chunk128", 1, 128, 40), ("chunk64", 1, 64, 40), ("serial", 2, 0, 0)):
    ctx = _lib.Context(0, _lib.DCA_F64)
    ctx.set_msa(X, q)
    ctx.compute_weights(0.8, _lib.DCA_F64)
    w = ctx.weights()
    ctx.plm_configure(lh, lJ, mode, chunk, warm)
    ctx.plm_init_x()
    x0 = ctx.plm_get_x(np.float64)
    out = []
    for x in (x0, perturbed(x0, L, q)):
        ctx.plm_set_x(x)
        fx = ctx.plm_gradient()
        out.append((fx, ctx.plm_get_g(np.float64)))
    res[name] = out
    ctx.close()
x0o = oplm.init_x(X, w, q)
res["oracle"] = [oplm.gradient(X, w, q, lh, lJ, x, carry=True) for x in (x0o, perturbed(x0o, L, q))]
print("x0 device vs oracle:", float(np.max(np.abs(x0 - x0o))))
names = list(res)
for pt in (0, 1):
    print("point", pt)
    for a in range(len(names)):
        for b in range(a + 1, len(names)):
            fa, ga = res[names[a]][pt]
            fb, gb = res[names[b]][pt]
            print("  %-10s vs %-10s  fx %.3e   g %.3e   max|dg| %.3e" % (names[a], names[b], abs(fa - fb) / abs(fb), np.linalg.norm(ga - gb) / np.linalg.norm(gb), np.max(np.abs(ga - gb))))
