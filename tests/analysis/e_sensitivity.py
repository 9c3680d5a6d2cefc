#!/usr/bin/env python3
"""Config E (L=150 N=200k q=5), float64, 100 L-BFGS iterations: how far apart do runs end that differ only in rounding?
Device path with the chunked scan (40 and 80 warm-up steps) and the serial chain, each against the others and against
the float64 oracle.  Writes gpurun_out/e_sensitivity.json.   (TEST / analysis tool: uses oracle/.)"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import mf as omf  # noqa: E402
from oracle import plm as oplm  # noqa: E402
from pydca_amd import _lib  # noqa: E402
from tools.gen_msa import SEEDS, dedup, generate  # noqa: E402

L, N, q, lh, lJ, cap = 150, 200000, 5, 29.8, 29.8, int(sys.argv[1]) if len(sys.argv) > 1 else 100
X = dedup(generate(L, N, q, SEEDS["E"]))
runs = {}
w64 = None
for name, mode, warm in (("chunked40", _lib.CARRY_CHUNKED, 40), ("chunked80", _lib.CARRY_CHUNKED, 80), ("serial", _lib.CARRY_SERIAL, 0)):
    ctx = _lib.Context(0, _lib.DCA_F64)
    ctx.set_msa(X, q)
    ctx.compute_weights(0.8, _lib.DCA_F64)
    w64 = ctx.weights()
    ctx.plm_configure(lh, lJ, mode, 0, warm)
    ctx.plm_init_x()
    ctx.plm_lbfgs_begin(cap)
    st = ctx.plm_lbfgs_iterate(cap)
    runs[name] = dict(x=ctx.plm_get_x(np.float64), fn=ctx.plm_scores(False), stats=[st.status, st.iterations, st.evaluations], fx=st.fx)
    ctx.close()
    print(name, runs[name]["stats"], runs[name]["fx"], flush=True)
if os.environ.get("DCA_WITH_ORACLE", "1") == "1":
    ref = oplm.lbfgs(X, w64, q, lh, lJ, cap, oplm.init_x(X, w64, q), carry=True)
    runs["oracle"] = dict(x=ref["x"], fn=omf.plm_fn(ref["x"], L, q, apc_correct=False), stats=[ref["status"], ref["iterations"], ref["evaluations"]], fx=ref["fx"])
out = {"cap": cap, "runs": {k: {"stats": v["stats"], "fx": v["fx"]} for k, v in runs.items()}, "pairs": {}}
names = list(runs)
for a in range(len(names)):
    for b in range(a + 1, len(names)):
        ra, rb = runs[names[a]], runs[names[b]]
        out["pairs"]["%s vs %s" % (names[a], names[b])] = {
            "rel_err_x": float(np.linalg.norm(ra["x"] - rb["x"]) / np.linalg.norm(rb["x"])),
            "max_rel_fn": float(np.max(np.abs(ra["fn"] - rb["fn"]) / np.abs(rb["fn"]))),
            "same_topL": bool(list(np.argsort(-ra["fn"], kind="stable")[:L]) == list(np.argsort(-rb["fn"], kind="stable")[:L]))}
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump(out, open(os.path.join(ROOT, "gpurun_out", "e_sensitivity_cap%d.json" % cap), "w"), indent=1)
print(json.dumps(out, indent=1))
