#!/usr/bin/env python3
"""How far apart do float64 device runs end that differ ONLY in rounding, after the reference's 100 L-BFGS iterations?
Variants of the same arithmetic: chunked scan with 40 / 80 warm-up steps, the serial chain, and a different slab split of the
scatter kernel (another summation order of the gradient).  The optimisation does not converge, so
rounding-level differences are amplified from iteration to iteration; this measures by how much at a given configuration --
the floor under any device-vs-oracle comparison there.  Writes gpurun_out/sensitivity_<config>_cap<cap>.json.
(Analysis tool; uses only the product library.)

    python tests/analysis/sensitivity.py --config D [--cap 100] [--precision 64]
"""
import argparse
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from pydca_amd import _lib  # noqa: E402
from tools.gen_msa import SEEDS, dedup, generate  # noqa: E402

FULL_SIZE = {"D": (500, 50000, 21, 1.0, 50.0), "E": (150, 200000, 5, 29.8, 29.8), "C": (200, 10000, 21, 1.0, 50.0)}

ap = argparse.ArgumentParser()
ap.add_argument("--config", default="D", choices=sorted(FULL_SIZE))
ap.add_argument("--cap", type=int, default=100)
ap.add_argument("--marks", default="5,10,25,50,75,100")
a = ap.parse_args()
L, N, q, lh, lJ = FULL_SIZE[a.config]
marks = [m for m in (int(v) for v in a.marks.split(",")) if m <= a.cap]
X = dedup(generate(L, N, q, SEEDS[a.config]))
variants = (("chunked40", _lib.CARRY_CHUNKED, 40, None), ("chunked80", _lib.CARRY_CHUNKED, 80, None),
            ("serial", _lib.CARRY_SERIAL, 0, None), ("chunked40_split2", _lib.CARRY_CHUNKED, 40, "2"))
runs = {}
for name, mode, warm, split in variants:
    if split:
        os.environ["DCA_SCATTER_SPLIT"] = split
    else:
        os.environ.pop("DCA_SCATTER_SPLIT", None)
    ctx = _lib.Context(0, _lib.DCA_F64)
    ctx.set_msa(X, q)
    ctx.compute_weights(0.8, _lib.DCA_F64)
    ctx.plm_configure(lh, lJ, mode, 0, warm)
    ctx.plm_init_x()
    ctx.plm_lbfgs_begin(a.cap)
    snaps, done, trace = {}, 0, []
    for m in marks:
        while done < m:
            st = ctx.plm_lbfgs_iterate(1)
            done = st.iterations
            trace.append((st.fx, st.step, st.evaluations))
            if st.finished:
                break
        snaps[done] = ctx.plm_scores(False)
        if st.finished:
            break
    runs[name] = dict(fn=snaps, stats=[st.status, st.iterations, st.evaluations], fx=st.fx, trace=np.array(trace), x=ctx.plm_get_x(np.float64))
    ctx.close()
    print(name, runs[name]["stats"], repr(st.fx), flush=True)
os.environ.pop("DCA_SCATTER_SPLIT", None)
out = {"config": a.config, "cap": a.cap, "runs": {k: {"stats": v["stats"], "fx": v["fx"]} for k, v in runs.items()}, "pairs": {}}
names = list(runs)
for i in range(len(names)):
    for j in range(i + 1, len(names)):
        ra, rb = runs[names[i]], runs[names[j]]
        n = min(len(ra["trace"]), len(rb["trace"]))
        rel_fx = np.abs(ra["trace"][:n, 0] - rb["trace"][:n, 0]) / np.abs(rb["trace"][:n, 0])
        out["pairs"]["%s vs %s" % (names[i], names[j])] = {
            "rel_err_x": float(np.linalg.norm(ra["x"] - rb["x"]) / np.linalg.norm(rb["x"])),
            "max_rel_fn_at": {str(m): float(np.max(np.abs(ra["fn"][m] - rb["fn"][m]) / np.abs(rb["fn"][m]))) for m in ra["fn"] if m in rb["fn"]},
            "rel_fx_at": {str(m): float(rel_fx[m - 1]) for m in marks if m <= n},
            "same_evaluation_counts": bool(np.array_equal(ra["trace"][:n, 2], rb["trace"][:n, 2])),
            "same_topL": bool(list(np.argsort(-ra["fn"][max(ra["fn"])], kind="stable")[:L]) == list(np.argsort(-rb["fn"][max(rb["fn"])], kind="stable")[:L]))}
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump(out, open(os.path.join(ROOT, "gpurun_out", "sensitivity_%s_cap%d.json" % (a.config, a.cap)), "w"), indent=1)
print(json.dumps(out, indent=1))
