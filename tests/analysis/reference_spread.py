#!/usr/bin/env python3
"""The compiled REFERENCE against itself at a full-size configuration: two runs of its own lbfgs() (oracle/_ref, the reference's
C++ sources compiled in place) that differ only in the number of OpenMP threads -- i.e. in the order in which the per-site
partial sums are merged (plmdca_numerics.cpp:570-602) -- to the reference's cap of 100 iterations.  This is the P4 yardstick
of SURVEY 8c4 for that configuration: how far apart the reference's own float32 runs end.  Needs /root/reference (this
container), host cores only.  Writes profiles/r04_reference_spread_<config>.json.
    python tests/analysis/reference_spread.py --config E --threads 8 5"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import mf as omf  # noqa: E402
from oracle import plm as oplm  # noqa: E402
from tools.gen_msa import SEEDS, dedup, generate, write_fasta  # noqa: E402

FULL = {"C": (200, 10000, 21, 1.0, 50.0), "D": (500, 50000, 21, 1.0, 50.0), "E": (150, 200000, 5, 29.8, 29.8)}
ap = argparse.ArgumentParser()
ap.add_argument("--config", default="E")
ap.add_argument("--threads", type=int, nargs="+", default=[8, 5])
ap.add_argument("--cap", type=int, default=100)
a = ap.parse_args()
L, N, q, lh, lJ = FULL[a.config]
X = dedup(generate(L, N, q, SEEDS[a.config]))
path = "/tmp/reference_spread_%s.fa" % a.config
write_fasta(path, X, q)
runs = []
for t in a.threads:
    ref = oplm.Reference(path, 1 if q == 21 else 2, L, q, 0.8, lh, lJ, threads=t)
    t0 = time.time()
    r = ref.lbfgs_run(a.cap)
    dt = time.time() - t0
    ref.close()
    r["fn"] = omf.plm_fn(r["x"].astype(np.float64), L, q, apc_correct=False)
    r["apc"] = omf.plm_fn(r["x"].astype(np.float64), L, q, apc_correct=True)
    r["threads"], r["seconds"] = t, dt
    runs.append(r)
    print("threads %d: status %d, %d iterations, %d evaluations, fx %.9g, %.0f s" % (t, r["status"], r["iterations"], r["evaluations"], r["fx"], dt), flush=True)
out = {"config": a.config, "cap": a.cap, "runs": [{k: r[k] for k in ("threads", "status", "iterations", "evaluations", "fx", "seconds")} for r in runs], "pairs": {}}
gpath = os.path.join(ROOT, "tests", "golden", "p3_config_%s_cap%d.npz" % (a.config, a.cap))
if os.path.exists(gpath):
    g = np.load(gpath)
    runs.append({"threads": "float64 oracle", "fn": g["fn"], "apc": g["fn_apc"], "x": None})


def top(v):
    return np.argsort(-v, kind="stable")[:L]


for i in range(len(runs)):
    for j in range(i + 1, len(runs)):
        ra, rb = runs[i], runs[j]
        tb = top(rb["apc"])
        out["pairs"]["%s vs %s" % (ra["threads"], rb["threads"])] = {
            "max_rel_fn": float(np.max(np.abs(ra["fn"] - rb["fn"]) / np.abs(rb["fn"]))),
            "max_rel_fn_apc_topL": float(np.max(np.abs(ra["apc"][tb] - rb["apc"][tb]) / np.abs(rb["apc"][tb]))),
            "topL_overlap_fn_apc": len(set(tb) & set(top(ra["apc"]))), "topL_same_order": bool(list(tb) == list(top(ra["apc"]))),
            "rel_err_x": None if ra.get("x") is None or rb.get("x") is None else float(np.linalg.norm(ra["x"].astype(np.float64) - rb["x"]) / np.linalg.norm(rb["x"]))}
json.dump(out, open(os.path.join(ROOT, "profiles", "r04_reference_spread_%s.json" % a.config), "w"), indent=1)
print(json.dumps(out, indent=1))
os.unlink(path)
