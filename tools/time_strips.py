#!/usr/bin/env python3
"""What ONE rank of the column-strip decomposition (exchange mode 4) does per evaluation at a full-size configuration,
measured on one GPU: `world` ranks run as threads over tests/fake_rccl (host-staged transfers, so the exchange itself is NOT
timed here), rank 0's kernel times are printed.  An emulation for DESIGN.md section 6, not a multi-GPU measurement.
    python tools/time_strips.py --world 8 [--config D]"""
import argparse
import json
import os
import sys
import threading

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pydca_amd import _lib  # noqa: E402
from tools.gen_msa import SEEDS, dedup, generate  # noqa: E402

FULL = {"C": (200, 10000, 21, 1.0, 50.0), "D": (500, 50000, 21, 1.0, 50.0), "E": (150, 200000, 5, 29.8, 29.8)}
FAKE = os.path.join(ROOT, "tests", "fake_rccl", "libfake_rccl.so")
ap = argparse.ArgumentParser()
ap.add_argument("--world", type=int, default=8)
ap.add_argument("--config", default="D")
ap.add_argument("--precision", type=int, default=32)
a = ap.parse_args()
L, N, q, lh, lJ = FULL[a.config]
X = dedup(generate(L, N, q, SEEDS[a.config]))
c0 = _lib.Context(0, a.precision)
c0.set_msa(X, q)
c0.compute_weights(0.8, _lib.DCA_F32)
counts = c0.weight_counts()
c0.close()
uid = _lib.comm_unique_id(FAKE)
out = [None] * a.world
TAGS = ("plm_expand", "plm_logits", "plm_softmax", "plm_scatter", "plm_fold")


def run(rank):
    try:
        c = _lib.Context(0, a.precision)
        c.set_msa(X, q)
        c.set_weight_counts(counts)
        c.comm_init(uid, a.world, rank, FAKE)
        c.plm_configure_strips(lh, lJ)
        c.plm_init_x()
        c.plm_lbfgs_begin(100)
        c.plm_lbfgs_iterate(1)
        c.set_profiling(True)
        c.reset_kernel_times()
        st = c.plm_lbfgs_iterate(3)
        out[rank] = {k: c.kernel_time(k)[0] / max(1, c.kernel_time(k)[1]) for k in TAGS}
        out[rank]["fx"] = st.fx
        c.close()
    except Exception as exc:
        out[rank] = {"error": repr(exc)}


ts = [threading.Thread(target=run, args=(r,)) for r in range(a.world)]
[t.start() for t in ts]
[t.join(timeout=600) for t in ts]
print(json.dumps({"config": a.config, "world": a.world, "precision": a.precision, "rank0": out[0], "rank_last": out[-1]}), flush=True)
os._exit(0)
