#!/usr/bin/env python3
"""Goldens of the float64 ORACLE's full L-BFGS runs at the reference's iteration cap (max_iterations = 100,
/root/reference/pydca/plmdca/plmdca.py:72; exit -997 at lbfgs/lib/lbfgs.cpp:535-539) for BASELINE.json's full-size
configurations D (plmdca protein L=500 N=50k q=21, lambda_h=1 lambda_J=50) and E (plmdca rna L=150 N=200k q=5,
default lambda = 0.2 (L-1)) -- protocol P3 of SURVEY.md 8c4: "same restated optimiser, same cap".

TEST INFRASTRUCTURE: runs oracle/plm_oracle.c (float64 instantiation, carry-over on) on host cores only; no GPU, no
product code.  One evaluation at D takes about half a minute on the GPU box's 256 host cores (hours on 8), so this is run
once per oracle revision through gpurun and its output committed:

    gpurun --timeout 5400 -- 'python tests/golden/make_p3_goldens.py --config D E --out gpurun_out/p3'
    cp gpurun_out/p3/p3_config_*_cap100.npz tests/golden/

What a golden holds (everything the gpu test tests/test_gpu_configs.py::test_P3_full_size_at_reference_cap compares):
status / iterations / evaluations, the per-iteration trace (fx, |x|, |g|, step), FN and FN_APC of the final x, FN_APC
at the checkpoints (iterations 10, 25, 50, 75), the top-L order, every `stride`-th element of the final x, and the
inputs' fingerprints (alignment shape after de-duplication, Meff, |x0|).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

from oracle import mf as oracle_mf   # noqa: E402
from oracle import plm as oracle_plm  # noqa: E402
from tools.gen_msa import SEEDS, dedup, generate  # noqa: E402

# bench.py WORKLOADS / SURVEY 8(d1): L, N, q, lambda_h, lambda_J; seqid 0.8
FULL_SIZE = {"D": (500, 50000, 21, 1.0, 50.0), "E": (150, 200000, 5, 29.8, 29.8),
             "C": (200, 10000, 21, 1.0, 50.0)}
CHECKPOINTS = (10, 25, 50, 75)
X_SAMPLE = 65521          # about this many elements of the final x are kept (every stride-th)


def make(cfg, cap, out_dir, threads, prefix="p3"):
    L, N, q, lh, lJ = FULL_SIZE[cfg]
    t0 = time.time()
    X = dedup(generate(L, N, q, SEEDS[cfg]))
    w = oracle_plm.weights(X, 0.8, np.float64, threads=threads)
    x0 = oracle_plm.init_x(X, w, q)
    t_setup = time.time() - t0
    os.makedirs(out_dir, exist_ok=True)
    prog = os.path.join(out_dir, "%s_config_%s_cap%d.progress.txt" % (prefix, cfg, cap))
    if os.path.exists(prog):
        os.remove(prog)
    os.environ["ORACLE_PROGRESS_FILE"] = prog
    t0 = time.time()
    snaps = [k for k in CHECKPOINTS if k < cap]
    run = oracle_plm.lbfgs(X, w, q, lh, lJ, cap, x0, carry=True, threads=threads, trace_cap=cap, snapshots=snaps)
    t_run = time.time() - t0
    del os.environ["ORACLE_PROGRESS_FILE"]
    fn = oracle_mf.plm_fn(run["x"], L, q, apc_correct=False)
    fn_apc = oracle_mf.plm_fn(run["x"], L, q, apc_correct=True)
    P = run["x"].shape[0]
    stride = max(1, P // X_SAMPLE)
    out = {
        "config": cfg, "L": L, "N_raw": N, "N_unique": X.shape[0], "q": q, "lambda_h": lh, "lambda_J": lJ, "seqid": 0.8,
        "seed": SEEDS[cfg], "cap": cap, "carry": 1,
        "meff": float(np.sum(w)), "x0_norm": float(np.linalg.norm(x0)), "msa_checksum": int(X.astype(np.uint64).sum()),
        "status": run["status"], "iterations": run["iterations"], "evaluations": run["evaluations"], "fx": run["fx"],
        "trace": run["trace"], "fn": fn, "fn_apc": fn_apc,
        "topL_fn": np.argsort(-fn, kind="stable")[:L].astype(np.int32),
        "topL_fn_apc": np.argsort(-fn_apc, kind="stable")[:L].astype(np.int32),
        "x_stride": stride, "x_sample": run["x"][::stride].copy(), "x_norm": float(np.linalg.norm(run["x"])),
        "checkpoints": np.array(sorted(run.get("snapshots", {})), dtype=np.int32),
        "oracle_seconds": t_run, "oracle_threads": threads,
    }
    for k, xk in sorted(run.get("snapshots", {}).items()):
        out["fn_apc_it%d" % k] = oracle_mf.plm_fn(xk, L, q, apc_correct=True)
        out["fn_it%d" % k] = oracle_mf.plm_fn(xk, L, q, apc_correct=False)
    path = os.path.join(out_dir, "%s_config_%s_cap%d.npz" % (prefix, cfg, cap))
    np.savez_compressed(path, **out)
    summary = {k: (v if not isinstance(v, np.ndarray) else "array%r" % (v.shape,)) for k, v in out.items()}
    summary["setup_seconds"] = t_setup
    with open(os.path.join(out_dir, "%s_config_%s_cap%d.summary.json" % (prefix, cfg, cap)), "w") as fh:
        json.dump(summary, fh, indent=1, default=str)
    print("config %s: status %d, %d iterations, %d evaluations, fx %.12g, %.1f s (%.2f s / evaluation, %d threads) -> %s" % (
        cfg, run["status"], run["iterations"], run["evaluations"], run["fx"], t_run, t_run / max(1, run["evaluations"]), threads, path), flush=True)


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", nargs="+", default=["E", "D"], choices=sorted(FULL_SIZE))
    ap.add_argument("--cap", type=int, default=100)
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "p3"))
    ap.add_argument("--threads", type=int, default=os.cpu_count())
    ap.add_argument("--plain", action="store_true",
                    help="the same runs with the oracle built -DORACLE_PLAIN_F64: the reference's LITERAL order of operations in float64 "
                         "(no canonical order, no blocks) -> plain_f64_config_<id>_cap<cap>.npz, what "
                         "tests/test_gpu_configs.py::test_float64_device_vs_reference_order_at_cap bounds the device against")
    a = ap.parse_args()
    oracle_plm.build(ref=False)
    prefix = "p3"
    if a.plain:
        import subprocess
        os.makedirs(a.out, exist_ok=True)
        so = os.path.join(a.out, "liboracle_plain_f64.so")
        subprocess.check_call(["gcc", "-O3", "-fopenmp", "-fno-fast-math", "-ffp-contract=off", "-shared", "-fPIC", "-DORACLE_PLAIN_F64",
                               "-o", so, os.path.join(ROOT, "oracle", "plm_oracle.c"), "-lm"])
        oracle_plm._LIB_PATH = so          # the module loads its library on first use
        oracle_plm._lib = None
        prefix = "plain_f64"
    for cfg in a.config:
        make(cfg, a.cap, a.out, a.threads, prefix)
