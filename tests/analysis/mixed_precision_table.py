#!/usr/bin/env python3
"""Which mixed-precision pipelines could keep protocol P3 (FN within 1e-4 of the float64 oracle after the reference's 100
L-BFGS iterations, identical top-L)?  The float64 engine of the ANALYSIS build (make -C pydca_amd/csrc ablate ->
lib/libdca_hip_ablate.so) rounds the OUTPUT of selected stages to float32 -- a lower bound on what computing that stage in
float32 would cost in accuracy (float32 accumulation adds its own error on top) -- and the run is compared with the float64
oracle's run (config C: computed here, 107 evaluations; D / E: the committed goldens).  One row per candidate pipeline,
with the step time that pipeline could reach at best (float32 / float64 kernel times of this box).

    DCA_LIB_PATH=pydca_amd/lib/libdca_hip_ablate.so python tests/analysis/mixed_precision_table.py --config C
Writes gpurun_out/mixed_precision_<config>.json.  (Analysis tool; uses oracle/.)
"""
import argparse
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
if "ablate" not in os.environ.get("DCA_LIB_PATH", ""):
    raise SystemExit("run with DCA_LIB_PATH=pydca_amd/lib/libdca_hip_ablate.so (make -C pydca_amd/csrc ablate)")
from oracle import mf as omf  # noqa: E402
from oracle import plm as oplm  # noqa: E402
from pydca_amd import _lib  # noqa: E402
from tools.gen_msa import SEEDS, dedup, generate  # noqa: E402

FULL = {"C": (200, 10000, 21, 1.0, 50.0), "D": (500, 50000, 21, 1.0, 50.0), "E": (150, 200000, 5, 29.8, 29.8)}
W, S, R, G, GRAD, X, D = 1, 2, 4, 8, 16, 32, 64
# (name, stages whose outputs are float32, which kernels would run in float32 in the real pipeline)
PIPELINES = [
    ("float64 (reference point)", 0, ()),
    ("optimiser vectors float32, evaluation float64", X | D | GRAD, ("lbfgs_vec",)),
    ("float32 logits (W, S), rest float64", W | S, ("plm_logits", "plm_expand")),
    ("float32 S only (float64 sums, float32 storage)", S, ()),
    ("float32 softmax output R, float64 scatter accumulation", R, ("plm_softmax",)),
    ("float32 scatter (R, G), rest float64", R | G, ("plm_softmax", "plm_scatter")),
    ("float32 G only (float64 sums, float32 storage)", G, ()),
    ("float32 evaluation (W, S, R, G, g), float64 optimiser vectors", W | S | R | G | GRAD, ("plm_expand", "plm_logits", "plm_softmax", "plm_scatter", "plm_fold")),
    ("float32 gradient vector only", GRAD, ()),
]


def top(a, L):
    return np.argsort(-a, kind="stable")[:L]


def kernel_times(Xm, q, lh, lJ, precision):
    ctx = _lib.Context(0, precision)
    ctx.set_msa(Xm, q)
    ctx.set_profiling(True)
    ctx.compute_weights(0.8, precision)
    ctx.plm_configure(lh, lJ, _lib.CARRY_CHUNKED)
    ctx.plm_init_x()
    ctx.plm_lbfgs_begin(1000)
    ctx.plm_lbfgs_iterate(2)
    ctx.reset_kernel_times()
    st0 = ctx.plm_lbfgs_iterate(0)
    import time
    t0 = time.perf_counter()
    st = ctx.plm_lbfgs_iterate(8)
    dt = (time.perf_counter() - t0) / max(1, st.iterations - st0.iterations)
    kt = {k: ctx.kernel_time(k)[0] / max(1, st.iterations - st0.iterations) for k in ("plm_expand", "plm_logits", "plm_softmax", "plm_scatter", "plm_fold", "lbfgs_vec")}
    ctx.close()
    return dt * 1e3, kt


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="C", choices=sorted(FULL))
    ap.add_argument("--cap", type=int, default=100)
    a = ap.parse_args()
    L, N, q, lh, lJ = FULL[a.config]
    Xm = dedup(generate(L, N, q, SEEDS[a.config]))
    gpath = os.path.join(ROOT, "tests", "golden", "p3_config_%s_cap%d.npz" % (a.config, a.cap))
    if os.path.exists(gpath):
        gold = np.load(gpath)
        ref = {"fn": gold["fn"], "fn_apc": gold["fn_apc"], "stats": [int(gold["status"]), int(gold["iterations"]), int(gold["evaluations"])]}
    else:
        w = oplm.weights(Xm, 0.8, np.float64)
        r = oplm.lbfgs(Xm, w, q, lh, lJ, a.cap, oplm.init_x(Xm, w, q), carry=True)
        ref = {"fn": omf.plm_fn(r["x"], L, q, apc_correct=False), "fn_apc": omf.plm_fn(r["x"], L, q, apc_correct=True),
               "stats": [r["status"], r["iterations"], r["evaluations"]]}
    os.environ["DCA_ROUND_F32_STAGES"] = "0"
    ms64, k64 = kernel_times(Xm, q, lh, lJ, _lib.DCA_F64)
    ms32, k32 = kernel_times(Xm, q, lh, lJ, _lib.DCA_F32)
    rows = []
    for name, mask, f32_kernels in PIPELINES:
        os.environ["DCA_ROUND_F32_STAGES"] = str(mask)
        ctx = _lib.Context(0, _lib.DCA_F64)
        ctx.set_msa(Xm, q)
        ctx.compute_weights(0.8, _lib.DCA_F64)
        ctx.plm_configure(lh, lJ, _lib.CARRY_CHUNKED)
        ctx.plm_init_x()
        ctx.plm_lbfgs_begin(a.cap)
        st = ctx.plm_lbfgs_iterate(a.cap)
        fn, apc = ctx.plm_scores(False), ctx.plm_scores(True)
        ctx.close()
        t = top(ref["fn_apc"], L)
        best_ms = ms64 - sum(k64[k] - k32[k] for k in f32_kernels)
        row = {"pipeline": name, "rounded_stage_mask": mask, "stats": [st.status, st.iterations, st.evaluations], "oracle_stats": ref["stats"],
               "max_rel_fn": float(np.max(np.abs(fn - ref["fn"]) / np.abs(ref["fn"]))),
               "max_rel_fn_apc_vs_fn": float(np.max(np.abs(apc - ref["fn_apc"]) / np.abs(ref["fn"]))),
               "max_rel_fn_apc_topL": float(np.max(np.abs(apc[t] - ref["fn_apc"][t]) / np.abs(ref["fn_apc"][t]))),
               "topL_same_order": bool(list(top(apc, L)) == list(t)), "topL_overlap": len(set(t) & set(top(apc, L))),
               "best_case_ms_per_iteration": best_ms, "best_case_iterations_per_s": 1e3 / best_ms}
        row["P3"] = bool(row["stats"] == ref["stats"] and row["max_rel_fn"] <= 1e-4 and row["max_rel_fn_apc_vs_fn"] <= 1e-4 and row["topL_same_order"])
        rows.append(row)
        print(json.dumps(row), flush=True)
    out = {"config": a.config, "cap": a.cap, "ms_per_iteration": {"float64": ms64, "float32": ms32}, "kernels_ms": {"float64": k64, "float32": k32}, "rows": rows}
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    json.dump(out, open(os.path.join(ROOT, "gpurun_out", "mixed_precision_%s.json" % a.config), "w"), indent=1)


if __name__ == "__main__":
    main()
