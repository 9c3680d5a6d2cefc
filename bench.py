#!/usr/bin/env python3
"""Benchmark of the plmDCA / mfDCA hot path on MI355X (contract: see the task prompt).

    python bench.py [--gpus N --steps K --warmup W] [--workload D|C|E]

A "step" is one L-BFGS iteration of plmDCA (line search included: ~1.0-1.2 objective +
gradient evaluations) on the synthetic protein alignment L=500, N=50k, q=21 (config D of
BASELINE.json / SURVEY.md section 8), lambda_h=1, lambda_J=50, seqid 0.8, resident in HBM.
N>1: sequences are sharded over the ranks, every evaluation ends with one RCCL all-reduce
of the gradient (strong scaling: total work fixed).  Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

WORKLOADS = {
    # name: (L, N, q, seed, lambda_h, lambda_J)
    "D": (500, 50000, 21, 12346, 1.0, 50.0),
    "C": (200, 10000, 21, 12345, 1.0, 50.0),
    "E": (150, 200000, 5, 12347, 29.8, 29.8),
}
HBM_PEAK_GBS = 8000.0             # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
LDS_PEAK_GBS = 256 * 256 * 2.4    # 256 B/clk/CU x 256 CUs x 2.4 GHz = 157 TB/s (MI355X_MICROARCH.md, LDS)


def kernel_sources_fingerprint():
    """sha256 over the sources of the plmDCA kernels (the generator and what it generates included): profiles/traffic.json
    is stamped with it, so a PMC measurement of other kernels than the ones in this tree is recognised as stale (the GPU box
    has no .git to ask)."""
    import hashlib
    h = hashlib.sha256()
    for rel in ("pydca_amd/csrc/plm_engine.hip", "pydca_amd/csrc/logits_gather_asm.inc", "pydca_amd/csrc/scatter_gather_asm.inc",
                "tools/gen_plm_asm.py"):
        with open(os.path.join(ROOT, rel), "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()


def _time_reference_eval(oplm, Xs, q, lh, lJ, threads):
    """Seconds of ONE objective + gradient evaluation of the reference's own C++ (oracle/_ref; kind "reference") or,
    when that build is absent, of the C restatement (kind "port") on the sequences Xs with `threads` OpenMP threads."""
    from tools.gen_msa import write_fasta
    L = Xs.shape[1]
    if oplm.have_reference():
        try:
            path = "/tmp/bench_cpu_sample_%d.fa" % os.getpid()
            write_fasta(path, Xs, q)
            ref = oplm.Reference(path, 1 if q == 21 else 2, L, q, 0.8, lh, lJ, threads=threads)
            x = ref.init_x()
            ref.gradient(x)                       # warm
            dt = 1e30
            for _ in range(2):                    # best of two: thread start-up and page faults show in the first call
                t0 = time.perf_counter()
                ref.gradient(x)
                dt = min(dt, time.perf_counter() - t0)
            ref.close()
            os.unlink(path)
            return dt, "reference"
        except Exception as exc:                   # pragma: no cover
            print("reference baseline unavailable: %r" % (exc,), file=sys.stderr)
    w = oplm.weights(Xs, 0.8, np.float32, threads=threads)
    x = oplm.init_x(Xs, w, q)
    oplm.gradient(Xs, w, q, lh, lJ, x, carry=True, threads=threads)
    t0 = time.perf_counter()
    oplm.gradient(Xs, w, q, lh, lJ, x, carry=True, threads=threads)
    return time.perf_counter() - t0, "port"


def _fit_full_eval(oplm, X, q, lh, lJ, threads, sizes):
    """Evaluation time at two sample sizes -> t(n) = a + b n (the reference's evaluation has N-independent parts: the
    regulariser loop over all parameters, plmdca_numerics.cpp:463-486, and one critical-section merge of L q^2 floats per
    site, :570-602) -> extrapolated to the full alignment."""
    ts, kind = [], None
    for n in sizes:
        dt, kind = _time_reference_eval(oplm, np.ascontiguousarray(X[:n]), q, lh, lJ, threads)
        ts.append(dt)
    n1, n2 = sizes
    b = (ts[1] - ts[0]) / float(n2 - n1)
    a = ts[0] - b * n1
    if b <= 0 or a < 0:                           # noisy box: fall back to proportional scaling of the larger sample
        a, b = 0.0, ts[1] / float(n2)
    return a + b * X.shape[0], a, b, ts, kind


def cpu_baseline(X, q, lh, lJ, evals_per_iter, sample_rows):
    """plmDCA: the reference's C++/OpenMP gradient timed on this host at two sample sizes (all cores) and at two smaller
    ones on ONE thread, fitted a + b N and extrapolated to N.  mfDCA: the numpy / LAPACK float64 restatement of the
    reference's chain (msa_numerics.py:13-342; np.linalg.inv is the call of :340) -- the inverse at full size, the
    N-dependent stages on samples and scaled."""
    from oracle import mf as omf
    from oracle import plm as oplm
    cores = os.cpu_count() or 1
    N, L = X.shape
    threads = max(1, min(cores, L))
    n2 = min(N, sample_rows)
    n1 = max(1, n2 // 2)
    full, a, b, ts, kind = _fit_full_eval(oplm, X, q, lh, lJ, threads, (n1, n2))
    m2 = max(2, min(N, sample_rows // 16))
    m1 = max(1, m2 // 2)
    full1, a1, b1, ts1, _ = _fit_full_eval(oplm, X, q, lh, lJ, 1, (m1, m2))
    # two samples of a noisy host do not pin a line: besides the fit, the proportional scaling of each sample alone
    # brackets the figure (the fit's intercept moved the extrapolation 42.6 -> 55.2 s between two runs of round 3)
    prop = sorted((ts[0] * N / float(n1), ts[1] * N / float(n2), full))
    out = {"value": 1.0 / (full * evals_per_iter), "unit": "L-BFGS iterations/s", "cores": threads, "kind": kind,
           "value_range": [1.0 / (prop[-1] * evals_per_iter), 1.0 / (prop[0] * evals_per_iter)],
           "seconds_per_evaluation_range": [prop[0], prop[-1]],
           "sample": "one objective+gradient evaluation of the reference C++/OpenMP on the first %d and %d of %d sequences (L=%d, q=%d): "
                     "%.2f s and %.2f s on %d threads; fitted t = %.2f s + %.3f ms x N -> %.1f s at N, divided by the %.2f "
                     "evaluations/iteration of the GPU run" % (n1, n2, N, L, q, ts[0], ts[1], threads, a, b * 1e3, full, evals_per_iter),
           "seconds_per_evaluation": full, "fit": {"a_s": a, "b_s_per_sequence": b, "samples": [n1, n2], "seconds": ts},
           "one_thread": {"value": 1.0 / (full1 * evals_per_iter), "unit": "L-BFGS iterations/s", "cores": 1,
                          "seconds_per_evaluation": full1, "fit": {"a_s": a1, "b_s_per_sequence": b1, "samples": [m1, m2], "seconds": ts1}}}
    # ---- mfDCA on the host: float64 numpy / LAPACK restatement (oracle/mf.py), kind "port"
    try:
        n = L * (q - 1)
        rows = min(N, 8000)
        t0 = time.perf_counter()
        oplm.weights(np.ascontiguousarray(X[:rows]), 0.8, np.float64, threads=threads)    # C/OpenMP restatement of msa_numerics.py:13-50
        t_w = (time.perf_counter() - t0) * (N / float(rows)) ** 2                          # identity counts are N^2 L
        ns = min(N, 2000)
        X1 = X[:ns].astype(np.int32) + 1
        wS = np.ones(ns)
        t0 = time.perf_counter()
        fi = omf.get_reg_single_site_freqs(omf.compute_single_site_freqs(X1, q, wS), L, q, 0.5)
        fij = omf.get_reg_pair_site_freqs(omf.compute_pair_site_freqs(X1, q, wS), L, q, 0.5)
        t_counts = (time.perf_counter() - t0) * N / float(ns)
        t0 = time.perf_counter()
        Cm = omf.construct_corr_mat(fi, fij, L, q)
        t_corr = time.perf_counter() - t0
        Cm[np.diag_indices(n)] += 1.0                 # the sample's matrix may be near singular; timing only
        t0 = time.perf_counter()
        J = omf.compute_couplings(Cm)
        t_inv = time.perf_counter() - t0
        t0 = time.perf_counter()
        omf.apc(omf.frobenius_from_blocks(omf.mf_blocks(J, L, q)), L)
        t_sc = time.perf_counter() - t0
        total = t_w + t_counts + t_corr + t_inv + t_sc
        out["mfdca"] = {"value": (L * (L - 1) // 2) / total, "unit": "residue pairs/s", "cores": cores, "kind": "port",
                        "seconds": total, "stages_s": {"weights": t_w, "counts": t_counts, "corr": t_corr, "inverse": t_inv, "scores": t_sc},
                        "sample": "numpy/LAPACK float64 restatement: np.linalg.inv at the full n=%d (%.2f s), weights on %d sequences (scaled quadratically) and pair "
                                  "counts on %d sequences (scaled linearly) to N=%d" % (n, t_inv, rows, ns, N)}
    except Exception as exc:   # pragma: no cover
        out["mfdca"] = {"error": repr(exc)}
    return out


def gather_issue_model(q, precision):
    """What the two gather kernels ISSUE per (sequence, site, 512-byte strip) unit: packed adds and LDS row bytes.  q = 21 and
    the float64 mode: one indexed add per unit, the per-site blocks of tools/gen_plm_asm.py.  q = 5 in float32 (round 5) walks
    site PAIRS on the 25-state alphabet: the logits kernel (25 + 80) adds and 10 row reads per 160 units, the scatter
    kernel one add per two units and one row read per four."""
    from tools.gen_plm_asm import LOGITS_CFG
    if q == 5 and precision == 32 and os.environ.get("DCA_PLM_PAIRS") != "0":
        nseq = LOGITS_CFG[25][1]
        return {"plm_logits": {"adds": (25.0 + nseq) / (2.0 * nseq), "lds_bytes": 10 * 512.0 / (2.0 * nseq)},
                "plm_scatter": {"adds": 0.5, "lds_bytes": 512.0 / 4}, "formulation": "site-pair alphabet (25 combined states)"}
    return {"plm_logits": {"adds": 1.0, "lds_bytes": q * 512.0 / LOGITS_CFG[q][1]}, "plm_scatter": {"adds": 1.0, "lds_bytes": 512.0 / 2},
            "formulation": "per-site blocks"}


def rna_workload_block(device, steps=10, warmup=3):
    """Config E (plmdca rna, L=150 N=200k q=5: BASELINE.json's "bandwidth-bound regime" case) on this GPU, beside the headline
    configuration: iterations/s and what its two gather kernels reach of the same roofs.  q = 5 runs other kernel
    variants than q = 21 (logits: 16 waves x 48 sequences, scatter: strips dealt to the XCDs by (strip, split))."""
    from pydca_amd import _lib
    from tools.gen_msa import dedup, generate
    L, N, q, seed, lh, lJ = WORKLOADS["E"]
    X = dedup(generate(L, N, q, seed))
    N = X.shape[0]
    ctx = _lib.Context(device, _lib.DCA_F32)
    ctx.set_msa(X, q)
    ctx.set_profiling(True)
    ctx.compute_weights(0.8, _lib.DCA_F32)
    t_w, _ = ctx.kernel_time("weights")
    ctx.plm_configure(lh, lJ, _lib.CARRY_CHUNKED)
    ctx.plm_init_x()
    ctx.plm_lbfgs_begin(steps + warmup + 1000)
    st = ctx.plm_lbfgs_iterate(warmup)
    it0, ev0 = st.iterations, st.evaluations
    ctx.set_profiling(False)                 # the timed iterations without event records (main(): 4 % of this workload's step)
    t0 = time.perf_counter()
    st = ctx.plm_lbfgs_iterate(steps)
    dt = time.perf_counter() - t0
    done = st.iterations - it0
    ctx.set_profiling(True)
    ctx.reset_kernel_times()
    ctx.plm_lbfgs_iterate(steps)             # the stages clocked over as many iterations again
    kt = {tag: ctx.kernel_time(tag) for tag in ("plm_expand", "plm_logits", "plm_softmax", "plm_scatter", "plm_fold", "lbfgs_vec")}
    Lq, esz = L * q, 4
    units = N * L * (Lq * esz / 512.0)
    out = {"workload": "plmdca rna, synthetic MSA L=%d N=%d q=%d, lambda=%g" % (L, N, q, lh), "value": done / dt, "unit": "iterations/s",
           "ms_per_step": dt / max(done, 1) * 1e3, "evaluations_per_iteration": (st.evaluations - ev0) / max(done, 1),
           "weights_kernel_ms": t_w, "kernels": {k: {"avg_ms": v[0] / max(v[1], 1), "launches": v[1]} for k, v in kt.items()}, "roofline": {}}
    alg = {"plm_logits": Lq * Lq * esz + N * L + N * Lq * esz, "plm_scatter": N * Lq * esz + Lq * Lq * esz + N * L * 2,
           "plm_softmax": 2 * N * Lq * esz + N * L + 4 * N}
    model = gather_issue_model(q, 32)
    out["gather_formulation"] = model["formulation"]
    lds = {k: units * model[k]["lds_bytes"] for k in ("plm_logits", "plm_scatter")}
    for k in ("plm_logits", "plm_scatter", "plm_softmax"):
        avg_s = kt[k][0] / max(kt[k][1], 1) / 1e3
        r = {"avg_kernel_ms": avg_s * 1e3, "hbm": {"achieved": alg[k] / avg_s / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": alg[k] / avg_s / 1e9 / HBM_PEAK_GBS}}
        if k in lds:
            issued = N * L * Lq * model[k]["adds"]          # packed-add lanes the kernel really issues (the pair alphabet needs fewer than one per unit)
            r["valu"] = {"achieved": issued / avg_s / 1e12, "peak": 78.6, "unit": "Tadd/s", "frac": issued / avg_s / 1e12 / 78.6,
                         "issued_adds_per_sequence_site_column": model[k]["adds"],
                         "sequence_site_column_sums_per_s": N * L * Lq / avg_s / 1e12}
            r["onchip"] = {"achieved": lds[k] / avg_s / 1e9, "peak": LDS_PEAK_GBS, "unit": "GB/s", "frac": lds[k] / avg_s / 1e9 / LDS_PEAK_GBS}
        out["roofline"][k] = r
    ctx.close()
    return out


def precision_modes(_lib, X, q, lh, lJ, device, workload, f32_value, f32_ms, steps=10, warmup=2):
    """The float64 mode (PlmDCA(..., precision=64): float64 kernels, vectors and double-double reductions) timed like the
    headline -- `warmup` untimed and `steps` timed L-BFGS iterations on the resident alignment -- next to the float32
    headline, each with the parity class measured for it at this configuration (profiles/p3_config_<id>_cap100.json)."""
    ctx = _lib.Context(device, _lib.DCA_F64)
    ctx.set_msa(X, q)
    ctx.compute_weights(0.8, _lib.DCA_F64)
    ctx.plm_configure(lh, lJ, _lib.CARRY_CHUNKED)
    ctx.plm_init_x()
    ctx.plm_lbfgs_begin(steps + warmup + 1000)
    st = ctx.plm_lbfgs_iterate(warmup)
    it0 = st.iterations
    t0 = time.perf_counter()
    st = ctx.plm_lbfgs_iterate(steps)
    dt = time.perf_counter() - t0
    done = st.iterations - it0
    ctx.close()
    modes = {"f32": {"iterations_per_s": f32_value, "ms_per_step": f32_ms, "parity_class": "P4",
                     "what": "float32 storage and arithmetic as in the reference (lbfgs.h:50-62), double reductions; the bench headline"},
             "f64": {"iterations_per_s": done / dt, "ms_per_step": dt / max(done, 1) * 1e3, "steps": done, "warmup": warmup, "parity_class": "P3",
                     "what": "float64 kernels and vectors, double-double reductions: PlmDCA(..., precision=64), `plmdca ... --precision 64`"}}
    rpath = os.path.join(ROOT, "profiles", "p3_config_%s_cap100.json" % workload)
    if os.path.exists(rpath):
        try:
            r = json.load(open(rpath))
            modes["parity_report"] = os.path.relpath(rpath, ROOT)
            # the report is a committed file of an earlier run of tests/test_gpu_configs.py (the driver's GPU suite asserts the
            # same bounds live): stale when the kernel sources it was measured on are not this tree's
            modes["parity_report_is_stale"] = r.get("kernel_sources_sha256") != kernel_sources_fingerprint()
            modes["f64"]["vs_float64_oracle_at_cap_100"] = {k: r.get(k) for k in (
                "gpu", "oracle", "first_divergence", "max_rel_fn", "max_rel_fn_apc_vs_fn", "max_rel_fn_apc_topL_self", "topL_same_fn", "topL_same_fn_apc")}
            modes["f64"]["meets_north_star_tolerance"] = bool(
                r.get("first_divergence") is None and r.get("max_rel_fn", 1) <= 1e-4 and r.get("max_rel_fn_apc_vs_fn", 1) <= 1e-4
                and r.get("topL_same_fn") and r.get("topL_same_fn_apc"))
            f = r.get("float32", {})
            modes["f32"]["vs_float64_oracle_at_cap_100"] = {k: f.get(k) for k in (
                "status", "max_rel_fn", "max_rel_fn_apc_vs_fn", "max_rel_fn_apc_topL_self", "topL_same_fn_apc", "topL_overlap_fn_apc")}
            modes["f32"]["meets_north_star_tolerance"] = bool(f.get("max_rel_fn", 1) <= 1e-4 and f.get("topL_same_fn_apc"))
            spath = os.path.join(ROOT, "profiles", "r04_reference_spread_%s.json" % workload)
            if os.path.exists(spath):      # the reference's own run-to-run spread at this configuration (two thread counts), where it could be afforded
                sp = json.load(open(spath))
                k = [v for name, v in sp["pairs"].items() if "oracle" not in name]
                if k:
                    modes["f32"]["reference_vs_reference_at_cap_100"] = dict(k[0], file=os.path.relpath(spath, ROOT))
        except Exception as exc:   # pragma: no cover
            modes["parity_report_error"] = repr(exc)
    else:
        modes["parity_report"] = None
    return modes


def end_to_end(X, L, q, lh, lJ, device):
    """SURVEY 8 d2's end-to-end figures, outside the headline's timed region: what `plmdca compute_fn <bio> <file> --apc`
    (plmdca_main.py:136-256; max_iterations = the reference's default 100) and `mfdca compute_fn <bio> <file> --apc` do, from
    the FASTA FILE to the ranked list on the host, through the same classes the command lines use, with the split
    reader / setup / compute / ranking.  The file is the synthetic alignment written to /tmp (page-cache hot, as after
    any first read); the plmDCA run is also the bench's full 100-iteration figure with its evaluations per iteration."""
    from pydca_amd.meanfield_dca.meanfield_dca import MeanFieldDCA
    from pydca_amd.plmdca.plmdca import PlmDCA
    from tools.gen_msa import write_fasta
    bio = "protein" if q == 21 else "rna"
    path = "/tmp/bench_e2e_%d.fa" % os.getpid()
    write_fasta(path, X, q)
    npairs = L * (L - 1) // 2
    res = {"file_bytes": os.path.getsize(path), "pairs": npairs}
    try:
        best = None
        for rep in range(2):                      # the second pass is reported (first-use effects of the classes in the first)
            t0 = time.perf_counter()
            inst = PlmDCA(path, bio, seqid=0.8, lambda_h=lh, lambda_J=lJ, max_iterations=100, device=device)
            t1 = time.perf_counter()
            ranked = inst.compute_sorted_FN_APC()
            t2 = time.perf_counter()
            ls = inst.last_status
            best = {"seconds": t2 - t0, "plm_pairs_per_s": npairs / (t2 - t0), "iterations": ls["iterations"], "evaluations": ls["evaluations"],
                    "evaluations_per_iteration": ls["evaluations"] / max(ls["iterations"], 1), "lbfgs_status": ls["status"],
                    "iterations_per_s_full_run": ls["iterations"] / ls["stages_s"]["optimise"],
                    "stages_s": dict(ls["stages_s"], constructor_file_scan=t1 - t0), "top_pair": [int(v) for v in ranked[0][0]],
                    "unique_sequences": ls["unique_sequences"]}
            del inst
        res["plmdca_compute_fn"] = best
        # three passes, the median one reported (the first has the first-use effects of the classes; the second list of an L builds
        # the (i, j) cache of pydca_amd/_ranking.py, later ones use it); the list of the pass before is freed OUTSIDE the timed
        # region -- a command line builds one list, it never frees a quarter of a million tuples while it builds the next
        passes = []
        for rep in range(3):
            ranked = m = None
            t0 = time.perf_counter()
            m = MeanFieldDCA(path, bio, pseudocount=0.5, seqid=0.8, device=device)
            ranked = m.compute_sorted_FN_APC()
            t1 = time.perf_counter()
            passes.append({"seconds": t1 - t0, "mf_pairs_per_s": npairs / (t1 - t0), "stages_s": dict(m.last_timings),
                           "top_pair": [int(v) for v in ranked[0][0]], "unique_sequences": m.num_sequences})
        ranked = m = None
        best = dict(sorted(passes, key=lambda r: r["seconds"])[1], statistic="median of three passes", samples_s=[r["seconds"] for r in passes])
        res["mfdca_compute_fn"] = best
    finally:
        os.unlink(path)
    return res


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="D", choices=sorted(WORKLOADS))
    ap.add_argument("--precision", type=int, default=32, choices=(32, 64))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-mfdca", action="store_true")
    ap.add_argument("--no-e2e", action="store_true")
    ap.add_argument("--no-rna", action="store_true")
    ap.add_argument("--no-modes", action="store_true")
    ap.add_argument("--cpu-sample", type=int, default=0)
    ap.add_argument("--cpu-full", action="store_true",
                    help="also time ONE full evaluation of the reference C++/OpenMP on all sequences (about a minute at config D) and report it beside the fit")
    args = ap.parse_args()

    # DCA_BENCH_SELFTEST: several ranks on ONE GPU (tests): "1" = gloo + torch.distributed hooks; "native" = the library's
    # own communicators over the stand-in librccl named by DCA_RCCL_PATH (tests/fake_rccl/libfake_rccl_mp.so), i.e. the
    # very code path a multi-GPU node runs -- scheme timing, selection, column strips -- with host-staged transfers
    selftest = os.environ.get("DCA_BENCH_SELFTEST") in ("1", "native")
    selftest_native = os.environ.get("DCA_BENCH_SELFTEST") == "native"
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # plain `python bench.py --gpus N`: start the N ranks ourselves (one process per GPU under torch.distributed.run,
        # the launcher the driver uses), instead of silently measuring one GPU
        import socket
        import subprocess
        import torch
        have = torch.cuda.device_count()
        if have < args.gpus and not selftest:
            raise SystemExit("bench.py --gpus %d: only %d GPU(s) visible (torch.cuda.device_count())" % (args.gpus, have))
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
               "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        raise SystemExit(subprocess.call(cmd, env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))))

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE=%d (launch with --nproc-per-node equal to --gpus)" % (args.gpus, world))

    import torch
    from pydca_amd import _lib, parallel
    from tools.gen_msa import dedup, generate

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: torch.cuda.is_available() is False (no CPU fallback)")
    # self-test of the multi-process path on a box with ONE GPU: DCA_BENCH_SELFTEST=1 puts every rank on
    # device 0 and uses gloo (RCCL refuses two ranks on one device); numbers from it are meaningless
    if selftest:
        local_rank = 0
    elif torch.cuda.device_count() < world:
        raise SystemExit("bench.py: %d ranks but only %d GPU(s) visible" % (world, torch.cuda.device_count()))
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if selftest:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    L, N, q, seed, lh, lJ = WORKLOADS[args.workload]
    t0 = time.perf_counter()
    X = dedup(generate(L, N, q, seed))
    N = X.shape[0]
    t_gen = time.perf_counter() - t0

    # ---- setup (untimed, reported): weights on the full alignment, shard, lists, x0
    t0 = time.perf_counter()
    full = _lib.Context(local_rank, _lib.DCA_F64 if args.precision == 64 else _lib.DCA_F32)     # --precision 64: the float64 checking mode
    full.set_msa(X, q)
    full.set_profiling(True)
    # exchange scheme of a multi-GPU run: the library's own RCCL communicator on its stream (default), or the
    # torch.distributed hooks (DCA_BENCH_TORCH_COMM=1; always in the gloo self-test, where RCCL cannot run)
    native = world > 1 and (not selftest or selftest_native) and os.environ.get("DCA_BENCH_TORCH_COMM") != "1"
    tdev = "cpu" if selftest else "cuda"           # tensors of the torch.distributed calls (gloo in the self-tests)
    if world == 1:
        full.compute_weights(0.8, _lib.DCA_F32)

    def native_comm_up(context):
        """Gives `context` the library's own communicator; every rank must agree that it came up, otherwise all of
        them fall back to the torch.distributed hooks."""
        ok = 1
        try:
            parallel.init_native_comm(context, _lib, rank, world, dist)
        except Exception as exc:               # pragma: no cover (needs a multi-GPU node)
            print("rank %d: native communicator unavailable (%r): torch.distributed hooks instead" % (rank, exc), file=sys.stderr)
            ok = 0
        flag = torch.tensor([ok], dtype=torch.int32, device=tdev)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        return bool(flag.item())

    if native:
        native = native_comm_up(full)
    if world == 1:
        pass
    elif native:
        # every rank counts 1/world of the identity comparisons, ONE all-reduce of the N integer counts
        full.compute_weights_sharded(0.8, _lib.DCA_F32)
    else:
        part = torch.from_numpy(full.weights_partial_counts(0.8, _lib.DCA_F32, rank, world).astype(np.int64))
        if not selftest:
            part = part.cuda()
        dist.all_reduce(part, op=dist.ReduceOp.SUM)
        full.set_weight_counts(part.cpu().numpy().astype(np.uint32))
    counts = full.weight_counts()
    w32 = (np.float32(1.0) / counts.astype(np.float32)).astype(np.float32)
    t_weights_ms, _ = full.kernel_time("weights")
    hook = None
    comm_selection = None
    scheme = "1 GPU"
    if world == 1:
        ctx = full
        ctx.plm_configure(lh, lJ, _lib.CARRY_CHUNKED)
        ctx.plm_init_x()
    else:
        full.close()
        ctx = parallel.make_sharded_plm_context(_lib, X, q, w32.astype(np.float64), lh, lJ, rank, world, local_rank,
                                                precision=args.precision)
        ctx.plm_set_x(parallel.initial_x(X, w32, q, np.float32))
        # sequences AND optimiser vectors sharded: reduce-scatter(g) + all-gather(x) per evaluation
        # (DCA_BENCH_ALLREDUCE=1 selects the plain all-reduce of g with replicated vectors instead)
        # Small parameter vectors (config E: 1.1 MB) are latency-bound: one all-reduce per evaluation beats
        # four collectives per iteration, so vector sharding is used from 16 MB of parameters on.
        small = ctx.num_params() * (4 if args.precision == 32 else 8) < (16 << 20)
        allreduce = os.environ.get("DCA_BENCH_ALLREDUCE") == "1" or (small and os.environ.get("DCA_BENCH_VECTORS") != "1")
        if native:
            native = native_comm_up(ctx)
        comm_selection = None
        if native:
            # Which exchange scheme the wires like cannot be known before it runs on them: time the
            # three native schemes once on this node -- all-reduce of g (1), RCCL reduce-scatter + all-gather with sharded
            # optimiser vectors (2), the same as a direct exchange of grouped send / recv (3) -- and keep the fastest.
            # DCA_BENCH_SCHEME=1|2|3 forces one; DCA_BENCH_ALLREDUCE / DCA_BENCH_VECTORS keep their meaning.
            forced = os.environ.get("DCA_BENCH_SCHEME")
            x0 = parallel.initial_x(X, w32, q, np.float32)
            strip_ctx = None

            def strips_up():
                """mode 4: a context with the WHOLE alignment whose rank takes the columns of its share of the sites"""
                c = _lib.Context(local_rank, _lib.DCA_F64 if args.precision == 64 else _lib.DCA_F32)
                c.set_msa(X, q)
                c.set_weight_counts(counts)
                if not native_comm_up(c):
                    c.close()
                    return None
                c.plm_configure_strips(lh, lJ, _lib.CARRY_CHUNKED)
                return c

            def time_mode(c, mode):
                """three L-BFGS iterations (after one untimed) under an exchange scheme, max over the ranks, ms per iteration"""
                ok = 1
                try:
                    if mode != 4:
                        c.plm_set_native_comm(mode)
                    c.plm_set_x(x0)
                    c.plm_lbfgs_begin(4)                                     # 1 + 3 iterations, then the cap ends the run: the scheme can only change between runs
                    c.plm_lbfgs_iterate(1)                                   # warm: RCCL sets its channels up on first use
                except Exception as exc:                                     # pragma: no cover (needs a multi-GPU node)
                    print("rank %d: exchange mode %d unavailable (%r)" % (rank, mode, exc), file=sys.stderr)
                    ok = 0
                flag = torch.tensor([ok], dtype=torch.int32, device=tdev)
                dist.all_reduce(flag, op=dist.ReduceOp.MIN)                  # a mode is timed only if it came up on every rank
                if not bool(flag.item()):
                    return None
                barrier()
                t1 = time.perf_counter()
                c.plm_lbfgs_iterate(3)
                barrier()
                tm = torch.tensor([(time.perf_counter() - t1) / 3.0], dtype=torch.float64, device=tdev)
                dist.all_reduce(tm, op=dist.ReduceOp.MAX)
                return float(tm.item()) * 1e3

            if forced:
                chosen, timings = int(forced), {}
            elif os.environ.get("DCA_BENCH_ALLREDUCE") == "1":
                chosen, timings = 1, {}
            else:
                # Which scheme the wires like cannot be known before it runs on them: three iterations of
                # each are timed once on this node -- all-reduce of g (1), RCCL reduce-scatter + all-gather with sharded
                # optimiser vectors (2), the same as a direct exchange of grouped send / recv (3), the column-strip
                # decomposition (4: a fifth of the bytes, all-to-all) -- and the fastest runs the benchmark.
                timings = {}
                for mode in (1, 2, 3):
                    t = time_mode(ctx, mode)
                    if t is not None:
                        timings[mode] = t
                strip_ctx = strips_up()
                if strip_ctx is not None:
                    t = time_mode(strip_ctx, 4)
                    if t is not None:
                        timings[4] = t
                chosen = 2 if not timings else min(timings, key=lambda m: timings[m])
            if chosen == 4:
                if strip_ctx is None:
                    strip_ctx = strips_up()
                if strip_ctx is None:
                    raise SystemExit("bench.py: exchange mode 4 asked for but the communicator did not come up")
                ctx.close()
                ctx = strip_ctx
            else:
                if strip_ctx is not None:
                    strip_ctx.close()
                ctx.plm_set_native_comm(chosen)
            ctx.plm_set_x(x0)
            allreduce = chosen == 1
            comm_selection = {"ms_per_iteration": timings, "chosen_mode": chosen, "rccl_ranks": ctx.comm_info()[0],
                              "modes": {"1": "sequences sharded, all-reduce(g)", "2": "sequences sharded, RCCL reduce-scatter(g) + all-gather(x), sharded vectors",
                                        "3": "sequences sharded, direct exchange (grouped send / recv + rank-ordered local sum), sharded vectors",
                                        "4": "column strips: every rank all sequences x the columns of its sites; couplings up / gradient-table rows down by grouped send / recv"}}
        elif allreduce:
            hook = parallel.TorchAllReduceHook(local_rank)
            ctx.plm_set_reduce_hook(hook)
        else:
            hook = parallel.TorchVectorComm(local_rank, rank, world)
            ctx.plm_set_vector_sharding(rank, world, hook)
        scheme = ("sequences sharded x%d, all-reduce(g) over RCCL" % world if allreduce else
                  "sequences sharded x%d, reduce-scatter(g) + all-gather(x) over RCCL, L-BFGS vectors sharded x%d" % (world, world))
        if comm_selection is not None and comm_selection["chosen_mode"] == 4:
            scheme = "column strips x%d (sites sharded, all sequences on every rank), point-to-point exchange of couplings and gradient-table rows over RCCL" % world
        scheme += ", native collectives on the library's stream" if native else ", torch.distributed hooks"
        if comm_selection is not None:
            scheme += "; exchange mode %d of 4 picked by a start-up timing, RCCL communicator of %d ranks" % (comm_selection["chosen_mode"], world)
    t_setup = time.perf_counter() - t0

    # ---- warm-up iterations, then exactly K timed iterations
    total_cap = args.warmup + args.steps
    ctx.plm_lbfgs_begin(total_cap + 1000)
    st = ctx.plm_lbfgs_iterate(args.warmup) if args.warmup > 0 else None
    it0 = st.iterations if st else 0
    ev0 = st.evaluations if st else 1
    # The timed region carries HIP events around ONE stage only -- the roofline's kernel (the gather stage that dominates: scatter
    # at q = 21, logits on the site-pair alphabet of q = 5).  An event record costs the stream ~5 us; two per stage and seven stages
    # per iteration were 5 % of config C's step and 4 % of E's (0.4 % of D's) -- the instrument, not the path.  The other stages are
    # clocked over the same number of iterations right after the timed region (`kernel_clock_pass`).
    roof_stage = "plm_scatter" if q == 21 else "plm_logits"
    ctx.set_profiling_only(roof_stage)
    ctx.reset_kernel_times()
    barrier()
    t0 = time.perf_counter()
    st = ctx.plm_lbfgs_iterate(args.steps)
    barrier()
    dt = time.perf_counter() - t0
    per_rank_ms = None
    if dist is not None:
        tall = [torch.zeros(1, dtype=torch.float64, device=tdev) for _ in range(world)]
        dist.all_gather(tall, torch.tensor([dt], dtype=torch.float64, device=tdev))
        per_rank_ms = [float(t.item()) / max(args.steps, 1) * 1e3 for t in tall]       # every rank's own clock around the same K iterations
        dt = max(float(t.item()) for t in tall)
    steps_done = st.iterations - it0
    evals = st.evaluations - ev0
    roof_time = ctx.kernel_time(roof_stage)               # over the timed region
    # the same number of iterations again with every stage clocked (not part of `value`); the optimiser simply goes on
    ctx.set_profiling(True)
    ctx.reset_kernel_times()
    barrier()
    tk = time.perf_counter()
    st2 = ctx.plm_lbfgs_iterate(args.steps)
    barrier()
    clock_pass = {"steps": st2.iterations - st.iterations, "ms_per_step": (time.perf_counter() - tk) / max(st2.iterations - st.iterations, 1) * 1e3,
                  "what": "every stage bracketed with HIP events (14 event records per iteration); the timed region brackets only %s" % roof_stage}
    ktimes = {tag: ctx.kernel_time(tag) for tag in ("plm_expand", "plm_logits", "plm_softmax", "plm_scatter", "plm_fold", "lbfgs_vec")}
    steps_clocked = max(st2.iterations - st.iterations, 1)
    if roof_time[1] > 0:
        ktimes[roof_stage] = roof_time
    ctx.set_profiling(False)

    # The N = 1 point of the same run: rank 0 times the same K iterations on its GPU alone, in this
    # process group, while the other ranks wait -- the scaling curve's first point can be checked against BENCH without a
    # second launch.  After the timed region; not part of `value`.
    single_gpu_same_run = None
    if world > 1 and os.environ.get("DCA_BENCH_NO_SINGLE") != "1":
        if rank == 0:
            c1 = _lib.Context(local_rank, _lib.DCA_F64 if args.precision == 64 else _lib.DCA_F32)
            c1.set_msa(X, q)
            c1.set_weight_counts(counts)
            c1.plm_configure(lh, lJ, _lib.CARRY_CHUNKED)
            c1.plm_init_x()
            c1.plm_lbfgs_begin(total_cap + 1000)
            s1 = c1.plm_lbfgs_iterate(args.warmup) if args.warmup > 0 else None
            i1 = s1.iterations if s1 else 0
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            s1 = c1.plm_lbfgs_iterate(args.steps)
            torch.cuda.synchronize()
            d1 = time.perf_counter() - t1
            single_gpu_same_run = {"iterations_per_s": (s1.iterations - i1) / d1, "ms_per_step": d1 / max(s1.iterations - i1, 1) * 1e3,
                                   "steps": s1.iterations - i1, "what": "rank 0 alone on its GPU, same alignment / K / W, after the timed region"}
            c1.close()
        barrier()

    if rank != 0:
        if dist is not None:
            dist.destroy_process_group()
        return

    Lq = L * q
    P = ctx.num_params()
    n_local = ctx.N
    if comm_selection is not None and comm_selection["chosen_mode"] == 4:
        n_local = ctx.N / float(world)            # column strips: all sequences x 1/world of the columns = the same share of the adds
    esz = 4 if args.precision == 32 else 8
    # algorithmic HBM bytes per launch: compulsory reads + writes of each kernel
    alg_bytes = {
        "plm_logits": Lq * Lq * esz + n_local * L + n_local * Lq * esz,          # W once, alignment bytes, write S
        "plm_scatter": n_local * Lq * esz + Lq * Lq * esz + n_local * L * 2,     # read R, write G, 16-bit state images
    }
    # on-chip view.  One unit = one 512-byte row piece added to a running sum; N*L*(Lq*esz/512) units per launch.
    # LDS bytes actually read per unit: logits fetches the q rows of a site once per wave for its nseq sequences,
    # scatter reads every row once per wave for its two sites.
    units = n_local * L * (Lq * esz / 512.0)
    model = gather_issue_model(q, args.precision)
    lds_bytes = {k: units * model[k]["lds_bytes"] for k in ("plm_logits", "plm_scatter")}
    # the roofline's kernel: the stage the timed region clocked (the longer of the two gather stages by construction); should the
    # clock pass show the other one longer per launch, that one is reported -- from the clock pass, and said so
    per_launch = lambda k: ktimes[k][0] / max(ktimes[k][1], 1)
    dom = roof_stage if roof_stage in alg_bytes and ktimes[roof_stage][1] > 0 else max(alg_bytes, key=per_launch)
    longest = max(alg_bytes, key=per_launch)
    roof_measured_in = "timed region" if dom == roof_stage else "kernel clock pass"
    if longest != dom and per_launch(longest) > 1.05 * per_launch(dom):
        dom, roof_measured_in = longest, "kernel clock pass (the timed region clocked %s, which turned out shorter per launch)" % roof_stage
    ms, launches = ktimes[dom]
    avg_s = ms / max(launches, 1) / 1e3
    # HBM bytes per launch from the PMC counters (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, tools/profile_round.sh):
    # measured in a separate profiling run, NOT in this run -- reported with a fingerprint of the kernel sources it was
    # measured on, and flagged stale when the sources in this tree differ
    traffic, traffic_commit, traffic_stale = None, None, None
    tpath = os.path.join(ROOT, "profiles", "traffic.json")
    if os.path.exists(tpath):
        try:
            tj = json.load(open(tpath))
            traffic = tj.get(args.workload, {}).get(dom)
            traffic_commit = tj.get("measured_at_commit")
            traffic_stale = tj.get("kernel_sources_sha256") != kernel_sources_fingerprint()
        except Exception:
            traffic = None
    sums = n_local * L * Lq                       # (sequence, site, column) sums per launch
    adds = sums * model[dom]["adds"]              # packed-add lanes issued for them (one each, except on the q = 5 site-pair alphabet)
    valu_peak = 78.6 if args.precision == 32 else 39.3     # 157.3 TFLOP/s fp32 vector = 78.6e12 FMA slots/s; fp64 half of that
    hbm_view = {"bound": "hbm", "achieved": alg_bytes[dom] / avg_s / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": alg_bytes[dom] / avg_s / 1e9 / HBM_PEAK_GBS, "algorithmic_bytes": alg_bytes[dom]}
    # The dominant kernel is a register gather: its binding roof is the issue rate of the indexed packed adds (DESIGN.md
    # section 4), so THAT is the top-level object; the HBM view the contract names is the sub-object `hbm` (same launch,
    # same duration), `traffic` the measured fabric bytes per launch.
    roofline = {"kernel": dom, "measured_over": roof_measured_in, "bound": "valu", "achieved": adds / avg_s / 1e12, "peak": valu_peak, "unit": "Tadd/s",
                "frac": adds / avg_s / 1e12 / valu_peak, "traffic": traffic, "traffic_unit": "bytes per launch (PMC FETCH_SIZE + WRITE_SIZE, separate rocprofv3 run)",
                "traffic_measured_at_commit": traffic_commit, "traffic_is_stale": traffic_stale,
                "avg_kernel_ms": ms / max(launches, 1), "launches": launches,
                # one "launch" here = the stage of one evaluation, bracketed by HIP events on the library's stream; the scatter
                # stage is plm_scatter_kernel (main strips and, when the strip count is not a multiple of 8, the left-over strips
                # behind them in the same launch) + plm_sum_slabs_cols_kernel: a rocprofv3 kernel trace lists those separately
                "stage_kernels": {"plm_logits": ["plm_logits_kernel"],
                                  "plm_scatter": ["plm_scatter_kernel (main and left-over column strips in one launch)", "plm_sum_slabs_cols_kernel"]}[dom],
                "gather_formulation": model["formulation"], "issued_adds_per_sequence_site_column": model[dom]["adds"],
                "note": "gather kernel: bound on chip by the issue of the indexed packed adds (1 SALU + 1 VALU per 512-byte row piece), not by HBM; "
                        "peak = fp%d vector peak counted in adds" % (32 if args.precision == 32 else 64),
                "hbm": hbm_view,
                "onchip": {"bound": "lds", "achieved": lds_bytes[dom] / avg_s / 1e9, "peak": LDS_PEAK_GBS, "unit": "GB/s",
                           "frac": lds_bytes[dom] / avg_s / 1e9 / LDS_PEAK_GBS}}
    # both gather kernels, each against the same roof
    roofline["per_kernel"] = {}
    for k in alg_bytes:
        k_s = ktimes[k][0] / max(ktimes[k][1], 1) / 1e3
        if k_s > 0:
            roofline["per_kernel"][k] = {"avg_ms": k_s * 1e3, "valu_frac": sums * model[k]["adds"] / k_s / 1e12 / valu_peak,
                                         "hbm_frac": alg_bytes[k] / k_s / 1e9 / HBM_PEAK_GBS}
    kernels_ms = {k: {"avg_ms": v[0] / max(v[1], 1), "launches": v[1]} for k, v in ktimes.items()}

    out = {
        "metric": "plmDCA L-BFGS iterations/s (L=%d N=%d q=%d)" % (L, N, q),
        "value": steps_done / dt, "unit": "iterations/s", "n_gpus": world, "steps": steps_done, "warmup": args.warmup,
        "ms_per_step": dt / max(steps_done, 1) * 1e3, "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
        "dtype": "f32" if args.precision == 32 else "f64", "data": "synthetic",
        "config": {"workload": "plmdca compute_fn protein, synthetic MSA L=%d N=%d q=%d, lambda_h=%g lambda_J=%g seqid=0.8, "
                               "reference carry-over semantics (chunked scan)" % (L, N, q, lh, lJ),
                   "config_id": args.workload, "num_params": P, "parallelism": scheme},
        "evaluations_per_iteration": evals / max(steps_done, 1), "evaluations_per_s": evals / dt,
        "lbfgs_state": ("running" if not st.finished else "finished with libLBFGS status %d" % st.status), "fx": st.fx,
        "setup_s": {"generate_msa": t_gen, "weights_kernel": t_weights_ms / 1e3, "total_setup": t_setup},
        "kernels": kernels_ms, "kernel_clock_pass": clock_pass, "roofline": roofline,
        "host_cores": os.cpu_count(),
    }
    if world > 1:
        out["per_rank_ms_per_step"] = per_rank_ms
        if single_gpu_same_run is not None:
            out["single_gpu_same_run"] = single_gpu_same_run
            out["speedup_vs_single_gpu_same_run"] = out["value"] / single_gpu_same_run["iterations_per_s"]
    if world > 1 and comm_selection is not None:
        # bytes a rank puts on the wires per evaluation under each scheme (arithmetic from the layout; ring
        # all-reduce counted as reduce-scatter + all-gather)
        pb = float(P) * esz
        f = (world - 1) / float(world)
        comm_selection["wire_bytes_per_rank_per_evaluation"] = {
            "1": 2.0 * f * pb, "2": 2.0 * f * pb, "3": 2.0 * f * pb,
            "4": 2.0 * (float(Lq) * Lq / 2.0) * f * esz / world,
            "note": "sent per rank and evaluation; 1-3: reduce-scatter(g) + all-gather(x) of the P-vector (1: as one all-reduce); "
                    "4: couplings up + gradient-table rows down, (L q)^2 / 2 (1 - 1/world) elements each way over the node, averaged over the ranks"}
        out["comm_selection"] = comm_selection
        # the one-GPU prediction of this run (tools/scaling_prediction.py -> profiles/r06_scaling_prediction.json), if one was made for
        # this workload: measured / predicted turns the first node run into a test of the model
        try:
            with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "r06_scaling_prediction.json")) as fh:
                pred = json.load(fh)
            e = pred["worlds"].get(str(world), {})
            ms = e.get("predicted_ms_per_step", {}).get(str(comm_selection["chosen_mode"]))
            if pred.get("workload") == args.workload and args.precision == 32 and ms:
                out["prediction"] = {"predicted_ms_per_step": ms, "measured_ms_per_step": out["ms_per_step"],
                                     "measured_over_predicted": out["ms_per_step"] / ms, "all_schemes_predicted_ms_per_step": e.get("predicted_ms_per_step"),
                                     "source": "profiles/r06_scaling_prediction.json (per-rank kernel time on ONE GPU + wire arithmetic)"}
        except (OSError, ValueError, KeyError):
            pass

    if world == 1 and not args.no_mfdca:
        # second half of the headline metric: mfDCA residue pairs/s, encoded MSA on host ->
        # FN_APC scores ranked on host (weights + counts + C + inverse + scoring + sort)
        # one untimed pass first (like the warm-up iterations of the plmDCA leg: first-use code
        # loading and allocator growth are not part of the metric), then three timed passes, each on a
        # fresh context; the MEDIAN is reported and all three are listed (a shared box shows
        # occasional 2-3x outliers in host-side allocation time)
        samples = []
        mctx = None
        for rep in range(4):
            if mctx is not None:
                mctx.close()
            mctx = _lib.Context(local_rank, _lib.DCA_F64)
            t0 = time.perf_counter()
            mctx.set_msa(X, q)
            mctx.set_profiling(True)
            mctx.compute_weights(0.8, _lib.DCA_F64)
            scores = mctx.mf_run(0.5, True)
            order = mctx.scores_order()          # ranked on the device (stable radix sort)
            if rep:
                samples.append(time.perf_counter() - t0)
        t_mf = float(np.median(samples))
        npairs = L * (L - 1) // 2
        out["mfdca"] = {"pairs_per_s": npairs / t_mf, "seconds": t_mf, "statistic": "median of the timed passes", "samples_s": samples,
                        "fastest_pairs_per_s": npairs / min(samples), "pairs": npairs, "top_pair_index": int(order[0]),
                        "stages_ms": {k: mctx.kernel_time(k)[0] for k in ("weights", "mf_sort", "mf_counts", "mf_inverse", "scores")},
                        "inverse_flops": float((L * (q - 1)) ** 3),
                        "inverse_tflops": float((L * (q - 1)) ** 3) / max(mctx.kernel_time("mf_inverse")[0], 1e-9) / 1e9}
        inv_tf = out["mfdca"]["inverse_tflops"]
        # the MFMA half of the headline metric: SPD inverse on v_mfma_f64_16x16x4_f64, n^3 flop (Cholesky n^3/3 +
        # triangular inverse n^3/3 + X^T X n^3/3; the reference's LU route would count 2 n^3)
        out["roofline_mf"] = {"kernel": "mf_inverse", "bound": "mfma_f64", "achieved": inv_tf, "peak": 78.6, "unit": "TFLOP/s",
                              "frac": inv_tf / 78.6, "flop_convention": "n^3", "n": L * (q - 1),
                              "avg_kernel_ms": mctx.kernel_time("mf_inverse")[0] / max(mctx.kernel_time("mf_inverse")[1], 1)}
        # the other kernels SURVEY 8 d3 names, each against the roof that binds it (K2 weights, K6 optimiser vectors, M2 pair counts)
        t_cnt = mctx.kernel_time("mf_counts")[0] / max(mctx.kernel_time("mf_counts")[1], 1) / 1e3
        t_w = mctx.kernel_time("weights")[0] / max(mctx.kernel_time("weights")[1], 1) / 1e3
        t_vec = ktimes["lbfgs_vec"][0] / steps_clocked / 1e3
        more = {}
        if t_w > 0:
            # the integer work the kernel ISSUED (counted by the kernel itself: wave x 32-site groups compared, each 16 pairs
            # per lane x (planes xor / or + 1 popcount-add) VALU instructions) against the integer-VALU issue rate -- one wave
            # instruction per SIMD every 4 clocks; the exact early exit skips the rest of the N^2 L / 2 comparisons, which the
            # round-4 figure still counted (a "fraction" of 1.56)
            # one extra, untimed pass with the kernel's work counter on (DCA_WEIGHTS_WORK: off in the timed passes)
            os.environ["DCA_WEIGHTS_WORK"] = "1"
            wctx = _lib.Context(local_rank, _lib.DCA_F64)
            wctx.set_msa(X, q)
            wctx.compute_weights(0.8, _lib.DCA_F64)
            groups, groups_all, planes = wctx.weights_work()
            wctx.close()
            del os.environ["DCA_WEIGHTS_WORK"]
            valu_per_group = 16.0 * (planes + 3.0)          # q <= 32: 5 xor + 2 or3 + 1 bcnt = 8; q <= 8: 3 + 1 + 1 = 5
            peak = 256 * 4 * 2.4e9 / 4.0                    # wave instructions per second
            issued = groups * valu_per_group / t_w
            more["weights_K2"] = {"bound": "integer valu", "achieved": issued / 1e9, "peak": peak / 1e9, "unit": "G wave instructions/s",
                                  "frac": issued / peak, "avg_kernel_ms": t_w * 1e3,
                                  "site_groups_compared_fraction": groups / max(groups_all, 1),
                                  "site_comparisons_per_s_incl_skipped": N * N * L / 2.0 / t_w,
                                  "note": "issued xor / or / popcount instructions of the comparisons (counted in the kernel) over the kernel time; "
                                          "the exact early exit compared only `site_groups_compared_fraction` of the 32-site groups"}
        if t_vec > 0:
            vec_bytes = (4 * 5 + 12) * esz * float(P)
            more["lbfgs_vectors_K6"] = {"bound": "hbm", "achieved": vec_bytes / t_vec / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                        "frac": vec_bytes / t_vec / 1e9 / HBM_PEAK_GBS, "ms_per_iteration": t_vec * 1e3,
                                        "note": "(4 m + 12) P elements per iteration, m = 5 (SURVEY 8 d3)"}
        if t_cnt > 0:
            upd = N * L * (L - 1) / 2.0
            cnt_bytes = N * L + 8.0 * N + 8.0 * Lq * Lq
            # peak: what a loop of nothing but ds_add_f64 on this histogram shape sustains on this chip (tools/experiments/
            # lds_atomic_rate.hip, profiles/r05_lds_atomic_rate.txt: 8.1 lane updates per clock and CU = 4.4 T updates/s; round 4
            # assumed 16 per clock at 2.4 GHz without measuring it)
            lds_peak = 4.4e12
            more["pair_counts_M2"] = {"bound": "lds atomics", "achieved": upd / t_cnt / 1e12, "peak": lds_peak / 1e12, "unit": "T weighted updates/s",
                                      "frac": upd / t_cnt / lds_peak, "avg_kernel_ms": t_cnt * 1e3,
                                      "hbm": {"achieved": cnt_bytes / t_cnt / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": cnt_bytes / t_cnt / 1e9 / HBM_PEAK_GBS},
                                      "note": "histogram formulation (N L^2 / 2 ds_add_f64), not the 21 x larger one-hot GEMM; peak = the measured rate of a bare ds_add_f64 loop on the same histogram shape"}
        out["roofline_more"] = more
        mctx.close()

    if world == 1 and args.precision == 32 and not args.no_modes:
        # The two precisions of the product side by side, each with its MEASURED parity class.  north_star's tolerance
        # ("FN / DI within 1e-4 of the float64 CPU path, identical top-L") is protocol P3 of SURVEY 8c4; the float32 mode is the
        # reference's own arithmetic (lbfgs.h:50-62) and lands in the P4 regime after 100 iterations of an optimisation
        # that does not converge.  The figures come from the committed report of tests/test_gpu_configs.py::
        # test_P3_full_size_at_reference_cap against the float64 oracle's golden run at this configuration.
        try:
            out["modes"] = precision_modes(_lib, X, q, lh, lJ, local_rank, args.workload, out["value"], out["ms_per_step"])
        except Exception as exc:   # pragma: no cover
            out["modes"] = {"error": repr(exc)}

    if world == 1 and not args.no_e2e:
        try:
            out["e2e"] = end_to_end(X, L, q, lh, lJ, local_rank)
        except Exception as exc:   # pragma: no cover
            out["e2e"] = {"error": repr(exc)}

    if world == 1 and args.workload == "D" and not args.no_rna:
        try:
            out["workload_E"] = rna_workload_block(local_rank)
        except Exception as exc:   # pragma: no cover
            out["workload_E"] = {"error": repr(exc)}

    if world == 1 and not args.no_cpu_baseline:
        sample = args.cpu_sample or max(200, min(N, int(4000 * (500.0 / L) ** 2 * (21.0 / q))))
        sample = min(sample, N)
        try:
            out["cpu_baseline"] = cpu_baseline(X, q, lh, lJ, max(evals / max(steps_done, 1), 1.0), sample)
            if args.cpu_full:
                from oracle import plm as oplm
                threads = max(1, min(os.cpu_count() or 1, L))
                t_full, kind_full = _time_reference_eval(oplm, X, q, lh, lJ, threads)
                epi = max(evals / max(steps_done, 1), 1.0)
                out["cpu_baseline"]["full_evaluation"] = {
                    "seconds_per_evaluation": t_full, "value": 1.0 / (t_full * epi), "unit": "L-BFGS iterations/s", "cores": threads, "kind": kind_full,
                    "fit_over_measured": out["cpu_baseline"]["seconds_per_evaluation"] / t_full,
                    "sample": "ONE objective+gradient evaluation of the reference on ALL %d sequences (no extrapolation)" % N}
        except Exception as exc:   # pragma: no cover
            out["cpu_baseline"] = {"error": repr(exc)}

    # what was asked for is what ran: N ranks, and (native exchange) ONE communicator of N ranks
    if out["n_gpus"] != args.gpus:
        raise SystemExit("bench.py: measured %d rank(s) for --gpus %d" % (out["n_gpus"], args.gpus))
    if comm_selection is not None and comm_selection["rccl_ranks"] != world:
        raise SystemExit("bench.py: the RCCL communicator has %d ranks, expected %d" % (comm_selection["rccl_ranks"], world))
    print(json.dumps(out))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
