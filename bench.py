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


def cpu_baseline(X, q, lh, lJ, evals_per_iter, sample_rows):
    """Reference C++/OpenMP gradient (oracle/_ref, kind "reference") -- or the C restatement
    (kind "port") when the reference build is absent -- timed on this host's cores on the
    first `sample_rows` sequences; evaluation cost is linear in N."""
    from oracle import plm as oplm
    from tools.gen_msa import write_fasta
    cores = os.cpu_count() or 1
    L = X.shape[1]
    Xs = np.ascontiguousarray(X[:sample_rows])
    threads = max(1, min(cores, L))
    t_eval, kind = None, None
    if oplm.have_reference():
        try:
            path = "/tmp/bench_cpu_sample_%d.fa" % os.getpid()
            write_fasta(path, Xs, q)
            ref = oplm.Reference(path, 1 if q == 21 else 2, L, q, 0.8, lh, lJ, threads=threads)
            x = ref.init_x()
            ref.gradient(x)                       # warm
            t0 = time.perf_counter()
            ref.gradient(x)
            t_eval = time.perf_counter() - t0
            ref.close()
            os.unlink(path)
            kind = "reference"
        except Exception as exc:                   # pragma: no cover
            print("reference baseline unavailable: %r" % (exc,), file=sys.stderr)
    if t_eval is None:
        w = oplm.weights(Xs, 0.8, np.float32, threads=threads)
        x = oplm.init_x(Xs, w, q)
        t0 = time.perf_counter()
        oplm.gradient(Xs, w, q, lh, lJ, x, carry=True, threads=threads)
        t_eval = time.perf_counter() - t0
        kind = "port"
    full_eval = t_eval * X.shape[0] / sample_rows
    return {"value": 1.0 / (full_eval * evals_per_iter), "unit": "L-BFGS iterations/s", "cores": threads, "kind": kind,
            "sample": "one objective+gradient evaluation on the first %d of %d sequences (L=%d, q=%d) took %.2f s; "
                      "cost is linear in N, scaled to N and divided by the %.2f evaluations/iteration measured on the GPU run"
                      % (sample_rows, X.shape[0], L, q, t_eval, evals_per_iter)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="D", choices=sorted(WORKLOADS))
    ap.add_argument("--precision", type=int, default=32, choices=(32, 64))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-mfdca", action="store_true")
    ap.add_argument("--cpu-sample", type=int, default=0)
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus and world > 1:
        args.gpus = world

    import torch
    from pydca_amd import _lib, parallel
    from tools.gen_msa import dedup, generate

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: torch.cuda.is_available() is False (no CPU fallback)")
    # self-test of the multi-process path on a box with ONE GPU: DCA_BENCH_SELFTEST=1 puts every rank on
    # device 0 and uses gloo (RCCL refuses two ranks on one device); numbers from it are meaningless
    selftest = os.environ.get("DCA_BENCH_SELFTEST") == "1"
    if selftest:
        local_rank = 0
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
    full = _lib.Context(local_rank, _lib.DCA_F32)
    full.set_msa(X, q)
    full.set_profiling(True)
    full.compute_weights(0.8, _lib.DCA_F32)
    counts = full.weight_counts()
    w32 = (np.float32(1.0) / counts.astype(np.float32)).astype(np.float32)
    t_weights_ms, _ = full.kernel_time("weights")
    hook = None
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
        # four hook calls per iteration, so vector sharding is used from 16 MB of parameters on.
        small = ctx.num_params() * (4 if args.precision == 32 else 8) < (16 << 20)
        if os.environ.get("DCA_BENCH_ALLREDUCE") == "1" or (small and os.environ.get("DCA_BENCH_VECTORS") != "1"):
            hook = parallel.TorchAllReduceHook(local_rank)
            ctx.plm_set_reduce_hook(hook)
        else:
            hook = parallel.TorchVectorComm(local_rank, rank, world)
            ctx.plm_set_vector_sharding(rank, world, hook)
    t_setup = time.perf_counter() - t0

    # ---- warm-up iterations, then exactly K timed iterations
    total_cap = args.warmup + args.steps
    ctx.plm_lbfgs_begin(total_cap + 1000)
    st = ctx.plm_lbfgs_iterate(args.warmup) if args.warmup > 0 else None
    it0 = st.iterations if st else 0
    ev0 = st.evaluations if st else 1
    ctx.set_profiling(True)
    ctx.reset_kernel_times()
    barrier()
    t0 = time.perf_counter()
    st = ctx.plm_lbfgs_iterate(args.steps)
    barrier()
    dt = time.perf_counter() - t0
    if dist is not None:
        tmax = torch.tensor([dt], dtype=torch.float64, device="cuda")
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())
    steps_done = st.iterations - it0
    evals = st.evaluations - ev0
    ktimes = {tag: ctx.kernel_time(tag) for tag in ("plm_expand", "plm_logits", "plm_softmax", "plm_scatter", "plm_fold", "lbfgs_vec")}
    ctx.set_profiling(False)

    if rank != 0:
        if dist is not None:
            dist.destroy_process_group()
        return

    Lq = L * q
    P = ctx.num_params()
    n_local = ctx.N
    esz = 4 if args.precision == 32 else 8
    # algorithmic HBM bytes per launch: compulsory reads + writes of each kernel
    alg_bytes = {
        "plm_logits": Lq * Lq * esz + n_local * L + n_local * Lq * esz,          # W once, alignment bytes, write S
        "plm_scatter": n_local * Lq * esz + Lq * Lq * esz + n_local * L * 2,     # read R, write G, 16-bit state images
    }
    # on-chip view.  One unit = one 512-byte row piece added to a running sum; N*L*(Lq*esz/512) units per launch.
    # LDS bytes actually read per unit: logits fetches the q rows of a site once per wave for its nseq sequences,
    # scatter reads every row once per wave for its two sites.
    from tools.gen_plm_asm import LOGITS_CFG
    units = n_local * L * (Lq * esz / 512.0)
    lds_bytes = {"plm_logits": units * q * 512.0 / LOGITS_CFG[q][1], "plm_scatter": units * 512.0 / 2}
    dom = max(alg_bytes, key=lambda k: ktimes[k][0])
    ms, launches = ktimes[dom]
    avg_s = ms / max(launches, 1) / 1e3
    traffic = None
    tpath = os.path.join(ROOT, "profiles", "traffic.json")
    if os.path.exists(tpath):
        try:
            traffic = json.load(open(tpath)).get(args.workload, {}).get(dom)
        except Exception:
            traffic = None
    roofline = {"kernel": dom, "bound": "hbm", "achieved": alg_bytes[dom] / avg_s / 1e9, "peak": HBM_PEAK_GBS,
                "unit": "GB/s", "frac": alg_bytes[dom] / avg_s / 1e9 / HBM_PEAK_GBS, "traffic": traffic,
                "avg_kernel_ms": ms / max(launches, 1), "launches": launches,
                "note": "gather kernels: bound on chip (VALU/SALU issue of the indexed adds, LDS reads), not by HBM (DESIGN.md section 4); see valu / onchip",
                "onchip": {"bound": "lds", "achieved": lds_bytes[dom] / avg_s / 1e9, "peak": LDS_PEAK_GBS, "unit": "GB/s",
                           "frac": lds_bytes[dom] / avg_s / 1e9 / LDS_PEAK_GBS},
                # the adds themselves: N*L*Lq fp32 (fp64) adds per launch against the vector peak counted in
                # adds (157.3 TFLOP/s fp32 = 78.6e12 FMA slots/s; fp64 half of that)
                "valu": {"bound": "valu", "achieved": n_local * L * Lq / avg_s / 1e12,
                         "peak": 78.6 if args.precision == 32 else 39.3, "unit": "Tadd/s",
                         "frac": n_local * L * Lq / avg_s / 1e12 / (78.6 if args.precision == 32 else 39.3)}}
    kernels_ms = {k: {"avg_ms": v[0] / max(v[1], 1), "launches": v[1]} for k, v in ktimes.items()}

    out = {
        "metric": "plmDCA L-BFGS iterations/s (L=%d N=%d q=%d)" % (L, N, q),
        "value": steps_done / dt, "unit": "iterations/s", "n_gpus": world, "steps": steps_done, "warmup": args.warmup,
        "ms_per_step": dt / max(steps_done, 1) * 1e3, "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
        "dtype": "f32" if args.precision == 32 else "f64", "data": "synthetic",
        "config": {"workload": "plmdca compute_fn protein, synthetic MSA L=%d N=%d q=%d, lambda_h=%g lambda_J=%g seqid=0.8, "
                               "reference carry-over semantics (chunked scan)" % (L, N, q, lh, lJ),
                   "config_id": args.workload, "num_params": P, "parallelism": ("1 GPU" if world == 1 else ("sequences sharded x%d, all-reduce(g) over RCCL" % world) if isinstance(hook, parallel.TorchAllReduceHook)
                                   else "sequences sharded x%d, reduce-scatter(g) + all-gather(x) over RCCL, L-BFGS vectors sharded x%d" % (world, world))},
        "evaluations_per_iteration": evals / max(steps_done, 1), "evaluations_per_s": evals / dt,
        "lbfgs_status": st.status, "fx": st.fx,
        "setup_s": {"generate_msa": t_gen, "weights_kernel": t_weights_ms / 1e3, "total_setup": t_setup},
        "kernels": kernels_ms, "roofline": roofline,
        "host_cores": os.cpu_count(),
    }

    if world == 1 and not args.no_mfdca:
        # second half of the headline metric: mfDCA residue pairs/s, encoded MSA on host ->
        # FN_APC scores ranked on host (weights + counts + C + inverse + scoring + sort)
        # one untimed pass first (like the warm-up iterations of the plmDCA leg: first-use code
        # loading and allocator growth are not part of the metric), then three timed passes, each on a
        # fresh context; the fastest is reported and all three are listed (a shared box shows
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
        t_mf = min(samples)
        npairs = L * (L - 1) // 2
        out["mfdca"] = {"pairs_per_s": npairs / t_mf, "seconds": t_mf, "samples_s": samples, "pairs": npairs, "top_pair_index": int(order[0]),
                        "stages_ms": {k: mctx.kernel_time(k)[0] for k in ("weights", "mf_sort", "mf_counts", "mf_inverse", "scores")},
                        "inverse_flops": float((L * (q - 1)) ** 3),
                        "inverse_tflops": float((L * (q - 1)) ** 3) / max(mctx.kernel_time("mf_inverse")[0], 1e-9) / 1e9}
        mctx.close()

    if world == 1 and not args.no_cpu_baseline:
        sample = args.cpu_sample or max(200, min(N, int(4000 * (500.0 / L) ** 2 * (21.0 / q))))
        sample = min(sample, N)
        try:
            out["cpu_baseline"] = cpu_baseline(X, q, lh, lJ, max(evals / max(steps_done, 1), 1.0), sample)
        except Exception as exc:   # pragma: no cover
            out["cpu_baseline"] = {"error": repr(exc)}

    print(json.dumps(out))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
