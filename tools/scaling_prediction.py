#!/usr/bin/env python3
"""A PREDICTION of `bench.py --gpus N` for N = 1, 2, 4, 8 from ONE GPU (round-5 verdict, item 6) -- not a scaling curve.

For every world size: the kernel time of ONE rank's iteration, measured on this GPU at the rank's share of the work
  * sequence-sharded schemes (1 all-reduce, 2 reduce-scatter + all-gather, 3 direct exchange): the first N / world sequences of the
    alignment (+ the halo of the chunked scan), whole parameter vector evaluated; schemes 2 / 3 divide the optimiser's vector work
    by world, scheme 1 keeps it whole
  * column strips (scheme 4): rank r's window of the sites over all sequences, through the analysis build of the library
    (`make -C pydca_amd/csrc ablate`, DCA_STRIP_EMULATE=r,world: the exchanges are skipped, the kernels are those of one rank)
plus the wire arithmetic of DESIGN.md section 6 (bytes a rank sends per evaluation / the rate assumed per rank: 7 xGMI links of
~64 GB/s effective each way for all-to-all patterns, ONE link for a ring hop).  `bench.py --gpus N` prints its measured
ms_per_step beside `predicted_ms_per_step` of the scheme it ran and the ratio.

    python tools/scaling_prediction.py [--workload D] > profiles/r06_scaling_prediction.json"""
import argparse
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

WORKLOADS = {"C": (200, 10000, 21, 1.0, 50.0, 12345), "D": (500, 50000, 21, 1.0, 50.0, 12346), "E": (150, 200000, 5, 29.8, 29.8, 12347)}
LINK_GBS = 64.0          # effective bytes/s one xGMI link moves one way (153 GB/s raw per link and direction pair)
LINKS = 7

CHILD = r'''
import json, os, sys
import numpy as np
sys.path.insert(0, %(root)r)
from pydca_amd import _lib
from tools.gen_msa import dedup, generate
L, N, q, lh, lJ, seed = %(wl)r
X = dedup(generate(L, N, q, seed))
n = X.shape[0]
world, mode = %(world)d, %(mode)r
ctx = _lib.Context(0, _lib.DCA_F32)
if mode == "seq":
    hi = (n + world - 1) // world
    warm = 40
    ctx.set_msa(X[:min(n, hi)], q)          # rank 0's block (no halo in front of the first block; the others walk `warm` rows more)
else:
    ctx.set_msa(X, q)
ctx.compute_weights(0.8, _lib.DCA_F32)
ctx.plm_configure(lh, lJ)
ctx.plm_init_x()
keys = ("plm_expand", "plm_logits", "plm_softmax", "plm_scatter", "plm_fold", "lbfgs_vec")
its = 10
if mode == "seq":
    ctx.plm_lbfgs_begin(1000)
    ev0 = ctx.plm_lbfgs_iterate(3).evaluations
    ctx.set_profiling(True)
    ctx.reset_kernel_times()
    evs = ctx.plm_lbfgs_iterate(its).evaluations - ev0
    out = {k: ctx.kernel_time(k)[0] / max(1, evs if k != "lbfgs_vec" else its) for k in keys}
    epi = evs / its
else:
    # the emulated window skips the exchanges: its numbers are wrong, its kernel times are one rank's -- evaluations only
    ctx.plm_gradient()
    ctx.set_profiling(True)
    ctx.reset_kernel_times()
    for _ in range(its):
        ctx.plm_gradient()
    out = {k: ctx.kernel_time(k)[0] / its for k in keys}
    epi = None
print(json.dumps(dict(kernels_ms=out, sequences=int(ctx.weights().size), evaluations_per_iteration=epi)))
'''


def child(wl, world, mode, env=None):
    code = CHILD % dict(root=ROOT, wl=wl, world=world, mode=mode)
    p = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=900, env=dict(os.environ, **(env or {})))
    if p.returncode != 0:
        return {"error": p.stderr[-400:]}
    return json.loads(p.stdout.strip().splitlines()[-1])


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="D", choices=sorted(WORKLOADS))
    a = ap.parse_args()
    wl = WORKLOADS[a.workload]
    L, N, q = wl[:3]
    P = L * q + L * (L - 1) // 2 * q * q
    esz = 4
    ablate = os.path.join(ROOT, "pydca_amd", "lib", "libdca_hip_ablate.so")
    out = {"what": "PREDICTION from one GPU, not a measurement of several: per-rank kernel time at the rank's share + wire arithmetic",
           "workload": a.workload, "L": L, "N": N, "q": q, "parameters": P, "dtype": "f32",
           "assumed_link_GBs_one_way": LINK_GBS, "links_per_gpu": LINKS, "worlds": {}}
    for world in (1, 2, 4, 8):
        seq = child(wl, world, "seq")
        entry = {"sequence_sharded_rank": seq}
        if "kernels_ms" in seq:
            k = seq["kernels_ms"]
            ev = k["plm_expand"] + k["plm_logits"] + k["plm_softmax"] + k["plm_scatter"] + k["plm_fold"]
            epi = seq["evaluations_per_iteration"]
            f = (world - 1) / world
            wire = 2.0 * f * P * esz                       # reduce-scatter + all-gather of the P-vector (1: as one all-reduce)
            ring_ms = wire / (LINK_GBS * 1e9) * 1e3        # a ring moves every byte over ONE link per hop
            mesh_ms = wire / (LINK_GBS * 1e9 * min(LINKS, max(world - 1, 1))) * 1e3
            vec1 = k["lbfgs_vec"] * world                  # measured on the rank's sequences but the WHOLE vector: already whole
            vec = k["lbfgs_vec"]
            # every link of the rank busy (what RCCL's several rings / the direct exchange aim at) ...
            entry["predicted_ms_per_step"] = {
                "1": ev * epi + vec + (mesh_ms * epi if world > 1 else 0.0),
                "2": ev * epi + vec / world + (mesh_ms * epi if world > 1 else 0.0),
                "3": ev * epi + vec / world + (mesh_ms * epi if world > 1 else 0.0)}
            # ... and the pessimistic end for the ring collectives: every byte over ONE link per hop
            entry["predicted_ms_per_step_one_link_ring"] = {"1": ev * epi + vec + (ring_ms * epi if world > 1 else 0.0),
                                                            "2": ev * epi + vec / world + (ring_ms * epi if world > 1 else 0.0)}
            entry["wire_bytes_per_rank_per_evaluation"] = {"1": wire, "2": wire, "3": wire}
            entry["wire_ms_per_evaluation"] = {"ring": ring_ms, "full_mesh": mesh_ms}
            del vec1
        if world > 1 and os.path.exists(ablate):
            ranks = {}
            for r in sorted({0, world // 2, world - 1}):
                ranks[str(r)] = child(wl, world, "strip", env={"DCA_LIB_PATH": ablate, "DCA_STRIP_EMULATE": "%d,%d" % (r, world)})
            entry["column_strip_ranks"] = ranks
            ok = [v for v in ranks.values() if "kernels_ms" in v]
            if ok:
                slow = max(ok, key=lambda v: sum(v["kernels_ms"][x] for x in ("plm_expand", "plm_logits", "plm_softmax", "plm_scatter", "plm_fold")))
                k = slow["kernels_ms"]
                ev = k["plm_expand"] + k["plm_logits"] + k["plm_softmax"] + k["plm_scatter"] + k["plm_fold"]
                Lq = L * q
                wire4 = 2.0 * (Lq * Lq / 2.0) * (world - 1) / world * esz / world
                mesh4 = wire4 / (LINK_GBS * 1e9 * min(LINKS, world - 1)) * 1e3
                epi4 = seq.get("evaluations_per_iteration") or 1.0
                # the owned parameters are uneven: rank 0 owns the pairs (i, j) with i among the first L / world sites, 2 / w - 1 / w^2 of
                # them (23 % at 8 ranks): the slowest rank's share of the optimiser's vector work
                vec4 = seq["kernels_ms"]["lbfgs_vec"] * (2.0 / world - 1.0 / world ** 2) if "kernels_ms" in seq else 0.0
                entry.setdefault("predicted_ms_per_step", {})["4"] = ev * epi4 + vec4 + mesh4 * epi4
                entry.setdefault("wire_bytes_per_rank_per_evaluation", {})["4"] = wire4
                entry.setdefault("wire_ms_per_evaluation", {})["strips_full_mesh"] = mesh4
        elif world > 1:
            entry["column_strip_ranks"] = "analysis build missing: make -C pydca_amd/csrc ablate"
        out["worlds"][str(world)] = entry
    base = out["worlds"]["1"].get("predicted_ms_per_step", {}).get("2")
    if base:
        for w, e in out["worlds"].items():
            if "predicted_ms_per_step" in e:
                e["predicted_speedup_vs_1"] = {s: base / v for s, v in e["predicted_ms_per_step"].items()}
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
