#!/usr/bin/env python3
"""Exchange-step overhead of a sharded plmDCA run, measured on ONE GPU at per-rank shard sizes (gpurun has one GPU; a
1-rank communicator makes every collective an identity, so what is timed is the exchange machinery itself: RCCL
launches on the library's stream against the torch.distributed hooks with their Python callback and device syncs).

    python tools/comm_overhead.py            # configs C and D at N/8 sequences per rank
"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pydca_amd import _lib, parallel  # noqa: E402
from tools.gen_msa import SEEDS, dedup, generate  # noqa: E402

CONFIGS = {"C": (200, 10000, 21, 1.0, 50.0), "D": (500, 50000, 21, 1.0, 50.0)}


def run(tag, scheme, iters=12, shards=8):
    import torch
    import torch.distributed as dist
    L, N, q, lh, lJ = CONFIGS[tag]
    X = dedup(generate(L, N, q, SEEDS[tag]))
    full = _lib.Context(0, _lib.DCA_F32)
    full.set_msa(X, q)
    w = full.compute_weights(0.8, _lib.DCA_F32)
    full.close()
    ctx = parallel.make_sharded_plm_context(_lib, X, q, w, lh, lJ, 0, shards, 0)       # rank 0's shard of `shards`
    ctx.plm_set_x(parallel.initial_x(X, w.astype(np.float32), q, np.float32))
    hook = None
    if scheme == "native_vectors":
        parallel.init_native_comm(ctx, _lib, 0, 1)
        ctx.plm_set_native_comm(2)
    elif scheme == "native_allreduce":
        parallel.init_native_comm(ctx, _lib, 0, 1)
        ctx.plm_set_native_comm(1)
    elif scheme == "torch_vectors":
        hook = parallel.TorchVectorComm(0, 0, 1)
        ctx.plm_set_vector_sharding(0, 1, hook)
    elif scheme == "torch_allreduce":
        hook = parallel.TorchAllReduceHook(0)
        ctx.plm_set_reduce_hook(hook)
    ctx.plm_lbfgs_begin(1000)
    ctx.plm_lbfgs_iterate(3)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    st = ctx.plm_lbfgs_iterate(iters)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / iters * 1e3
    ctx.close()
    return {"config": tag, "sequences_per_rank": N // shards, "scheme": scheme, "ms_per_iteration": dt, "status": st.status}


def main():
    import torch
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29593")
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    rows = []
    for tag in ("C", "D"):
        for scheme in ("none", "native_allreduce", "native_vectors", "torch_allreduce", "torch_vectors"):
            r = run(tag, scheme)
            rows.append(r)
            print(json.dumps(r), flush=True)
    dist.destroy_process_group()
    out = os.path.join(ROOT, "gpurun_out", "comm_overhead.json")
    os.makedirs(os.path.dirname(out), exist_ok=True)
    json.dump(rows, open(out, "w"), indent=1)


if __name__ == "__main__":
    main()
