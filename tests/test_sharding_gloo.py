"""The N>1 path on CPU: two gloo ranks, each evaluating its sequence shard (halo rows only
warm up the carry-over scan), one all-reduce(sum) of g and fx -- must reproduce the unsharded
evaluation.  The per-shard arithmetic is done by the oracle here (tests may use it); the GPU
counterpart of the same protocol is tests/test_api_gpu.py::test_sharded_contexts_sum_to_unsharded."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT, golden, perturbed

sys.path.insert(0, ROOT)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, carry, outdir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle import plm as oplm
    from pydca_amd import parallel
    G = golden("plm_rf71")
    X, q, L = G["X"], int(G["q"]), int(G["L"])
    w = oplm.weights(X, 0.8, np.float64)
    x = perturbed(oplm.init_x(X, w, q), L, q)
    first, stop, halo = parallel.shard_with_halo(X.shape[0], world, rank, 40 if carry else 0)
    ws = w[first:stop].copy()
    ws[:halo] = 0.0                               # halo rows: scan only, no contribution
    lam_h, lam_J = (1.0, 20.0) if rank == 0 else (0.0, 0.0)   # regulariser on rank 0 only
    fx, g = oplm.gradient(X[first:stop], ws, q, lam_h, lam_J, x, carry=carry, threads=2)
    tg = torch.from_numpy(g)
    tf = torch.tensor([fx], dtype=torch.float64)
    dist.all_reduce(tg)
    dist.all_reduce(tf)
    if rank == 0:
        np.savez(os.path.join(outdir, "sharded_%d.npz" % int(carry)), g=tg.numpy(), fx=tf.numpy())
    dist.destroy_process_group()


@pytest.mark.parametrize("carry", [False, True])
def test_two_rank_gloo_allreduce_equals_unsharded(tmp_path, carry):
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), carry, str(tmp_path)), nprocs=world, join=True)
    from oracle import plm as oplm
    G = golden("plm_rf71")
    X, q, L = G["X"], int(G["q"]), int(G["L"])
    w = oplm.weights(X, 0.8, np.float64)
    x = perturbed(oplm.init_x(X, w, q), L, q)
    fx, g = oplm.gradient(X, w, q, 1.0, 20.0, x, carry=carry, threads=4)
    S = np.load(os.path.join(str(tmp_path), "sharded_%d.npz" % int(carry)))
    assert abs(S["fx"][0] - fx) <= 1e-11 * abs(fx)
    assert np.linalg.norm(S["g"] - g) <= 1e-11 * np.linalg.norm(g)
