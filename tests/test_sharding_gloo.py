"""The N>1 path on CPU: two gloo ranks, each evaluating its sequence shard (halo rows only
warm up the carry-over scan), one all-reduce(sum) of g and fx -- must reproduce the unsharded
evaluation.  On CPU the per-shard arithmetic is done by the oracle (tests may use it: this validates the
partition / halo / regulariser protocol); the second half of the file runs the SAME two-rank gloo protocol with the
product's kernels doing the arithmetic (marked gpu: both ranks on the box's one device)."""
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


# ---------------------------------------------------------------------------------------------------------------------
# The same two-rank protocol with the PRODUCT's kernels doing the per-shard arithmetic (GPU box: both ranks on the one
# device, gloo instead of RCCL, which refuses two ranks per device): sharded contexts of libdca_hip.so, the library's
# reduce hook carried by torch.distributed, evaluation and a few L-BFGS iterations against the unsharded product run
# and the oracle.
def _worker_product(rank, world, port, carry, outdir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from pydca_amd import _lib, parallel
    G = golden("plm_rf71")
    X, q, L = G["X"], int(G["q"]), int(G["L"])
    full = _lib.Context(0, _lib.DCA_F64)
    full.set_msa(X, q)
    # sequence weights: every rank counts its share of the tile pairs, the integer counts are summed over the ranks
    part = torch.from_numpy(full.weights_partial_counts(0.8, _lib.DCA_F64, rank, world).astype(np.int64))
    dist.all_reduce(part)
    full.set_weight_counts(part.numpy().astype(np.uint32))
    w = full.weights()
    full.close()
    mode = _lib.CARRY_CHUNKED if carry else _lib.CARRY_EXACT
    ctx = parallel.make_sharded_plm_context(_lib, X, q, w, 1.0, 20.0, rank, world, 0, precision=64, carry_mode=mode)
    hook = parallel.TorchAllReduceHook(0)
    ctx.plm_set_reduce_hook(hook)
    x = perturbed(parallel.initial_x(X, w, q, np.float64), L, q)
    ctx.plm_set_x(x)
    fx = ctx.plm_gradient()
    g = ctx.plm_get_g(np.float64)
    ctx.plm_lbfgs_begin(6)
    st = ctx.plm_lbfgs_iterate(6)
    xe = ctx.plm_get_x(np.float64)
    ctx.close()
    np.savez(os.path.join(outdir, "product_%d_rank%d.npz" % (int(carry), rank)), g=g, fx=fx, w=w, x0=x, x_end=xe,
             stats=np.array([st.status, st.iterations, st.evaluations]), fx_end=st.fx, hook_calls=hook.calls)
    dist.destroy_process_group()


@pytest.mark.gpu
@pytest.mark.parametrize("carry", [False, True])
def test_two_rank_gloo_product_kernels_equal_unsharded(tmp_path, carry):
    world = 2
    mp.spawn(_worker_product, args=(world, _free_port(), carry, str(tmp_path)), nprocs=world, join=True)
    from oracle import plm as oplm
    from pydca_amd import _lib
    G = golden("plm_rf71")
    X, q, L = G["X"], int(G["q"]), int(G["L"])
    R = [np.load(os.path.join(str(tmp_path), "product_%d_rank%d.npz" % (int(carry), r))) for r in range(world)]
    w = oplm.weights(X, 0.8, np.float64)
    assert np.array_equal(R[0]["w"], w) and np.array_equal(R[1]["w"], w)             # sharded identity counts: exact
    x0 = R[0]["x0"]
    fx_o, g_o = oplm.gradient(X, w, q, 1.0, 20.0, x0, carry=carry, threads=4)
    ref = _lib.Context(0, _lib.DCA_F64)
    ref.set_msa(X, q)
    ref.compute_weights(0.8, _lib.DCA_F64)
    ref.plm_configure(1.0, 20.0, _lib.CARRY_CHUNKED if carry else _lib.CARRY_EXACT)
    ref.plm_set_x(x0)
    ref.plm_gradient()
    ref.plm_lbfgs_begin(6)
    st = ref.plm_lbfgs_iterate(6)
    x_ref = ref.plm_get_x(np.float64)
    ref.close()
    for r in R:
        assert int(r["hook_calls"]) >= 7                                            # one exchange per evaluation
        assert abs(float(r["fx"]) - fx_o) <= 1e-11 * abs(fx_o)
        assert np.linalg.norm(r["g"] - g_o) <= 1e-11 * np.linalg.norm(g_o)
        assert list(r["stats"]) == [st.status, st.iterations, st.evaluations]
        assert abs(float(r["fx_end"]) - st.fx) <= 1e-9 * abs(st.fx)
        assert np.linalg.norm(r["x_end"] - x_ref) <= 1e-8 * np.linalg.norm(x_ref)
    assert np.array_equal(R[0]["x_end"], R[1]["x_end"])                              # both ranks hold the same parameters
