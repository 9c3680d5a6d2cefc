#!/usr/bin/env python3
"""Drives the product's NATIVE exchange path (csrc/comm_rccl.cpp: RCCL entry points on the context's stream) with
`world` ranks as THREADS on one GPU.  librccl.so is replaced by tests/fake_rccl/libfake_rccl.so (test infrastructure: the
same entry points between threads) because real RCCL refuses two ranks on one device; the library itself is the product
build, unmodified -- it only receives another path for dlopen.  Run in its own process (the library binds the first
librccl it opens):   python tests/native_comm_threads.py WORLD  -> prints one JSON line."""
import json
import os
import sys
import threading

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import golden  # noqa: E402
from pydca_amd import _lib, parallel  # noqa: E402

FAKE = os.path.join(ROOT, "tests", "fake_rccl", "libfake_rccl.so")


def main():
    world = int(sys.argv[1])
    G = golden("plm_rf71")
    X, q = G["X"], int(G["q"])
    iters = 10
    full = _lib.Context(0, _lib.DCA_F64)
    full.set_msa(X, q)
    w = full.compute_weights(0.8, _lib.DCA_F64)
    counts = full.weight_counts()
    full.plm_configure(1.0, 20.0)
    full.plm_init_x()
    x0 = full.plm_get_x(np.float64)
    fx_ref = full.plm_gradient()
    g_ref = full.plm_get_g(np.float64)
    full.plm_lbfgs_begin(iters)
    st_ref = full.plm_lbfgs_iterate(iters)
    x_ref = full.plm_get_x(np.float64)
    full.close()
    # the shipped float32 path, unsharded, for the float32 run of the column-strip decomposition
    full32 = _lib.Context(0, _lib.DCA_F32)
    full32.set_msa(X, q)
    full32.set_weight_counts(counts)
    full32.plm_configure(1.0, 20.0)
    full32.plm_set_x(x0.astype(np.float32))
    fx32_ref = full32.plm_gradient()
    g32_ref = full32.plm_get_g(np.float64)
    full32.close()
    M = golden("mf_toy_protein")
    XM = (M["X"] - 1).astype(np.uint8)

    uid_w, uid_p1, uid_p2, uid_p3, uid_m, uid_p4, uid_p5, uid_p6 = (_lib.comm_unique_id(FAKE) for _ in range(8))
    out = [None] * world

    def run(rank):
        try:
            res = {}
            # sequence weights: every rank counts 1/world of the tile pairs, one integer all-reduce
            c = _lib.Context(0, _lib.DCA_F64)
            c.set_msa(X, q)
            c.comm_init(uid_w, world, rank, FAKE)
            ws = c.compute_weights_sharded(0.8, _lib.DCA_F64)
            res["weights_equal"] = bool(np.array_equal(ws, w) and np.array_equal(c.weight_counts(), counts))
            c.close()
            for mode, uid in ((1, uid_p1), (2, uid_p2), (3, uid_p3)):
                s = parallel.make_sharded_plm_context(_lib, X, q, w, 1.0, 20.0, rank, world, 0, precision=64)
                s.comm_init(uid, world, rank, FAKE)
                s.plm_set_native_comm(mode)
                s.plm_set_x(x0)
                fx = s.plm_gradient()
                g = s.plm_get_g(np.float64)
                s.plm_lbfgs_begin(iters)
                st = s.plm_lbfgs_iterate(iters)
                x = s.plm_get_x(np.float64)
                res["mode%d" % mode] = dict(
                    fx_err=abs(fx - fx_ref) / abs(fx_ref), g_err=float(np.linalg.norm(g - g_ref) / np.linalg.norm(g_ref)),
                    status=[st.status, st.iterations, st.evaluations], fx_end_err=abs(st.fx - st_ref.fx) / abs(st_ref.fx),
                    x_err=float(np.linalg.norm(x - x_ref) / np.linalg.norm(x_ref)), x_sum=float(x.sum()))
                s.close()
            # what bench.py --gpus N does at start-up: ONE sharded context, a complete 4-iteration run under each scheme in
            # turn (the scheme can only change between runs), then the chosen scheme for the real run
            s = parallel.make_sharded_plm_context(_lib, X, q, w, 1.0, 20.0, rank, world, 0, precision=64)
            s.comm_init(uid_p6, world, rank, FAKE)
            seq = []
            for mode in (1, 2, 3, 2):
                s.plm_set_native_comm(mode)
                s.plm_set_x(x0)
                s.plm_lbfgs_begin(4)
                s.plm_lbfgs_iterate(1)
                st = s.plm_lbfgs_iterate(3)
                seq.append([st.status, st.iterations, st.finished, st.fx])
            s.plm_set_x(x0)
            s.plm_lbfgs_begin(iters)
            st = s.plm_lbfgs_iterate(iters)
            res["scheme_switching"] = dict(runs=seq, final=[st.status, st.iterations, st.evaluations], fx_end_err=abs(st.fx - st_ref.fx) / abs(st_ref.fx))
            s.close()
            # mode 4, the column-strip decomposition: every rank holds the WHOLE alignment and the columns of its sites
            if world > 1:
                s = _lib.Context(0, _lib.DCA_F64)
                s.set_msa(X, q)
                s.set_weights(w)
                s.comm_init(uid_p4, world, rank, FAKE)
                s.plm_configure_strips(1.0, 20.0)
                s.plm_set_x(x0)
                fx = s.plm_gradient()
                g = s.plm_get_g(np.float64)
                s.plm_lbfgs_begin(iters)
                st = s.plm_lbfgs_iterate(iters)
                x = s.plm_get_x(np.float64)
                sc = s.plm_scores(True)
                res["mode4"] = dict(
                    fx_err=abs(fx - fx_ref) / abs(fx_ref), g_err=float(np.linalg.norm(g - g_ref) / np.linalg.norm(g_ref)),
                    status=[st.status, st.iterations, st.evaluations], fx_end_err=abs(st.fx - st_ref.fx) / abs(st_ref.fx),
                    x_err=float(np.linalg.norm(x - x_ref) / np.linalg.norm(x_ref)), x_sum=float(x.sum()), score_sum=float(sc.sum()))
                s.close()
                s = _lib.Context(0, _lib.DCA_F32)
                s.set_msa(X, q)
                s.set_weight_counts(counts)
                s.comm_init(uid_p5, world, rank, FAKE)
                s.plm_configure_strips(1.0, 20.0)
                s.plm_set_x(x0.astype(np.float32))
                fx = s.plm_gradient()
                g = s.plm_get_g(np.float64)
                res["mode4_f32"] = dict(fx_err=abs(fx - fx32_ref) / abs(fx32_ref), g_err=float(np.linalg.norm(g - g32_ref) / np.linalg.norm(g32_ref)))
                s.close()
            # mfDCA pair counts summed through the communicator
            m = parallel.make_sharded_mf_context(_lib, XM, int(M["q"]), M["w"], rank, world, 0)
            m.comm_init(uid_m, world, rank, FAKE)
            fi_local = m.mf_single_site_freqs()          # a query BEFORE the reduction is switched on caches this shard's counts ...
            m.mf_set_native_comm(True)                   # ... which switching it on must drop (round-2 advisor finding)
            fi_global = m.mf_single_site_freqs()
            res["mf_stale_counts_dropped"] = bool(world == 1 or not np.array_equal(fi_local, fi_global))
            res["mf_fi_err"] = float(np.max(np.abs(fi_global - M["fi"])))
            m.set_weights(m.weights())                   # re-weighting keeps the exchange scheme (engines invalidated, not dropped)
            scores = m.mf_run(float(M["pseudocount"]), True)
            ranked = sorted(zip(scores, range(len(scores))), key=lambda t: (-t[0], t[1]))
            res["mf_err"] = float(np.max(np.abs(np.array([sc for sc, _ in ranked]) - M["apc_scores"]) / np.maximum(np.abs(M["apc_scores"]), 1e-3)))
            m.close()
            out[rank] = res
        except Exception as exc:      # pragma: no cover
            out[rank] = {"error": repr(exc)}

    threads = [threading.Thread(target=run, args=(r,)) for r in range(world)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=240)
    print(json.dumps({"world": world, "reference_status": [st_ref.status, st_ref.iterations, st_ref.evaluations], "ranks": out}), flush=True)
    os._exit(0)                       # a rank that failed leaves its peers inside a barrier


if __name__ == "__main__":
    main()
