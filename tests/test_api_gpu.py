"""GPU tests of the Python surface that mirrors the reference (PlmDCA / MeanFieldDCA classes,
msa_numerics module, command lines) and of the sharded evaluation."""
import os
import sys

import numpy as np
import pytest

from conftest import ROOT, data_file, golden, perturbed, rel_err

pytestmark = pytest.mark.gpu
sys.path.insert(0, ROOT)


def test_meanfield_class_full_ranking_rf00167():
    """mfdca compute_fn rna MSA_RF00167.fa --pseudocount 0.5 --apc (config 1 of BASELINE.json):
    identical ranked list as the real reference (golden), scores <= 1e-9."""
    from pydca_amd.meanfield_dca.meanfield_dca import MeanFieldDCA
    G = golden("mf_rf00167")
    inst = MeanFieldDCA(data_file("MSA_RF00167.fa"), "rna", pseudocount=0.5, seqid=0.8)
    assert (inst.num_sequences, inst.sequences_len, inst.num_site_states) == (2544, 102, 5)
    assert abs(inst.effective_num_sequences - 1608.8065456172653) < 1e-9
    apc = inst.compute_sorted_FN_APC()
    assert [p for p, _ in apc] == [tuple(p) for p in G["apc_pairs"]]
    np.testing.assert_allclose([s for _, s in apc], G["apc_scores"], rtol=1e-9)
    raw = inst.compute_sorted_FN()
    assert [p for p, _ in raw] == [tuple(p) for p in G["fn_pairs"]]


def test_meanfield_file_and_in_memory_alignment_agree():
    """The reference's own consistency test (tests/meanfield_dca_test.py:50-61): FN_APC from a
    file path equals FN_APC from an in-memory alignment."""
    from pydca_amd.fasta_reader import fasta_reader
    from pydca_amd.meanfield_dca.meanfield_dca import MeanFieldDCA
    f = data_file("toy_protein.fa")
    a = MeanFieldDCA(f, "protein").compute_sorted_FN_APC()
    b = MeanFieldDCA(fasta_reader.get_alignment_from_fasta_file(f), "protein").compute_sorted_FN_APC()
    assert a == b


def test_meanfield_notebook_kat():
    """examples/pydca_demo.ipynb cell 10: published top-5 FN_APC on trimmed RF00167."""
    from pydca_amd.meanfield_dca.meanfield_dca import MeanFieldDCA
    K = golden("kat_notebook")
    top = MeanFieldDCA(data_file("MSA_RF00167_trimmed71.fa"), "rna", pseudocount=0.5, seqid=0.8).compute_sorted_FN_APC()[:5]
    assert [p for p, _ in top] == [tuple(p) for p in K["mf_pairs"]]
    np.testing.assert_allclose([s for _, s in top], K["mf_scores"], rtol=1e-11)


def test_msa_numerics_module_stage_functions():
    from pydca_amd.meanfield_dca import msa_numerics as mn
    G = golden("mf_toy_protein")
    X, q = G["X"], int(G["q"])
    L = X.shape[1]
    w = mn.compute_sequences_weight(alignment_data=X, seqid=0.8)
    np.testing.assert_array_equal(w, G["w"])
    fi = mn.compute_single_site_freqs(alignment_data=X, num_site_states=q, seqs_weight=w)
    np.testing.assert_allclose(fi, G["fi"], rtol=1e-13, atol=1e-16)
    reg_fi = mn.get_reg_single_site_freqs(single_site_freqs=fi, seqs_len=L, num_site_states=q, pseudocount=0.5)
    assert reg_fi is fi
    np.testing.assert_allclose(reg_fi, G["reg_fi"], rtol=1e-13)
    fij = mn.compute_pair_site_freqs(alignment_data=X, num_site_states=q, seqs_weight=w)
    reg_fij = mn.get_reg_pair_site_freqs(pair_site_freqs=fij, seqs_len=L, num_site_states=q, pseudocount=0.5)
    np.testing.assert_allclose(reg_fij, G["reg_fij"], rtol=1e-12)
    corr = mn.construct_corr_mat(reg_fi=reg_fi, reg_fij=reg_fij, seqs_len=L, num_site_states=q)
    np.testing.assert_allclose(corr, G["corr_mat"], rtol=1e-11, atol=1e-15)
    np.testing.assert_allclose(mn.compute_couplings(corr_mat=corr), G["couplings"], rtol=1e-8, atol=1e-10)
    with pytest.raises(np.linalg.LinAlgError):
        mn.compute_couplings(corr_mat=np.zeros((8, 8)))
    # the two helpers of the module that the classes do not call: the serial twin of the pair counts (msa_numerics.py:128-179)
    # and the zero-padded block cut (:346-374; plmdca twin plmdca/msa_numerics.py:128-152)
    fij2 = mn.compute_pair_site_freqs_serial(alignment_data=X, num_site_states=q, seqs_weight=w)
    np.testing.assert_allclose(mn.get_reg_pair_site_freqs(pair_site_freqs=fij2, seqs_len=L, num_site_states=q, pseudocount=0.5),
                               G["reg_fij"], rtol=1e-12)
    J = G["couplings"]
    blk = mn.slice_couplings(couplings=J, site_pair=(1, 3), num_site_states=q)
    assert blk.shape == (q, q) and not blk[q - 1].any() and not blk[:, q - 1].any()
    np.testing.assert_array_equal(blk[:q - 1, :q - 1], J[(q - 1):2 * (q - 1), 3 * (q - 1):4 * (q - 1)])
    from pydca_amd.plmdca import msa_numerics as pn
    flat = np.arange(L * (L - 1) // 2 * (q - 1) ** 2, dtype=np.float64)
    b2 = pn.slice_couplings(couplings=flat, site_pair=(1, 3), num_site_states=q, seqs_len=L)
    pair = (L - 1) + (3 - 1 - 1)                                   # (0,1)..(0,L-1), then (1,2), (1,3)
    np.testing.assert_array_equal(b2[:q - 1, :q - 1].reshape(-1), flat[pair * (q - 1) ** 2:(pair + 1) * (q - 1) ** 2])
    assert not b2[q - 1].any() and not b2[:, q - 1].any()


def test_plmdca_class_against_reference_run(oracle_mf):
    """PlmDCA.compute_sorted_FN_APC on the toy RNA alignment vs the reference's own 1-thread
    plmdcaBackend run held in the fixture (float32, chaotic last digits: P4 regime)."""
    from pydca_amd.plmdca.plmdca import PlmDCA
    G = golden("plm_toy_rna")
    L, q = int(G["L"]), int(G["q"])
    inst = PlmDCA(data_file("toy_rna.fa"), "rna", seqid=0.8, lambda_h=1.8, lambda_J=1.8, max_iterations=100)
    apc = inst.compute_sorted_FN_APC()
    ref = oracle_mf.sort_scores(oracle_mf.plm_fn(G["run_a"], L, q), L)
    assert len(apc) == L * (L - 1) // 2
    assert [p for p, _ in apc[:3]] == [p for p, _ in ref[:3]]
    d = dict(ref)
    assert max(abs(s - d[p]) / abs(d[p]) for p, s in apc[:L]) < 2e-2
    x = inst.get_fields_and_couplings_from_backend()
    assert x.dtype == np.float32 and x.size == L * q + L * (L - 1) // 2 * q * q
    assert inst.get_couplings_no_gap_state(x).size == L * (L - 1) // 2 * (q - 1) ** 2


def test_command_lines_write_reference_named_files(tmp_path):
    from pydca_amd import mfdca_main, plmdca_main
    out = plmdca_main.run_plm_dca(["compute_fn", "rna", data_file("toy_rna.fa"), "--max_iterations", "5", "--apc",
                                   "--output_dir", str(tmp_path / "p")])
    assert os.path.basename(out) == "PLMDCA_apc_fn_scores_toy_rna.txt"
    lines = [ln for ln in open(out).read().splitlines() if not ln.startswith("#")]
    assert len(lines) == 45 and len(lines[0].split()) == 3
    out = mfdca_main.run_meanfield_dca(["compute_fn", "rna", data_file("toy_rna.fa"), "--pseudocount", "0.5",
                                        "--output_dir", str(tmp_path / "m")])
    assert os.path.basename(out) == "MFDCA_raw_fn_scores_toy_rna.txt"
    G = golden("mf_toy_rna")
    first = [ln for ln in open(out).read().splitlines() if not ln.startswith("#")][0].split()
    assert (int(first[0]) - 1, int(first[1]) - 1) == tuple(G["fn_pairs"][0])
    assert abs(float(first[2]) - G["fn_scores"][0]) < 1e-9 * G["fn_scores"][0]


@pytest.mark.parametrize("world", [2, 3])
def test_sharded_contexts_sum_to_unsharded(oracle_plm, world):
    """In-process emulation of `world` ranks on one GPU: shard + halo contexts, sum of their
    (fx, g) in rank order == the unsharded evaluation (float64 kernels, reference semantics)."""
    from pydca_amd import _lib, parallel
    G = golden("plm_rf71")
    X, q, L = G["X"], int(G["q"]), int(G["L"])
    w = oracle_plm.weights(X, 0.8, np.float64)
    x = perturbed(oracle_plm.init_x(X, w, q), L, q)
    full = _lib.Context(0, _lib.DCA_F64)
    full.set_msa(X, q)
    full.set_weights(w)
    full.plm_configure(1.0, 20.0)
    full.plm_set_x(x)
    fx_full = full.plm_gradient()
    g_full = full.plm_get_g(np.float64)
    full.close()
    fx_sum, g_sum = 0.0, np.zeros_like(g_full)
    for rank in range(world):
        ctx = parallel.make_sharded_plm_context(_lib, X, q, w, 1.0, 20.0, rank, world, 0, precision=64)
        ctx.plm_set_x(x)
        fx_sum += ctx.plm_gradient()
        g_sum += ctx.plm_get_g(np.float64)
        ctx.close()
    assert abs(fx_sum - fx_full) <= 1e-11 * abs(fx_full)
    assert rel_err(g_sum, g_full) < 1e-11


@pytest.mark.parametrize("world", [2, 3])
def test_sharded_mf_counts_sum_to_unsharded(world):
    """mfDCA sequence sharding emulated on one GPU: every shard context counts its block with the
    global weights; a recording hook collects the partial counts, a second pass hands every shard
    the sum -- scores and frequencies must equal the unsharded run (<= 1e-12)."""
    import ctypes as C
    from pydca_amd import _lib, parallel
    M = golden("mf_rf71")
    X = (M["X"] - 1).astype(np.uint8)
    q, theta, w = int(M["q"]), float(M["pseudocount"]), M["w"]
    Lq = X.shape[1] * q
    full = _lib.Context(0, _lib.DCA_F64)
    full.set_msa(X, q)
    full.set_weights(w)
    ref_scores = full.mf_run(theta, True)
    ref_fi = full.mf_single_site_freqs()
    full.close()
    hip = C.CDLL("libamdhip64.so")
    hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
    total = np.zeros(Lq * Lq)
    meff = np.zeros(1)

    def recorder(dev, count, dtype, scalar_dev):
        part, m = np.zeros(count), np.zeros(1)
        assert dtype == 64 and count == Lq * Lq
        assert hip.hipMemcpy(part.ctypes.data, dev, count * 8, 2) == 0 and hip.hipMemcpy(m.ctypes.data, scalar_dev, 8, 2) == 0
        total[:] += part
        meff[:] += m
        return 0

    def replayer(dev, count, dtype, scalar_dev):
        assert hip.hipMemcpy(dev, total.ctypes.data, count * 8, 1) == 0 and hip.hipMemcpy(scalar_dev, meff.ctypes.data, 8, 1) == 0
        return 0

    for rank in range(world):
        ctx = parallel.make_sharded_mf_context(_lib, X, q, w, rank, world, 0)
        ctx.mf_set_reduce_hook(recorder)
        ctx.mf_single_site_freqs()
        ctx.close()
    assert abs(meff[0] - w.sum()) <= 1e-12 * w.sum()
    for rank in range(world):
        ctx = parallel.make_sharded_mf_context(_lib, X, q, w, rank, world, 0)
        ctx.mf_set_reduce_hook(replayer)
        np.testing.assert_allclose(ctx.mf_single_site_freqs(), ref_fi, rtol=1e-12, atol=1e-15)
        np.testing.assert_allclose(ctx.mf_run(theta, True), ref_scores, rtol=1e-9)
        ctx.close()


def test_reduce_hook_through_torch_distributed():
    """The all-reduce hook path (torch.distributed, backend nccl = RCCL) with a 1-rank group:
    the gradient must come back unchanged and the optimiser must still run."""
    import torch
    import torch.distributed as dist
    from pydca_amd import _lib, parallel
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29591")
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        G = golden("plm_toy_protein")
        X, q = G["X"], int(G["q"])
        ctx = _lib.Context(0, _lib.DCA_F32)
        ctx.set_msa(X, q)
        ctx.compute_weights(0.8, _lib.DCA_F32)
        ctx.plm_configure(1.0, 5.0)
        ctx.plm_init_x()
        fx0 = ctx.plm_gradient()
        g0 = ctx.plm_get_g(np.float32)
        hook = parallel.TorchAllReduceHook(0)
        ctx.plm_set_reduce_hook(hook)
        fx1 = ctx.plm_gradient()
        g1 = ctx.plm_get_g(np.float32)
        assert hook.calls == 1 and fx1 == fx0 and np.array_equal(g0, g1)
        assert hook.direct_calls == hook.calls, "the all-reduce should run in place on the library's buffer"
        ctx.plm_lbfgs_begin(5)
        st = ctx.plm_lbfgs_iterate(5)
        assert st.iterations == 5 and hook.calls == 1 + st.evaluations
        ctx.close()
        # sharded optimiser vectors through torch.distributed (1-rank group: the collectives are
        # identities, but reduce_scatter_tensor / all_gather_into_tensor / all_reduce all run on RCCL)
        ctx = _lib.Context(0, _lib.DCA_F32)
        ctx.set_msa(X, q)
        ctx.compute_weights(0.8, _lib.DCA_F32)
        ctx.plm_configure(1.0, 5.0)
        ctx.plm_init_x()
        vcomm = parallel.TorchVectorComm(0, 0, 1)
        ctx.plm_set_vector_sharding(0, 1, vcomm)
        assert ctx.plm_gradient() == fx0 and np.array_equal(ctx.plm_get_g(np.float32), g0)
        ctx.plm_lbfgs_begin(5)
        st2 = ctx.plm_lbfgs_iterate(5)
        assert (st2.status, st2.iterations, st2.evaluations, st2.fx) == (st.status, st.iterations, st.evaluations, st.fx)
        assert vcomm.calls[1] == 1 + st2.evaluations and vcomm.calls[2] >= st2.evaluations and vcomm.calls[0] > st2.iterations
        assert vcomm.direct_calls == sum(vcomm.calls)
        ctx.close()
        # the same hook on the mfDCA pair counts (float64 buffer of (L q)^2 counts + Meff)
        M = golden("mf_toy_protein")
        mctx = _lib.Context(0, _lib.DCA_F64)
        mctx.set_msa((M["X"] - 1).astype(np.uint8), int(M["q"]))
        mctx.compute_weights(float(M["seqid"]), _lib.DCA_F64)
        mhook = parallel.TorchAllReduceHook(0)
        mctx.mf_set_reduce_hook(mhook)
        scores = mctx.mf_run(float(M["pseudocount"]), True)
        assert mhook.calls == 1
        ranked = sorted(zip(scores, range(len(scores))), key=lambda t: (-t[0], t[1]))
        np.testing.assert_allclose([s for s, _ in ranked], M["apc_scores"], rtol=1e-9)
        mctx.close()
    finally:
        dist.destroy_process_group()


def test_an_optimisation_left_below_its_cap_can_be_abandoned():
    """dca_plm_lbfgs_iterate is resumable, so a caller may stop driving a run below its cap (bench.py does).  Such a run is
    never `finished`: the exchange scheme refuses to change under it -- until dca_plm_lbfgs_end, or until the communicator is
    given back (dca_comm_destroy ends the run itself instead of refusing for ever)."""
    from pydca_amd import _lib, parallel
    G = golden("plm_toy_protein")
    ctx = _lib.Context(0, _lib.DCA_F32)
    ctx.set_msa(G["X"], int(G["q"]))
    ctx.compute_weights(0.8, _lib.DCA_F32)
    parallel.init_native_comm(ctx, _lib, 0, 1)
    ctx.plm_configure(1.0, 5.0)
    ctx.plm_init_x()
    ctx.plm_set_native_comm(2)
    ctx.plm_lbfgs_begin(50)
    st = ctx.plm_lbfgs_iterate(2)
    assert st.iterations == 2 and not st.finished
    with pytest.raises(_lib.DcaBackendError):
        ctx.plm_set_native_comm(1)                     # in the middle of a run
    ctx.plm_lbfgs_end()
    ctx.plm_set_native_comm(1)
    ctx.plm_lbfgs_begin(50)
    assert ctx.plm_lbfgs_iterate(2).iterations == 2
    ctx.comm_destroy()                                 # ends the unfinished run and releases the communicator
    ctx.plm_lbfgs_begin(3)
    assert ctx.plm_lbfgs_iterate(3).iterations == 3    # the engine lives on, unsharded
    ctx.close()


def test_native_rccl_communicator_single_rank():
    """The library's own RCCL communicator (csrc/comm_rccl.cpp) with one rank: every collective runs on the context's
    stream and is an identity, so all-reduce mode, sharded-vector mode, the sharded weights and the mfDCA count
    reduction must reproduce the unsharded results bit for bit -- and no Python hook is involved."""
    from pydca_amd import _lib, parallel
    G = golden("plm_toy_protein")
    X, q = G["X"], int(G["q"])
    ref = _lib.Context(0, _lib.DCA_F32)
    ref.set_msa(X, q)
    w_ref = ref.compute_weights(0.8, _lib.DCA_F32)
    ref.plm_configure(1.0, 5.0)
    ref.plm_init_x()
    fx0 = ref.plm_gradient()
    g0 = ref.plm_get_g(np.float32)
    ref.plm_lbfgs_begin(6)
    st0 = ref.plm_lbfgs_iterate(6)
    x0 = ref.plm_get_x(np.float32)
    ref.close()
    for mode in (1, 2, 3):
        ctx = _lib.Context(0, _lib.DCA_F32)
        ctx.set_msa(X, q)
        with pytest.raises(_lib.DcaBackendError):
            ctx.compute_weights_sharded(0.8, _lib.DCA_F32)           # no communicator yet: loud
        parallel.init_native_comm(ctx, _lib, 0, 1)
        w = ctx.compute_weights_sharded(0.8, _lib.DCA_F32)
        assert np.array_equal(w, w_ref)
        ctx.plm_configure(1.0, 5.0)
        ctx.plm_init_x()
        ctx.plm_set_native_comm(mode)
        assert ctx.plm_gradient() == fx0 and np.array_equal(ctx.plm_get_g(np.float32), g0)
        ctx.plm_lbfgs_begin(6)
        st = ctx.plm_lbfgs_iterate(6)
        assert (st.status, st.iterations, st.evaluations, st.fx) == (st0.status, st0.iterations, st0.evaluations, st0.fx)
        assert np.array_equal(ctx.plm_get_x(np.float32), x0)
        ctx.plm_set_native_comm(0)
        ctx.comm_destroy()
        with pytest.raises(_lib.DcaBackendError):
            ctx.plm_set_native_comm(mode)                            # communicator gone
        ctx.close()
    # the column-strip decomposition with one rank is the whole window: the unsharded run again; without a communicator: loud
    ctx = _lib.Context(0, _lib.DCA_F32)
    ctx.set_msa(X, q)
    ctx.compute_weights(0.8, _lib.DCA_F32)
    with pytest.raises(_lib.DcaBackendError):
        ctx.plm_configure_strips(1.0, 5.0)
    parallel.init_native_comm(ctx, _lib, 0, 1)
    assert ctx.comm_info() == (1, 0)
    ctx.plm_configure_strips(1.0, 5.0)
    ctx.plm_init_x()
    assert ctx.plm_gradient() == fx0 and np.array_equal(ctx.plm_get_g(np.float32), g0)
    ctx.plm_lbfgs_begin(6)
    st = ctx.plm_lbfgs_iterate(6)
    assert (st.status, st.iterations, st.evaluations, st.fx) == (st0.status, st0.iterations, st0.evaluations, st0.fx)
    assert np.array_equal(ctx.plm_get_x(np.float32), x0)
    ctx.close()
    M = golden("mf_toy_protein")
    mctx = _lib.Context(0, _lib.DCA_F64)
    mctx.set_msa((M["X"] - 1).astype(np.uint8), int(M["q"]))
    parallel.init_native_comm(mctx, _lib, 0, 1)
    mctx.compute_weights_sharded(float(M["seqid"]), _lib.DCA_F64)
    np.testing.assert_array_equal(mctx.weights(), M["w"])
    mctx.mf_set_native_comm(True)
    scores = mctx.mf_run(float(M["pseudocount"]), True)
    ranked = sorted(zip(scores, range(len(scores))), key=lambda t: (-t[0], t[1]))
    np.testing.assert_allclose([s for s, _ in ranked], M["apc_scores"], rtol=1e-9, atol=1e-12)
    mctx.close()


@pytest.mark.parametrize("parts", [2, 3, 8])
def test_weights_partial_counts_sum_to_full(parts):
    """the identity comparisons divided over `parts` ranks (every parts-th tile pair of the upper triangle);
    the integer partial counts sum to the counts of the unsharded kernel, and set_weight_counts gives the same weights."""
    from pydca_amd import _lib
    sys.path.insert(0, ROOT)
    from tools.gen_msa import dedup, generate
    X = dedup(generate(70, 3000, 21, 3))
    ctx = _lib.Context(0, _lib.DCA_F32)
    ctx.set_msa(X, 21)
    w = ctx.compute_weights(0.8, _lib.DCA_F32)
    full = ctx.weight_counts()
    total = np.zeros_like(full, dtype=np.uint64)
    for r in range(parts):
        part = ctx.weights_partial_counts(0.8, _lib.DCA_F32, r, parts)
        assert part.sum() < full.sum()
        total += part
    assert np.array_equal(total, full)
    ctx.set_weight_counts(total.astype(np.uint32))
    assert np.array_equal(ctx.weights(), w) and np.array_equal(ctx.weight_counts(), full)
    with pytest.raises(_lib.DcaBackendError):
        ctx.set_weight_counts(np.zeros_like(full))
    ctx.close()


def test_changed_weights_invalidate_derived_state():
    """Weights set after an engine was built must not be answered from the old weights: the engines are
    dropped, the mfDCA chain recomputes, the plmDCA engine asks for a new configure."""
    from pydca_amd import _lib
    M = golden("mf_toy_rna")
    X = (M["X"] - 1).astype(np.uint8)
    ctx = _lib.Context(0, _lib.DCA_F64)
    ctx.set_msa(X, 5)
    ctx.compute_weights(0.8, _lib.DCA_F64)
    s_w = ctx.mf_run(0.5, True)
    ctx.set_weights(np.ones(X.shape[0]))
    s_1 = ctx.mf_run(0.5, True)
    fresh = _lib.Context(0, _lib.DCA_F64)
    fresh.set_msa(X, 5)
    fresh.set_weights(np.ones(X.shape[0]))
    assert np.array_equal(s_1, fresh.mf_run(0.5, True)) and not np.array_equal(s_1, s_w)
    fresh.close()
    ctx.plm_configure(1.0, 1.0)
    ctx.plm_init_x()
    ctx.compute_weights(0.8, _lib.DCA_F64)
    with pytest.raises(_lib.DcaBackendError):
        ctx.plm_gradient()                                           # configure again after the weights changed
    ctx.close()


def test_direct_information_through_the_classes(tmp_path):
    """compute_sorted_DI[_APC] of both classes and the compute_di sub-commands."""
    from pydca_amd import mfdca_main, plmdca_main
    from pydca_amd.meanfield_dca.meanfield_dca import MeanFieldDCA
    from pydca_amd.plmdca.plmdca import PlmDCA
    G = golden("di_rf71")
    L = int(G["L"])
    f = data_file("MSA_RF00167_trimmed71.fa")
    di = MeanFieldDCA(f, "rna", pseudocount=0.5, seqid=0.8).compute_sorted_DI()
    order = np.argsort(-G["mf_di"], kind="stable")
    iu, ju = np.triu_indices(L, k=1)
    assert [p for p, _ in di[:L]] == [(int(iu[k]), int(ju[k])) for k in order[:L]]
    np.testing.assert_allclose([s for _, s in di[:L]], G["mf_di"][order[:L]], rtol=1e-7)
    # plmDCA: the regularised frequencies (python-reader alignment, pseudocount 0.5) are exact;
    # the scores inherit the optimiser's float32 trajectory (P4 regime, like compute_fn)
    T = golden("di_toy_rna")
    inst = PlmDCA(data_file("toy_rna.fa"), "rna", seqid=0.8, lambda_h=1.8, lambda_J=1.8, max_iterations=100)
    np.testing.assert_allclose(inst.get_reg_single_site_freqs(), T["plm_reg_fi"], rtol=1e-13)
    pdi = inst.compute_sorted_DI()
    ref_order = np.argsort(-T["plm_di"], kind="stable")
    iu, ju = np.triu_indices(int(T["L"]), k=1)
    assert [p for p, _ in pdi[:3]] == [(int(iu[k]), int(ju[k])) for k in ref_order[:3]]
    assert abs(pdi[0][1] - T["plm_di"][ref_order[0]]) < 2e-2 * T["plm_di"][ref_order[0]]
    assert len(inst.compute_sorted_DI_APC()) == 45
    # PlmDCA.compute_seqs_weight (plmdca.py:565-591) and compute_two_site_model_fields (:652-680) against the
    # reference's own output: weights of the Python reader's alignment, fields of the stored reference run
    np.testing.assert_array_equal(inst.compute_seqs_weight(), golden("mf_toy_rna")["w"])
    Lt, qt = int(T["L"]), int(T["q"])
    blocks = golden("plm_toy_rna")["run_a"][Lt * qt:].reshape(Lt * (Lt - 1) // 2, qt, qt)[:, :qt - 1, :qt - 1].reshape(-1)
    np.testing.assert_allclose(inst.compute_two_site_model_fields(blocks), T["plm_fields"], rtol=1e-10, atol=1e-13)
    out = mfdca_main.run_meanfield_dca(["compute_di", "rna", data_file("toy_rna.fa"), "--apc", "--output_dir",
                                        str(tmp_path / "m")])
    assert os.path.basename(out) == "MFDCA_apc_di_scores_toy_rna.txt"
    out = plmdca_main.run_plm_dca(["compute_di", "rna", data_file("toy_rna.fa"), "--max_iterations", "5",
                                   "--output_dir", str(tmp_path / "p")])
    assert os.path.basename(out) == "PLMDCA_raw_di_scores_toy_rna.txt"


@pytest.mark.parametrize("tag,fname,bio", [("toy_rna", "toy_rna.fa", "rna"), ("toy_protein", "toy_protein.fa", "protein"),
                                           ("rf71", "MSA_RF00167_trimmed71.fa", "rna")])
def test_meanfield_two_site_fields_and_di_dict_vs_reference(tag, fname, bio):
    """MeanFieldDCA.compute_two_site_model_fields(couplings, reg_fi) (meanfield_dca.py:556) and
    get_site_pair_di_score() (:793), called the way the reference's own get_site_pair_di_score chains them,
    against what the reference's class returned for the same calls (di_<tag>.npz: mf_fields, mf_di_dict_*)."""
    from pydca_amd.meanfield_dca.meanfield_dca import MeanFieldDCA
    G = golden("di_" + tag)
    inst = MeanFieldDCA(data_file(fname), bio, pseudocount=float(G["pseudocount"]), seqid=float(G["seqid"]))
    reg_fi = inst.get_reg_single_site_freqs()
    np.testing.assert_allclose(reg_fi, G["mf_reg_fi"], rtol=1e-13)
    couplings = inst.compute_couplings(inst.construct_corr_mat(reg_fi, inst.get_reg_pair_site_freqs()))
    fields = inst.compute_two_site_model_fields(couplings, reg_fi)
    assert fields.shape == G["mf_fields"].shape and fields.dtype == np.float64
    np.testing.assert_allclose(fields, G["mf_fields"], rtol=1e-7, atol=1e-12)
    d = inst.get_site_pair_di_score()
    assert isinstance(d, dict)
    assert [tuple(k) for k in d.keys()] == [tuple(int(v) for v in k) for k in G["mf_di_dict_keys"]]    # pair order kept
    np.testing.assert_allclose(np.array(list(d.values())), G["mf_di_dict_values"], rtol=1e-7, atol=1e-12)
    # and it is what compute_sorted_DI ranks
    ranked = inst.compute_sorted_DI()
    assert ranked[0][1] == max(d.values()) and d[ranked[0][0]] == ranked[0][1]


def test_compute_params_and_frequency_outputs(tmp_path, oracle_mf):
    """compute_params of both classes and the mfdca compute_params / compute_fi /
    compute_fij sub-commands: reference-named files, reference's own values."""
    from pydca_amd import mfdca_main, plmdca_main
    from pydca_amd.meanfield_dca.meanfield_dca import MeanFieldDCA, MeanFieldDCAException
    from pydca_amd.plmdca.plmdca import PlmDCA
    G = golden("params_toy_protein")
    f = data_file("toy_protein.fa")
    inst = MeanFieldDCA(f, "protein", pseudocount=0.5, seqid=0.8)
    fd = inst.compute_fields()
    assert rel_err(np.array([fd[i] for i in range(int(G["L"]))]), G["fields"]) <= 1e-8
    for name, kw in (("default", {}), ("fn_ld2_n5", dict(ranked_by="fn", linear_dist=2, num_site_pairs=5)),
                     ("diapc_ld1_n40", dict(ranked_by="DI_APC", linear_dist=1, num_site_pairs=40))):
        fields, couplings = inst.compute_params(**kw)
        assert [s for s, _ in fields] == list(G[name + "_field_sites"])
        assert [tuple(p) for p, _ in couplings] == [tuple(p) for p in G[name + "_pairs"]]
        assert rel_err(np.array([c for _, c in couplings]), G[name + "_couplings"]) <= 1e-7
    with pytest.raises(MeanFieldDCAException):
        inst.compute_params(ranked_by="xyz")
    # a caller-supplied couplings matrix goes through the host branch of compute_fields
    fd2 = inst.compute_fields(couplings=inst.get_couplings())
    assert rel_err(np.array([fd2[i] for i in range(int(G["L"]))]), G["fields"]) <= 1e-8

    p = PlmDCA(data_file("toy_rna.fa"), "rna", seqid=0.8, lambda_h=1.8, lambda_J=1.8, max_iterations=20)
    fields, couplings = p.compute_params(ranked_by="fn", linear_dist=2, num_site_pairs=4)
    x = p.get_fields_and_couplings_from_backend()
    ref_fields, ref_couplings = oracle_mf.plm_compute_params(x, oracle_mf.sort_scores(oracle_mf.plm_fn(x, 10, 5, apc_correct=False), 10),
                                                             10, 5, linear_dist=2, num_site_pairs=4)
    assert [pr for pr, _ in couplings] == [pr for pr, _ in ref_couplings]
    np.testing.assert_allclose(np.array([c for _, c in couplings]), np.array([c for _, c in ref_couplings]), rtol=1e-4, atol=1e-6)
    np.testing.assert_array_equal(np.array([h for _, h in fields]), np.array([h for _, h in ref_fields]))

    out = mfdca_main.run_meanfield_dca(["compute_params", "protein", f, "--output_dir", str(tmp_path / "m")])
    assert [os.path.basename(o) for o in out] == ["fields_toy_protein.txt", "couplings_toy_protein.txt"]
    rows = [ln for ln in open(out[0]).read().splitlines() if not ln.startswith("#")]
    assert len(rows) == 8 and len(rows[0].split(",")) == 21
    assert abs(float(rows[0].split(",")[1]) - G["fields"][0, 0]) <= 1e-7 * abs(G["fields"][0, 0])
    crow = [ln for ln in open(out[1]).read().splitlines() if not ln.startswith("#")][0].split(",")
    assert (int(crow[0]) - 1, int(crow[1]) - 1) == tuple(G["default_pairs"][0]) and len(crow) == 2 + 400
    M = golden("mf_toy_rna")
    out = mfdca_main.run_meanfield_dca(["compute_fi", "rna", data_file("toy_rna.fa"), "--output_dir", str(tmp_path / "f")])
    assert os.path.basename(out) == "fi_toy_rna.txt"
    rows = [ln.split(",") for ln in open(out).read().splitlines() if not ln.startswith("#")]
    assert len(rows) == 10 * 5 and abs(float(rows[7][2]) - M["reg_fi"][1, 2]) < 1e-12
    assert "# (1, 'A')(2, 'C')(3, 'G')(4, 'U')(5, '-')" in open(out).read()
    out = mfdca_main.run_meanfield_dca(["compute_fij", "rna", data_file("toy_rna.fa"), "--output_dir", str(tmp_path / "f")])
    rows = [ln.split(",") for ln in open(out).read().splitlines() if not ln.startswith("#")]
    assert len(rows) == 45 * 16 and abs(float(rows[16 + 5][4]) - M["reg_fij"][1, 1, 1]) < 1e-12
    out = plmdca_main.run_plm_dca(["compute_params", "rna", data_file("toy_rna.fa"), "--max_iterations", "5", "--ranked_by",
                                   "di", "--num_site_pairs", "3", "--output_dir", str(tmp_path / "p")])
    assert [os.path.basename(o) for o in out] == ["fields_toy_rna.txt", "couplings_toy_rna.txt"]
    assert len([ln for ln in open(out[1]).read().splitlines() if not ln.startswith("#")]) == 3


def test_refseq_backmapping_through_classes_and_cli(tmp_path):
    """--refseq_file: scores are filtered to the MSA columns that map to the reference
    sequence and renamed to its positions.  For RF00167 the mapping is columns 16..88 minus the
    template's gaps -> the mapped FN_APC list must be the unmapped one restricted and renamed."""
    from pydca_amd import mfdca_main
    from pydca_amd.meanfield_dca.meanfield_dca import MeanFieldDCA
    from pydca_amd.sequence_backmapper.sequence_backmapper import SequenceBackmapper
    f, r = data_file("MSA_RF00167.fa"), data_file("ref_RF00167.fa")
    inst = MeanFieldDCA(f, "rna", pseudocount=0.5, seqid=0.8)
    bm = SequenceBackmapper(alignment_data=inst.alignment, refseq_file=r, biomolecule="rna")
    mapping = bm.map_to_reference_sequence()
    assert len(mapping) == 71
    plain = inst.compute_sorted_FN_APC()
    mapped = inst.compute_sorted_FN_APC(seqbackmapper=bm)
    want = [((mapping[i], mapping[j]), s) for (i, j), s in plain if i in mapping and j in mapping]
    assert list(mapped) == want and len(mapped) == 71 * 70 // 2
    fields, couplings = inst.compute_params(seqbackmapper=bm, num_site_pairs=7)
    assert [s for s, _ in fields] == sorted(mapping.values()) and len(couplings) == 7
    inv = {v: k for k, v in mapping.items()}
    first_pair = couplings[0][0]
    raw = inst.compute_params(num_site_pairs=400)[1]
    d = {p: c for p, c in raw}
    np.testing.assert_allclose(couplings[0][1], d[(inv[first_pair[0]], inv[first_pair[1]])], rtol=1e-12)
    out = mfdca_main.run_meanfield_dca(["compute_fn", "rna", f, "--apc", "--refseq_file", r, "--output_dir", str(tmp_path / "m")])
    rows = [ln.split() for ln in open(out).read().splitlines() if not ln.startswith("#")]
    assert len(rows) == 71 * 70 // 2
    assert (int(rows[0][0]) - 1, int(rows[0][1]) - 1) == mapped[0][0]


@pytest.mark.parametrize("world", [2, 3])
def test_sharded_optimiser_vectors_with_thread_comm(oracle_plm, world):
    """Sequence sharding + sharded L-BFGS vectors (dca_plm_set_vector_sharding) exercised on one GPU:
    `world` threads, one context each, collectives through parallel.ThreadComm.  float64: the run must
    follow the unsharded optimiser (same status / iterations / evaluations, fx and x to rounding),
    every rank must end with the same full x, and a standalone gradient must come back complete."""
    import threading
    from pydca_amd import _lib, parallel
    G = golden("plm_rf71")
    X, q, L = G["X"], int(G["q"]), int(G["L"])
    w = oracle_plm.weights(X, 0.8, np.float64)
    x0 = oracle_plm.init_x(X, w, q).astype(np.float64)
    iters = 12
    full = _lib.Context(0, _lib.DCA_F64)
    full.set_msa(X, q)
    full.set_weights(w)
    full.plm_configure(1.0, 20.0)
    full.plm_set_x(x0)
    fx_ref = full.plm_gradient()
    g_ref = full.plm_get_g(np.float64)
    full.plm_lbfgs_begin(iters)
    st_ref = full.plm_lbfgs_iterate(iters)
    x_ref = full.plm_get_x(np.float64)
    full.close()

    comm = parallel.ThreadComm(world)
    out = [None] * world

    def run(rank):
        try:
            ctx = parallel.make_sharded_plm_context(_lib, X, q, w, 1.0, 20.0, rank, world, 0, precision=64)
            ctx.plm_set_vector_sharding(rank, world, comm.hook(rank))
            ctx.plm_set_x(x0)
            fx = ctx.plm_gradient()
            g = ctx.plm_get_g(np.float64)
            ctx.plm_lbfgs_begin(iters)
            st = ctx.plm_lbfgs_iterate(iters)
            out[rank] = (fx, g, st.status, st.iterations, st.evaluations, st.fx, ctx.plm_get_x(np.float64), ctx.plm_scores(True))
            ctx.close()
        except Exception as exc:      # pragma: no cover
            out[rank] = exc
            comm.barrier.abort()

    threads = [threading.Thread(target=run, args=(r,)) for r in range(world)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=300)
    for r in range(world):
        assert not isinstance(out[r], Exception) and out[r] is not None, out[r]
    for r in range(world):
        fx, g, status, its, evals, fx_end, x_end, scores = out[r]
        assert abs(fx - fx_ref) <= 1e-11 * abs(fx_ref) and rel_err(g, g_ref) < 1e-11
        assert (status, its, evals) == (st_ref.status, st_ref.iterations, st_ref.evaluations)
        assert abs(fx_end - st_ref.fx) <= 1e-9 * abs(st_ref.fx)
        assert rel_err(x_end, x_ref) < 1e-7
        assert np.array_equal(x_end, out[0][6]) and np.array_equal(scores, out[0][7])


@pytest.mark.parametrize("world", [2, 3, 8])
def test_native_exchange_path_with_several_ranks(world):
    """The product's native exchange path -- RCCL entry points enqueued on the context's stream, in place: all-reduce of
    g / fx, reduce-scatter + all-gather of the sharded optimiser vectors, the integer all-reduce of the sharded weights,
    the mfDCA count reduction -- with 2 and 3 ranks (threads) on one GPU.  librccl.so is replaced by the thread-level
    stand-in tests/fake_rccl (real RCCL refuses two ranks on one device); libdca_hip.so is the product build.  Every
    rank must follow the unsharded float64 run."""
    import json
    import subprocess
    fake = os.path.join(ROOT, "tests", "fake_rccl", "libfake_rccl.so")
    assert os.path.exists(fake), "build it: python -c 'import __graft_entry__ as g; g.build()'"
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "native_comm_threads.py"), str(world)], capture_output=True,
                       text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-3000:]
    res = json.loads(p.stdout.strip().splitlines()[-1])
    assert len(res["ranks"]) == world
    for r in res["ranks"]:
        assert r is not None and "error" not in r, r
        assert r["weights_equal"]
        for mode in ("mode1", "mode2", "mode3"):          # 3: direct exchange (grouped send / recv + rank-ordered local sum)
            m = r[mode]
            assert m["fx_err"] <= 1e-11 and m["g_err"] < 1e-11, m
            assert m["status"] == res["reference_status"], m
            assert m["fx_end_err"] <= 1e-9 and m["x_err"] < 1e-7, m
            assert m["x_sum"] == res["ranks"][0][mode]["x_sum"]           # every rank ends with the same x
        sw = r["scheme_switching"]                # bench.py's start-up sequence: a 4-iteration run per scheme on one context, then the real run
        assert all(run[1] == 4 and run[2] == 1 for run in sw["runs"]) and len({round(run[3], 3) for run in sw["runs"]}) == 1, sw
        assert sw["final"] == res["reference_status"] and sw["fx_end_err"] <= 1e-9, sw
        # mode 4, the column-strip decomposition (every rank the whole alignment and the columns of its sites; two grouped
        # point-to-point exchanges per evaluation): the same sums in the same order as the unsharded run, so it follows
        # it even more closely than the sequence-sharded schemes
        m = r["mode4"]
        assert m["fx_err"] <= 1e-13 and m["g_err"] < 1e-13, m
        assert m["status"] == res["reference_status"], m
        assert m["fx_end_err"] <= 1e-9 and m["x_err"] < 1e-7, m
        assert m["x_sum"] == res["ranks"][0]["mode4"]["x_sum"] and m["score_sum"] == res["ranks"][0]["mode4"]["score_sum"]
        assert r["mode4_f32"]["fx_err"] <= 1e-6 and r["mode4_f32"]["g_err"] < 1e-5, r["mode4_f32"]      # float32: the window changes the slab split
        assert r["mf_err"] < 1e-9
        assert r["mf_stale_counts_dropped"] and r["mf_fi_err"] < 1e-13, r      # reduction switched on after a query; re-weighted afterwards


@pytest.mark.parametrize("mode", ["vectors", "allreduce", "selflaunch"])
def test_bench_two_process_selftest(mode):
    """bench.py under torch.distributed.run with two ranks on ONE GPU (DCA_BENCH_SELFTEST=1: gloo
    instead of RCCL, which refuses two ranks per device).  Exercises the whole multi-process path --
    sharding, hooks, barriers, max-over-ranks timing -- and checks the optimiser follows the
    single-process run."""
    import json
    import subprocess
    env = dict(os.environ, DCA_BENCH_SELFTEST="1", MASTER_ADDR="127.0.0.1")
    if mode == "allreduce":
        env["DCA_BENCH_ALLREDUCE"] = "1"
    port = "29621" if mode == "vectors" else "29622"
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", port, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "4", "--warmup", "1", "--workload", "C"]
    if mode == "selflaunch":
        # plain `python bench.py --gpus 2` (no launcher, WORLD_SIZE unset) must start its two ranks itself -- round 3's
        # bench printed a 1-GPU line for it
        cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "4", "--warmup", "1", "--workload", "C"]
        for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
            env.pop(k, None)
    two = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert two.returncode == 0, two.stderr[-2000:]
    d2 = json.loads(two.stdout.strip().splitlines()[-1])
    one = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "4", "--warmup", "1", "--workload", "C",
                          "--no-cpu-baseline", "--no-mfdca", "--no-e2e"], capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert one.returncode == 0, one.stderr[-2000:]
    d1 = json.loads(one.stdout.strip().splitlines()[-1])
    assert d2["n_gpus"] == 2 and d2["steps"] == d1["steps"] == 4 and d2["lbfgs_state"] == d1["lbfgs_state"] == "running"
    assert abs(d2["fx"] - d1["fx"]) <= 1e-6 * abs(d1["fx"])
    assert d2["scaling"] == "strong" and d1["roofline"]["bound"] == "valu" and d1["roofline"]["hbm"]["bound"] == "hbm"
    assert "modes" in d1 and d1["modes"]["f64"]["iterations_per_s"] > 0


@pytest.mark.parametrize("scheme", ["timed", "4", "3"])
def test_bench_native_path_two_processes(scheme):
    """`python bench.py --gpus 2` through its NATIVE multi-rank path -- the path an 8-GPU node runs: self-launch of the ranks,
    the library's own communicators (sharded weights, one per context), the start-up timing of the four exchange schemes,
    the selection, the column-strip decomposition, the asserted rank counts -- on one GPU, with two PROCESSES over the
    stand-in tests/fake_rccl/libfake_rccl_mp.so (DCA_BENCH_SELFTEST=native; real RCCL refuses two ranks on one device).
    The optimiser must follow the single-process run whatever scheme runs."""
    import json
    import subprocess
    fake = os.path.join(ROOT, "tests", "fake_rccl", "libfake_rccl_mp.so")
    assert os.path.exists(fake), "build it: python -c 'import __graft_entry__ as g; g.build()'"
    env = dict(os.environ, DCA_BENCH_SELFTEST="native", DCA_RCCL_PATH=fake, MASTER_ADDR="127.0.0.1")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "DCA_BENCH_SCHEME"):
        env.pop(k, None)
    if scheme != "timed":
        env["DCA_BENCH_SCHEME"] = scheme
    two = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "4", "--warmup", "1", "--workload", "C"],
                         env=env, capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert two.returncode == 0, two.stderr[-3000:]
    d2 = json.loads(two.stdout.strip().splitlines()[-1])
    one = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "4", "--warmup", "1", "--workload", "C",
                          "--no-cpu-baseline", "--no-mfdca", "--no-e2e", "--no-modes"], capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert one.returncode == 0, one.stderr[-2000:]
    d1 = json.loads(one.stdout.strip().splitlines()[-1])
    sel = d2["comm_selection"]
    assert d2["n_gpus"] == 2 and sel["rccl_ranks"] == 2 and d2["steps"] == 4, d2
    if scheme == "timed":
        assert sorted(sel["ms_per_iteration"]) == ["1", "2", "3", "4"], sel          # all four schemes came up and were timed
        assert str(sel["chosen_mode"]) == min(sel["ms_per_iteration"], key=lambda m: sel["ms_per_iteration"][m])
    else:
        assert sel["chosen_mode"] == int(scheme)
    if sel["chosen_mode"] == 4:
        assert "column strips" in d2["config"]["parallelism"]
    assert abs(d2["fx"] - d1["fx"]) <= 1e-6 * abs(d1["fx"]), (d2["fx"], d1["fx"])
    # round 5: every rank's own time around the K iterations, the bytes each scheme puts on the wires, and the N = 1 point of the
    # same run (rank 0 alone after the timed region)
    assert len(d2["per_rank_ms_per_step"]) == 2 and all(t > 0 for t in d2["per_rank_ms_per_step"])
    assert abs(max(d2["per_rank_ms_per_step"]) - d2["ms_per_step"]) <= 1e-6 * d2["ms_per_step"]
    wb = sel["wire_bytes_per_rank_per_evaluation"]
    assert wb["1"] == wb["2"] == wb["3"] > wb["4"] > 0
    one_gpu = d2["single_gpu_same_run"]
    assert one_gpu["steps"] == d2["steps"] and one_gpu["iterations_per_s"] > 0
    assert abs(d2["speedup_vs_single_gpu_same_run"] - d2["value"] / one_gpu["iterations_per_s"]) < 1e-9


def test_msa_numerics_direct_information_functions():
    """Module-level DI functions of both msa_numerics mirrors against the
    reference's own output (goldens): two-site model fields and DI from caller-provided arrays."""
    from pydca_amd.meanfield_dca import msa_numerics as mf_num
    from pydca_amd.plmdca import msa_numerics as plm_num
    D, M, P = golden("di_toy_protein"), golden("mf_toy_protein"), golden("plm_toy_protein")
    L, q = int(D["L"]), int(D["q"])
    fields = mf_num.compute_two_site_model_fields(couplings=M["couplings"], reg_fi=M["reg_fi"], seqs_len=L, num_site_states=q)
    assert fields.shape == (L * (L - 1) // 2, 2, q) and np.allclose(fields.sum(axis=2), 1.0, atol=1e-12)
    di = mf_num.compute_direct_info(couplings=M["couplings"], fields_ij=fields, reg_fi=M["reg_fi"], seqs_len=L, num_site_states=q)
    np.testing.assert_allclose(di, D["mf_di"], rtol=1e-9, atol=1e-14)
    x = P["run_a"]
    blocks = x[L * q:].reshape(L * (L - 1) // 2, q, q)[:, :q - 1, :q - 1].reshape(-1)
    f2 = plm_num.compute_two_site_model_fields(couplings=blocks, reg_fi=D["plm_reg_fi"], seqs_len=L, num_site_states=q)
    np.testing.assert_allclose(f2, D["plm_fields"], rtol=1e-10, atol=1e-13)
    d2 = plm_num.compute_direct_info(couplings=blocks, fields_ij=f2, reg_fi=D["plm_reg_fi"], seqs_len=L, num_site_states=q)
    np.testing.assert_allclose(d2, D["plm_di"], rtol=1e-9, atol=1e-14)
    # a caller's own fields_ij is used as given, like the reference does (not silently recomputed)
    from oracle import mf as omf
    rng = np.random.default_rng(4)
    other = rng.random(f2.shape) + 0.1
    other /= other.sum(axis=2, keepdims=True)
    blk3 = blocks.reshape(-1, q - 1, q - 1).astype(np.float64)
    want = omf.direct_info(blk3, D["plm_reg_fi"], L, q, fields_ij=other)
    d3 = plm_num.compute_direct_info(couplings=blocks, fields_ij=other, reg_fi=D["plm_reg_fi"], seqs_len=L, num_site_states=q)
    np.testing.assert_allclose(d3, want, rtol=1e-10, atol=1e-14)
    assert not np.allclose(d3, d2, rtol=1e-3)
    d4 = mf_num.compute_direct_info(couplings=M["couplings"], fields_ij=other, reg_fi=M["reg_fi"], seqs_len=L, num_site_states=q)
    want4 = omf.direct_info(omf.mf_blocks(M["couplings"], L, q), M["reg_fi"], L, q, fields_ij=other)
    np.testing.assert_allclose(d4, want4, rtol=1e-10, atol=1e-14)
    X1 = M["X"]
    np.testing.assert_array_equal(plm_num.compute_sequences_weight(alignment_data=X1, sequence_identity=0.8), M["w"])
    with pytest.raises(ValueError):
        plm_num.compute_direct_info(couplings=blocks[:-1], reg_fi=D["plm_reg_fi"], seqs_len=L, num_site_states=q)


# ------------------------------------------------------------------------------------------------ devices= / --devices (row N1)
FAKE_MP = os.path.join(ROOT, "tests", "fake_rccl", "libfake_rccl_mp.so")


def _scores_of(path):
    rows = [ln.split() for ln in open(path).read().splitlines() if ln and not ln.startswith("#")]
    return [(int(r[0]), int(r[1])) for r in rows], np.array([float(r[2]) for r in rows])


def _cli(module, argv, env=None, timeout=900):
    import subprocess
    e = dict(os.environ, PYTHONPATH=ROOT, **(env or {}))
    p = subprocess.run([sys.executable, "-m", "pydca_amd." + module] + argv, env=e, capture_output=True, text=True, timeout=timeout, cwd=ROOT)
    assert p.returncode == 0, (p.stdout[-1500:], p.stderr[-3000:])
    return p


@pytest.mark.parametrize("msa,bio", [("MSA_RF00167_trimmed71.fa", "rna"), ("toy_protein.fa", "protein")])
def test_plmdca_command_line_devices_equals_one_gpu_float64(tmp_path, msa, bio):
    """Row N1: `plmdca compute_fn ... --devices 0,0` -- the calling process as rank 0 plus one helper process, the library's
    own communicators over the multi-process stand-in for librccl (two ranks on the one GPU of the box; on a node the same
    code runs over RCCL) -- writes the same ranked file as the one-GPU run.  float64 takes the column strips, whose
    gradient is bit-identical to the unsharded one: the same pairs in the same order, the scores equal to 1e-12 (the
    optimiser's dot products are summed per rank and then over the ranks -- in double-double, so usually to the last bit)."""
    assert os.path.exists(FAKE_MP), "build it: python -c 'import __graft_entry__ as g; g.build()'"
    common = ["compute_fn", bio, data_file(msa), "--max_iterations", "12", "--apc", "--precision", "64"]
    _cli("plmdca_main", common + ["--output_dir", str(tmp_path / "one")])
    _cli("plmdca_main", common + ["--output_dir", str(tmp_path / "two"), "--devices", "0,0"], env={"DCA_RCCL_PATH": FAKE_MP})
    name = "PLMDCA_apc_fn_scores_%s.txt" % os.path.splitext(msa)[0]
    p1, s1 = _scores_of(str(tmp_path / "one" / name))
    p2, s2 = _scores_of(str(tmp_path / "two" / name))
    assert p1 == p2
    np.testing.assert_allclose(s2, s1, rtol=1e-12, atol=1e-15)
    print("\n%s: identical order; %d of %d scores byte-identical" % (msa, int(np.sum(s1 == s2)), len(s1)))


def test_plmdca_devices_float32_scheme_choice_and_failures():
    """The default float32 mode with devices=[0, 0].  The exchange scheme is chosen deterministically (the column strips): two runs
    of the same command give byte-identical scores, and the result follows the one-GPU run (another order of float32 sums: the
    P4 regime, bounded loosely).  DCA_EXCHANGE_SCHEME=auto still brings all four schemes up, times them and runs the fastest.
    The exception type of a rank that fails before the collectives is the class's own; a rank that DIES after it reported
    'ready' ends in the same exception -- before the communicators are up (rank 0's worker thread, stuck in the set-up, is left
    behind after 15 s) as well as between two collectives (dca_comm_abort releases rank 0 at once) -- and the calling process
    lives on and runs the next job."""
    import subprocess
    code = (
        "import json, os, sys, numpy as np; sys.path.insert(0, %r)\n"
        "from pydca_amd.plmdca import plmdca\n"
        "f = %r\n"
        "a = plmdca.PlmDCA(f, 'rna', max_iterations=10)\n"
        "sa = a.compute_sorted_FN_APC()\n"
        "b = plmdca.PlmDCA(f, 'rna', max_iterations=10, devices=[0, 0])\n"
        "sb = b.compute_sorted_FN_APC()\n"
        "sel = b.last_status['multi_gpu']\n"
        "b2 = plmdca.PlmDCA(f, 'rna', max_iterations=10, devices=[0, 0])\n"
        "same = b2.compute_sorted_FN_APC() == sb\n"
        "os.environ['DCA_EXCHANGE_SCHEME'] = 'auto'\n"
        "c = plmdca.PlmDCA(f, 'rna', max_iterations=10, devices=[0, 0])\n"
        "sc = c.compute_sorted_FN_APC()\n"
        "auto = c.last_status['multi_gpu']\n"
        "del os.environ['DCA_EXCHANGE_SCHEME']\n"
        "da = dict(sa); top = [p for p, _ in sa[:20]]\n"
        "dev = max(abs(dict(sb)[p] - da[p]) / abs(da[p]) for p in top)\n"
        "dev_auto = max(abs(dict(sc)[p] - da[p]) / abs(da[p]) for p in top)\n"
        "try:\n"
        "    plmdca.PlmDCA(f, 'rna', max_iterations=2, devices=[0, 4097]).compute_sorted_FN()\n"
        "    err = 'no error'\n"
        "except plmdca.PlmDCAException as e:\n"
        "    err = 'PlmDCAException'\n"
        "os.environ['DCA_MULTI_GPU_TEST_DIE_AFTER_READY'] = '1'\n"
        "try:\n"
        "    plmdca.PlmDCA(f, 'rna', max_iterations=2, devices=[0, 0]).compute_sorted_FN()\n"
        "    died = 'no error'\n"
        "except plmdca.PlmDCAException as e:\n"
        "    died = 'PlmDCAException'\n"
        "del os.environ['DCA_MULTI_GPU_TEST_DIE_AFTER_READY']\n"
        "os.environ['DCA_MULTI_GPU_TEST_DIE_IN_RUN'] = '1'\n"
        "import time; t0 = time.time()\n"
        "try:\n"
        "    plmdca.PlmDCA(f, 'rna', max_iterations=2, devices=[0, 0]).compute_sorted_FN()\n"
        "    died2 = 'no error'\n"
        "except plmdca.PlmDCAException as e:\n"
        "    died2 = 'PlmDCAException'\n"
        "abort_s = time.time() - t0\n"
        "del os.environ['DCA_MULTI_GPU_TEST_DIE_IN_RUN']\n"
        "d = plmdca.PlmDCA(f, 'rna', max_iterations=10, devices=[0, 0])\n"
        "alive = d.compute_sorted_FN_APC() == sb\n"
        "print(json.dumps(dict(sel=sel, auto=auto, dev=dev, dev_auto=dev_auto, same=same, alive=alive, died=died, died2=died2, abort_s=abort_s,\n"
        "                      status=[a.last_status['status'], b.last_status['status']],\n"
        "                      its=[a.last_status['iterations'], b.last_status['iterations']], err=err)))\n" % (ROOT, data_file("MSA_RF00167_trimmed71.fa")))
    p = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=900, env=dict(os.environ, DCA_RCCL_PATH=FAKE_MP))
    assert p.returncode == 0, p.stderr[-3000:]
    import json
    d = json.loads(p.stdout.strip().splitlines()[-1])
    assert d["sel"]["chosen_scheme"] == 4 and d["sel"]["ms_per_iteration"] == {} and d["same"], d
    assert sorted(d["auto"]["ms_per_iteration"]) == ["1", "2", "3", "4"], d
    assert str(d["auto"]["chosen_scheme"]) == min(d["auto"]["ms_per_iteration"], key=lambda m: d["auto"]["ms_per_iteration"][m])
    assert d["sel"]["ranks"] == 2 and d["its"] == [10, 10] and d["status"][0] == d["status"][1], d
    assert d["dev"] < 1e-3 and d["dev_auto"] < 1e-3, d
    assert d["err"] == "PlmDCAException" and d["died"] == "PlmDCAException" and d["alive"], d
    # a death BETWEEN collectives (the communicators are up): the abort releases rank 0 at once, well inside the abandon time-out
    assert d["died2"] == "PlmDCAException" and d["abort_s"] < 10.0, d


def test_mfdca_command_line_devices_equals_one_gpu(tmp_path):
    """`mfdca compute_fn ... --devices 0,0`: weights with the comparisons divided over two ranks, each rank counting half of the
    sequences, ONE all-reduce of the raw pair counts, then the one-GPU chain on rank 0 -- same ranking, scores to 1e-10 (the
    count sums are split at the window boundary)."""
    common = ["compute_fn", "rna", data_file("MSA_RF00167.fa"), "--pseudocount", "0.5", "--apc"]
    _cli("mfdca_main", common + ["--output_dir", str(tmp_path / "one")])
    _cli("mfdca_main", common + ["--output_dir", str(tmp_path / "two"), "--devices", "0,0"], env={"DCA_RCCL_PATH": FAKE_MP})
    p1, s1 = _scores_of(str(tmp_path / "one" / "MFDCA_apc_fn_scores_MSA_RF00167.txt"))
    p2, s2 = _scores_of(str(tmp_path / "two" / "MFDCA_apc_fn_scores_MSA_RF00167.txt"))
    assert p1 == p2
    np.testing.assert_allclose(s2, s1, rtol=1e-10, atol=1e-14)


def test_set_msa_rejects_codes_outside_the_alphabet():
    """The range check of dca_set_msa runs on the device (fused into the row-padding kernel); the error names the first bad element."""
    from pydca_amd import _lib as L_
    rng = np.random.default_rng(2)
    X = rng.integers(0, 5, size=(300, 77), dtype=np.uint8)
    ctx = L_.Context(0, L_.DCA_F32)
    ctx.set_msa(X, 5)
    bad = X.copy()
    bad[123, 45] = 5
    with pytest.raises(L_.DcaBackendError) as ei:
        ctx.set_msa(bad, 5)
    assert ei.value.code == L_.DCA_ERR_ARG and "element %d" % (123 * 77 + 45) in str(ei.value)
    ctx.set_msa(X, 5)                     # the context is usable again
    assert ctx.compute_weights(0.8).shape == (300,)
    ctx.close()


def test_plm_release_frees_the_engine_and_keeps_the_context():
    """dca_plm_release: the plmDCA tables and vectors go (a multi-GPU rank configures its whole-alignment context only for the initial
    point), alignment and weights stay, a second configure gives the same initial point, releasing twice is harmless."""
    from pydca_amd import _lib as L_
    rng = np.random.default_rng(3)
    X = rng.integers(0, 21, size=(400, 60), dtype=np.uint8)
    ctx = L_.Context(0, L_.DCA_F32)
    ctx.set_msa(X, 21)
    w = ctx.compute_weights(0.8)
    ctx.plm_configure(1.0, 20.0)
    ctx.plm_init_x()
    x0 = ctx.plm_get_x()
    ctx.plm_release()
    ctx.plm_release()
    with pytest.raises(L_.DcaBackendError) as ei:
        ctx.plm_get_x()
    assert ei.value.code == L_.DCA_ERR_STATE
    np.testing.assert_array_equal(ctx.weights(), w)
    ctx.plm_configure(1.0, 20.0)
    ctx.plm_init_x()
    np.testing.assert_array_equal(ctx.plm_get_x(), x0)
    fx = ctx.plm_gradient()
    assert np.isfinite(fx)
    ctx.close()


def test_dca_plm_run_one_call_entry():
    """SURVEY section 8 b1's richer entry, `int dca_plm_run(const dca_plm_args*, x_out, dtype, dca_plm_stats*)`: through ctypes as a C
    host would call it.  One device: the bytes and the status / iterations / evaluations of the stage API.  devices = {0, 0} and
    {0, 0, 0}: ranks as host threads over the stand-in librccl -- the float64 result of one GPU (the optimiser's dot products are
    summed per rank and then over the ranks in double-double: usually to the last bit, bounded at 1e-12).  From a file in
    float32: the bytes of the drop-in symbol plmdcaBackend.  Failing ranks and missing files are error codes, not hangs."""
    import json
    import subprocess
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "plm_run_threads.py")], capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stderr[-3000:]
    d = json.loads(p.stdout.strip().splitlines()[-1])
    print("\n", d)
    assert d["one_device_bytes_equal"] and d["one_device_stats"], d
    assert d["two_ranks_stats"] and d["two_ranks_max_rel"] < 1e-12 and d["three_ranks_max_rel"] < 1e-12, d
    assert d["file_float32_equals_dropin"], d
    assert d["bad_device"] != "no error" and d["no_file"][0] == -2, d
