"""Pins the CPU oracle (oracle/) against fixtures produced by the REAL reference
(tests/golden/make_golden.py) and against the values published in the reference's
notebook.  CPU only."""
import numpy as np
import pytest

from conftest import data_file, golden, perturbed, rel_err

PLM_CASES = [
    # tag, file, biomolecule, full arrays?
    ("toy_rna", "toy_rna.fa", 2, True),
    ("toy_protein", "toy_protein.fa", 1, True),
    ("rf71", "MSA_RF00167_trimmed71.fa", 2, True),
    ("rf00167", "MSA_RF00167.fa", 2, False),
    ("pf02826", "PF02826.faa", 1, False),       # the reference's own protein test input (q = 21, L = 195, N' = 2012)
]


@pytest.mark.parametrize("tag,fname,bio,full", PLM_CASES)
def test_plm_reader_weights_init_match_reference(oracle_plm, tag, fname, bio, full):
    G = golden("plm_" + tag)
    L, q = int(G["L"]), int(G["q"])
    X, raw = oracle_plm.read_msa(data_file(fname), bio, L)
    assert raw == int(G["raw_count"])
    assert X.shape == G["X"].shape and np.array_equal(X, G["X"])          # bit-exact
    w = oracle_plm.weights(X, float(G["seqid"]), np.float32)
    assert np.array_equal(w, G["w"])                                       # bit-exact
    x0 = oracle_plm.init_x(X, w, q)
    ref_h = G["x0"][:L * q] if full else G["h0"]
    np.testing.assert_allclose(x0[:L * q], ref_h, rtol=2e-6, atol=2e-6)
    assert not x0[L * q:].any()


@pytest.mark.parametrize("tag,fname,bio,full", PLM_CASES)
def test_plm_gradient_matches_reference(oracle_plm, tag, fname, bio, full):
    """(x, fx, g) triples of PlmDCA::gradient (plmdca_numerics.cpp:436); the reference
    is float32, so 1e-5 relative is the bar."""
    G = golden("plm_" + tag)
    L, q = int(G["L"]), int(G["q"])
    X, w = G["X"], G["w"]
    lh, lJ = float(G["lambda_h"]), float(G["lambda_J"])
    x0 = oracle_plm.init_x(X, w, q)
    for x, fkey, gkey in ((x0, "fx0", "g0"), (perturbed(x0, L, q), "fx1", "g1")):
        fx, g = oracle_plm.gradient(X, w, q, lh, lJ, x, carry=True, threads=4)
        assert abs(fx - float(G[fkey])) <= 2e-6 * abs(float(G[fkey]))
        if full:
            assert rel_err(g, G[gkey]) < 1e-5
        else:
            idx = G["idx"]
            assert rel_err(g[idx], G[gkey + "_sub"]) < 1e-5
            assert abs(np.linalg.norm(g.astype(np.float64)) - float(G[gkey + "_norm"])) < 1e-5 * float(G[gkey + "_norm"])
        # float64 oracle agrees with the float32 reference to float32 accuracy
        fx64, g64 = oracle_plm.gradient(X, w.astype(np.float64), q, lh, lJ, x.astype(np.float64), carry=True, threads=4)
        assert abs(fx64 - float(G[fkey])) <= 5e-5 * abs(float(G[fkey]))   # float32 sequential sums in the reference
        ref_g = G[gkey] if full else G[gkey + "_sub"]
        assert rel_err(g64 if full else g64[G["idx"]], ref_g) < 2e-5


def test_plm_gradient_matches_reference_at_config_C(oracle_plm):
    """BASELINE.json's config C itself (tools/gen_msa.py seed 12345, L=200 N=10k q=21, lambda_h=1, lambda_J=50): the
    float64 oracle against the compiled reference's fx and sampled gradient elements at the initial and the perturbed
    point.  The reference is float32 with accumulation chains of N terms: its own distance from the float64 oracle was
    measured when the fixture was made (x0 1.8e-5, x1 1.8e-6) and is the bar here."""
    import os
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from tools.gen_msa import SEEDS, dedup, generate
    G = golden("config_C_reference")
    X = dedup(generate(200, 10000, 21, SEEDS["C"]))
    assert X.shape[0] == int(G["n_unique"])
    w = oracle_plm.weights(X, 0.8, np.float32)
    x0 = oracle_plm.init_x(X, w, 21)
    for name, x in (("x0", x0), ("x1", perturbed(x0, 200, 21))):
        fx, g = oracle_plm.gradient(X, w.astype(np.float64), 21, 1.0, 50.0, x.astype(np.float64), carry=True)
        ref_err = float(G[name + "_ref_err_vs_f64"].max())
        assert abs(fx - float(G[name + "_fx"])) <= 5e-6 * abs(fx)
        sub = g[::int(G["stride"])]
        assert rel_err(sub, G[name + "_g_sub"]) < 2.0 * ref_err + 1e-6, (name, rel_err(sub, G[name + "_g_sub"]), ref_err)
        assert abs(np.linalg.norm(g) - float(G[name + "_gnorm"])) <= 1e-5 * np.linalg.norm(g)


def test_canonical_float64_order_is_a_reordering_only(oracle_plm, tmp_path):
    """The float64 oracle fixes the order of its additions (ORACLE_CANONICAL_F64 in oracle/plm_oracle.c: coupling rows, field,
    carry; one rounded residual; compensated field sums).  Built WITHOUT that switch -- the reference's own order of
    operations, in float64 -- it gives the same objective and gradient up to float64 rounding, with and without carry-over."""
    import ctypes as C
    import os
    import subprocess
    here = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    so = str(tmp_path / "liboracle_plain.so")
    subprocess.check_call(["gcc", "-O3", "-fopenmp", "-fno-fast-math", "-ffp-contract=off", "-shared", "-fPIC", "-DORACLE_PLAIN_F64",
                           "-o", so, os.path.join(here, "oracle", "plm_oracle.c"), "-lm"])
    plain = C.CDLL(so)
    dp = np.ctypeslib.ndpointer(np.float64, flags="C_CONTIGUOUS")
    u8 = np.ctypeslib.ndpointer(np.uint8, flags="C_CONTIGUOUS")
    f = plain.oracle_gradient_f64
    f.restype = C.c_double
    f.argtypes = [u8, dp, C.c_int, C.c_int, C.c_int, C.c_double, C.c_double, dp, dp, C.c_int, C.c_int]
    for tag in ("toy_rna", "toy_protein", "rf71"):
        G = golden("plm_" + tag)
        L, q = int(G["L"]), int(G["q"])
        X = np.ascontiguousarray(G["X"])
        w = oracle_plm.weights(X, 0.8, np.float64)
        x = perturbed(oracle_plm.init_x(X, w, q), L, q)
        for carry in (1, 0):
            fx_c, g_c = oracle_plm.gradient(X, w, q, float(G["lambda_h"]), float(G["lambda_J"]), x, carry=bool(carry), threads=4)
            g_p = np.zeros_like(x)
            fx_p = f(X, w, X.shape[0], L, q, float(G["lambda_h"]), float(G["lambda_J"]), x, g_p, carry, 4)
            assert abs(fx_c - fx_p) <= 1e-13 * abs(fx_p), (tag, carry, fx_c, fx_p)
            assert rel_err(g_c, g_p) < 1e-12, (tag, carry, rel_err(g_c, g_p))
            assert not np.array_equal(g_c, g_p)            # it IS another order


def test_canonical_float64_blocks_are_a_reordering_only(oracle_plm, tmp_path):
    """Round 5: the float64 oracle's per-slot chains run over blocks of 16384 sequences whose sums are added in ascending
    order (ORACLE_CANONICAL_BLOCK).  Against a build with one block for everything (round 4's single chain) and against the
    reference's literal order (-DORACLE_PLAIN_F64): same objective bits, gradient equal up to float64 rounding -- and really
    another order once the alignment is deeper than one block; identical bits when it is not."""
    import ctypes as C
    import os
    import subprocess
    here = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    libs = {}
    for name, flag in (("chain", "-DORACLE_CANONICAL_BLOCK=1000000000"), ("plain", "-DORACLE_PLAIN_F64")):
        so = str(tmp_path / ("liboracle_%s.so" % name))
        subprocess.check_call(["gcc", "-O3", "-fopenmp", "-fno-fast-math", "-ffp-contract=off", "-shared", "-fPIC", flag,
                               "-o", so, os.path.join(here, "oracle", "plm_oracle.c"), "-lm"])
        f = C.CDLL(so).oracle_gradient_f64
        f.restype = C.c_double
        dp = np.ctypeslib.ndpointer(np.float64, flags="C_CONTIGUOUS")
        f.argtypes = [np.ctypeslib.ndpointer(np.uint8, flags="C_CONTIGUOUS"), dp, C.c_int, C.c_int, C.c_int, C.c_double,
                      C.c_double, dp, dp, C.c_int, C.c_int]
        libs[name] = f
    rng = np.random.default_rng(5)
    L, q = 9, 5
    for N, differs in ((36000, True), (16384, False), (18000, True)):
        X = np.ascontiguousarray(rng.integers(0, q, size=(N, L), dtype=np.uint8))
        w = np.ascontiguousarray(rng.uniform(0.1, 1.0, size=N))
        x = perturbed(oracle_plm.init_x(X, w, q), L, q)
        for carry in (1, 0):
            fx_b, g_b = oracle_plm.gradient(X, w, q, 0.3, 2.0, x, carry=bool(carry), threads=3)
            for name in ("chain", "plain"):
                g_o = np.zeros_like(x)
                fx_o = libs[name](X, w, N, L, q, 0.3, 2.0, x, g_o, carry, 3)
                assert abs(fx_b - fx_o) <= 1e-13 * abs(fx_o), (N, name, carry)
                assert rel_err(g_b, g_o) < 1e-12, (N, name, carry, rel_err(g_b, g_o))
                if name == "chain":
                    assert np.array_equal(g_b, g_o) == (not differs), (N, carry)
                    assert fx_b == fx_o


def test_carry_over_is_what_the_reference_does(oracle_plm):
    """without the carried-over probabilities the result is far off."""
    G = golden("plm_toy_rna")
    L, q = int(G["L"]), int(G["q"])
    x = perturbed(G["x0"], L, q)
    _, g_exact = oracle_plm.gradient(G["X"], G["w"], q, float(G["lambda_h"]), float(G["lambda_J"]), x, carry=False)
    assert rel_err(g_exact, G["g1"]) > 1e-2


@pytest.mark.parametrize("tag,threads", [("toy_rna", 1), ("toy_rna", 3), ("toy_protein", 1), ("toy_protein", 2), ("rf71", 4)])
def test_plm_lbfgs_float32_equals_reference_run_exactly(oracle_plm, tag, threads):
    """The restated optimiser (lbfgs.cpp:248-644, Moré-Thuente :815-1004, update_trial_interval :1128-1295) in
    float32 reproduces the reference's own 1-thread run BIT FOR BIT: exit status, iterations, evaluations, the
    per-iteration (fx, xnorm, gnorm, step) the backend's verbose mode prints (plmdcaBackend.cpp:137-146) and the
    final parameter vector.  The oracle's gradient sums in a fixed order, so its thread count does not matter."""
    R = golden("plm_runs")
    G = golden("plm_" + tag)
    L, q = int(G["L"]), int(G["q"])
    mit = int(R[tag + "_max_iterations"])
    res = oracle_plm.lbfgs(G["X"], G["w"], q, float(R[tag + "_lambda_h"]), float(R[tag + "_lambda_J"]), mit,
                           oracle_plm.init_x(G["X"], G["w"], q), threads=threads, trace_cap=mit)
    assert (res["status"], res["iterations"], res["evaluations"]) == \
        (int(R[tag + "_status"]), int(R[tag + "_iterations"]), int(R[tag + "_evaluations"]))
    assert np.array_equal(res["trace"], R[tag + "_trace"][:, :4])
    assert np.array_equal(res["x"], R[tag + "_x"])
    assert np.array_equal(R[tag + "_x"], G["run_a"])          # == what `plmdcaBackend` itself returned
    assert np.float32(res["fx"]) == R[tag + "_fx"]


def test_plm_notebook_kat_rf71(oracle_plm, oracle_mf):
    """examples/pydca_demo.ipynb cell 5: plmDCA FN_APC top-5 on trimmed RF00167
    (lambda_h=1, lambda_J=20, 500 iterations).  The reference's own 1-thread run is held
    and an 8-thread run are held in the fixture.  The reference is not reproducible
    run-to-run (thread-arrival-order float32 sums, SURVEY 0.2: top-L FN spread up to
    0.3 %), so the published digits are matched to 5e-3 with the identical top-5 order."""
    G = golden("plm_rf71")
    K = golden("kat_notebook")
    L, q = int(G["L"]), int(G["q"])
    iu, ju = np.triu_indices(L, k=1)
    for key in ("run_a", "run_b"):
        apc = oracle_mf.plm_fn(G[key].astype(np.float32), L, q, dtype=np.float32)
        order = np.argsort(-apc, kind="stable")[:5]
        got = [(int(iu[k]), int(ju[k])) for k in order]
        assert got == [tuple(p) for p in K["plm_pairs"]]
        np.testing.assert_allclose(apc[order], K["plm_scores"], rtol=5e-3)


MF_STAGE_CASES = ["toy_rna", "toy_protein", "toy_rna_theta02_seqid1"]


@pytest.mark.parametrize("tag", MF_STAGE_CASES)
def test_mf_stages_match_reference(oracle_mf, tag):
    G = golden("mf_" + tag)
    X, q = G["X"], int(G["q"])
    N, L = X.shape
    theta, seqid = float(G["pseudocount"]), float(G["seqid"])
    w = oracle_mf.compute_sequences_weight(X, seqid) if seqid < 1.0 else np.ones(N)
    np.testing.assert_array_equal(w, G["w"])
    fi = oracle_mf.compute_single_site_freqs(X, q, w)
    np.testing.assert_allclose(fi, G["fi"], rtol=1e-13, atol=1e-16)
    reg_fi = oracle_mf.get_reg_single_site_freqs(fi, L, q, theta)
    np.testing.assert_allclose(reg_fi, G["reg_fi"], rtol=1e-13)
    fij = oracle_mf.compute_pair_site_freqs(X, q, w)
    np.testing.assert_allclose(fij, G["fij"], rtol=1e-12, atol=1e-16)
    reg_fij = oracle_mf.get_reg_pair_site_freqs(fij, L, q, theta)
    np.testing.assert_allclose(reg_fij, G["reg_fij"], rtol=1e-12)
    corr = oracle_mf.construct_corr_mat(reg_fi, reg_fij, L, q)
    np.testing.assert_allclose(corr, G["corr_mat"], rtol=1e-11, atol=1e-15)
    J = oracle_mf.compute_couplings(corr)
    np.testing.assert_allclose(J, G["couplings"], rtol=1e-8, atol=1e-10)


@pytest.mark.parametrize("tag", MF_STAGE_CASES + ["rf71", "rf00167", "pf02826"])
def test_mf_scores_and_ranking_match_reference(oracle_mf, tag):
    G = golden("mf_" + tag)
    X, q = G["X"], int(G["q"])
    L = X.shape[1]
    for apc_flag, pk, sk in ((False, "fn_pairs", "fn_scores"), (True, "apc_pairs", "apc_scores")):
        scores, _ = oracle_mf.mfdca_fn(X, q, float(G["pseudocount"]), float(G["seqid"]), weights=G["w"],
                                       apc_correct=apc_flag)
        ranked = oracle_mf.sort_scores(scores, L)
        assert [p for p, _ in ranked] == [tuple(p) for p in G[pk]]          # identical full ranking
        np.testing.assert_allclose([s for _, s in ranked], G[sk], rtol=1e-9)


def test_mf_notebook_kat(oracle_mf):
    """examples/pydca_demo.ipynb cell 10 (published values)."""
    G = golden("mf_rf71")
    K = golden("kat_notebook")
    scores, _ = oracle_mf.mfdca_fn(G["X"], 5, 0.5, 0.8)
    ranked = oracle_mf.sort_scores(scores, G["X"].shape[1])[:5]
    assert [p for p, _ in ranked] == [tuple(p) for p in K["mf_pairs"]]
    np.testing.assert_allclose([s for _, s in ranked], K["mf_scores"], rtol=1e-11)


def test_reader_codes(oracle_plm):
    """plmdca_numerics.cpp:699-732: the RNA map holds all 26 letters; everything but ACGU -- 'T' too,
    :729 -- is the gap state."""
    L = oracle_plm.lib()
    assert L.oracle_residue_code(2, ord("T")) == 4 and L.oracle_residue_code(2, ord("t")) == 4
    assert L.oracle_residue_code(2, ord("u")) == 3 and L.oracle_residue_code(2, ord("N")) == 4
    assert L.oracle_residue_code(1, ord("X")) == 20 and L.oracle_residue_code(1, ord("y")) == 19
    assert L.oracle_residue_code(1, ord("*")) == -1 and L.oracle_residue_code(2, ord("*")) == -1


def reader_sweep_cases():
    G = golden("reader_sweep")
    return [str(c) for c in G["cases"]]


@pytest.mark.parametrize("case", reader_sweep_cases())
def test_reader_sweep_oracle(oracle_plm, case, tmp_path):
    """oracle_read_msa == PlmDCA::readSequencesFromFile (plmdca_numerics.cpp:685-767) on the alphabet
    sweep: rows array_equal where the reference returns, an error where it throws."""
    G = golden("reader_sweep")
    p = tmp_path / (case + ".fa")
    p.write_bytes(G[case + "_text"].tobytes())
    if bool(G[case + "_throws"]):
        with pytest.raises(RuntimeError):
            oracle_plm.read_msa(str(p), int(G[case + "_bio"]), int(G[case + "_L"]))
    else:
        X, raw = oracle_plm.read_msa(str(p), int(G[case + "_bio"]), int(G[case + "_L"]))
        assert np.array_equal(X, G[case + "_rows"])


DI_CASES = [("toy_rna", "toy_rna.fa", "RNA"), ("toy_protein", "toy_protein.fa", "PROTEIN"),
            ("rf71", "MSA_RF00167_trimmed71.fa", "RNA")]


@pytest.mark.parametrize("tag,fname,bio", DI_CASES)
def test_mf_direct_information_matches_reference(oracle_mf, tag, fname, bio):
    """DI / DI_APC of the numpy restatement against MeanFieldDCA.compute_sorted_DI[_APC]
    (meanfield_dca.py:793-899).  Tolerance 1e-9 relative on the score vector; the
    fixed point stops at the same iteration as the reference (tolerance 1e-4 on the
    change), so only summation order differs; the full ranking must be identical."""
    g = golden("di_" + tag)
    X = oracle_mf.letter2int(oracle_mf.read_fasta(data_file(fname)), bio)
    L, q = int(g["L"]), int(g["q"])
    di = oracle_mf.mfdca_di(X, q, float(g["pseudocount"]), float(g["seqid"]))
    assert rel_err(di, g["mf_di"]) <= 1e-9
    assert np.array_equal(np.argsort(-di, kind="stable"), np.argsort(-g["mf_di"], kind="stable"))
    di_apc = oracle_mf.apc(di, L)
    assert rel_err(di_apc, g["mf_di_apc"]) <= 1e-9


@pytest.mark.parametrize("tag,fname,bio", DI_CASES)
def test_plm_direct_information_matches_reference(oracle_mf, tag, fname, bio):
    """plmDCA DI (plmdca.py:683-720; plmdca/msa_numerics.py:156-311) on the stored reference run."""
    g = golden("di_" + tag)
    x = golden("plm_" + tag)["run_a"]
    L, q = int(g["L"]), int(g["q"])
    X = oracle_mf.letter2int(oracle_mf.read_fasta(data_file(fname)), bio)
    w = oracle_mf.compute_sequences_weight(X, float(g["plm_seqid"]))
    reg_fi = oracle_mf.get_reg_single_site_freqs(oracle_mf.compute_single_site_freqs(X, q, w), L, q, 0.5)
    assert np.max(np.abs(reg_fi - g["plm_reg_fi"])) <= 1e-13
    E, hi, hj = oracle_mf.two_site_model_fields(oracle_mf.plm_blocks(x, L, q).astype(np.float64), reg_fi, L, q)
    assert np.max(np.abs(hi - g["plm_fields"][:, 0, :])) <= 1e-12
    assert np.max(np.abs(hj - g["plm_fields"][:, 1, :])) <= 1e-12
    di = oracle_mf.plm_di(x, reg_fi, L, q)
    assert rel_err(di, g["plm_di"]) <= 1e-9
    assert np.array_equal(np.argsort(-di, kind="stable"), np.argsort(-g["plm_di"], kind="stable"))


@pytest.mark.parametrize("tag,fname,bio", DI_CASES[:2])
def test_mf_fields_and_params_match_reference(oracle_mf, tag, fname, bio):
    """compute_fields / compute_params restatement vs the reference's own output
    (meanfield_dca.py:588-752): fields <= 1e-9 relative, identical pair selection for three
    rankings / filters, shifted couplings <= 1e-8 relative."""
    g = golden("params_" + tag)
    X = oracle_mf.letter2int(oracle_mf.read_fasta(data_file(fname)), bio)
    L, q = int(g["L"]), int(g["q"])
    theta, seqid = float(g["pseudocount"]), float(g["seqid"])
    w = oracle_mf.compute_sequences_weight(X, seqid)
    fi = oracle_mf.get_reg_single_site_freqs(oracle_mf.compute_single_site_freqs(X, q, w), L, q, theta)
    fij = oracle_mf.get_reg_pair_site_freqs(oracle_mf.compute_pair_site_freqs(X, q, w), L, q, theta)
    J = oracle_mf.compute_couplings(oracle_mf.construct_corr_mat(fi, fij, L, q))
    assert rel_err(oracle_mf.compute_fields(J, fi, L, q), g["fields"]) <= 1e-9
    fn = oracle_mf.frobenius_from_blocks(oracle_mf.mf_blocks(J, L, q))
    di = oracle_mf.direct_info(oracle_mf.mf_blocks(J, L, q), fi, L, q)
    rankings = {"default": (oracle_mf.apc(fn, L), {}), "fn_ld2_n5": (fn, dict(linear_dist=2, num_site_pairs=5)),
                "diapc_ld1_n40": (oracle_mf.apc(di, L), dict(linear_dist=1, num_site_pairs=40))}
    for name, (scores, kw) in rankings.items():
        fields, couplings = oracle_mf.mf_compute_params(J, fi, oracle_mf.sort_scores(scores, L), L, q, **kw)
        assert [s for s, _ in fields] == list(g[name + "_field_sites"])
        assert rel_err(np.array([f for _, f in fields]), g[name + "_fields"]) <= 1e-9
        assert [tuple(p) for p, _ in couplings] == [tuple(p) for p in g[name + "_pairs"]]
        if len(couplings):
            assert rel_err(np.array([c for _, c in couplings]), g[name + "_couplings"]) <= 1e-8


def test_plm_compute_params_restatement(oracle_mf):
    """PlmDCA.compute_params (plmdca.py:345-434) on the stored reference run: float32 fields and
    shifted blocks, selection rule shared with the mfDCA variant (pinned above)."""
    P = golden("plm_toy_rna")
    L, q = int(P["L"]), int(P["q"])
    x = P["run_a"]
    ranked = oracle_mf.sort_scores(oracle_mf.plm_fn(x, L, q), L)
    fields, couplings = oracle_mf.plm_compute_params(x, ranked, L, q, linear_dist=2, num_site_pairs=4)
    assert len(fields) == L and fields[3][1].dtype == np.float32 and fields[3][1].shape == (q - 1,)
    assert np.array_equal(fields[3][1], x[3 * q:3 * q + q - 1])
    assert len(couplings) == 4 and all(abs(i - j) > 2 for (i, j), _ in couplings)
    blk = couplings[0][1].reshape(q - 1, q - 1)
    assert blk.dtype == np.float32
    assert abs(blk.sum(axis=0)).max() < 1e-5 and abs(blk.sum(axis=1)).max() < 1e-5
