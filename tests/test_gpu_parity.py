"""GPU parity tests: the HIP path (through the C-ABI, pydca_amd._lib.Context) against the
CPU oracle and the golden fixtures produced by the real reference.  Run on an MI355X:
    python -m pytest tests -m gpu -x -q
"""
import ctypes as C
import os
import sys

import numpy as np
import pytest

from conftest import ROOT, assert_scores_within, data_file, golden, perturbed, rel_err

pytestmark = pytest.mark.gpu

sys.path.insert(0, ROOT)


@pytest.fixture(scope="module")
def L_():
    from pydca_amd import _lib
    _lib.lib()
    return _lib


def make_ctx(L_, X, q, precision, seqid=0.8, cmp=None):
    ctx = L_.Context(0, precision)
    ctx.set_msa(X, q)
    ctx.compute_weights(seqid, cmp if cmp is not None else precision)
    return ctx


PLM_TAGS = ["toy_rna", "toy_protein", "rf71", "rf00167", "pf02826"]


@pytest.mark.parametrize("order", ["file", "variable"])
@pytest.mark.parametrize("tag", PLM_TAGS)
def test_weights_bit_exact(L_, oracle_plm, tag, order, monkeypatch):
    """plmdca_numerics.cpp:611-671 -- integer counts and float32 1/count, bit for bit; with the sites compared in file
    order and most variable first (what large alignments get by default): the counts do not depend on it."""
    monkeypatch.setenv("DCA_WEIGHTS_ORDER", order)
    G = golden("plm_" + tag)
    ctx = make_ctx(L_, G["X"], int(G["q"]), L_.DCA_F32, float(G["seqid"]))
    counts = ctx.weight_counts()
    ref_counts = np.rint(1.0 / G["w"].astype(np.float64)).astype(np.uint32)
    assert np.array_equal(counts, ref_counts)
    assert np.array_equal((1.0 / counts.astype(np.float32)).astype(np.float32), G["w"])
    # double-compare variant (meanfield_dca/msa_numerics.py:13-50) vs the oracle
    ctx64 = make_ctx(L_, G["X"], int(G["q"]), L_.DCA_F64, 0.8, L_.DCA_F64)
    w64 = oracle_plm.weights(G["X"], 0.8, np.float64)
    assert np.array_equal(ctx64.weights(), w64)
    ctx.close(); ctx64.close()


def test_weights_thresholds_and_ragged_sizes(L_, oracle_plm):
    """Sizes that are not multiples of the 64x64 tile / 128-site stage, several seqid values."""
    rng = np.random.default_rng(5)
    for (N, L, q) in [(1, 7, 5), (65, 129, 21), (200, 33, 5), (513, 130, 21)]:
        base = rng.integers(0, q, size=(max(1, N // 8), L), dtype=np.uint8)
        X = base[rng.integers(len(base), size=N)].copy()
        flip = rng.random((N, L)) < 0.2
        X[flip] = rng.integers(0, q, size=int(flip.sum()), dtype=np.uint8)
        for seqid in (0.5, 0.8, 0.9, 0.999):
            for order in ("file", "variable"):
                os.environ["DCA_WEIGHTS_ORDER"] = order
                try:
                    ctx = L_.Context(0, L_.DCA_F32)
                    ctx.set_msa(X, q)
                    w = ctx.compute_weights(seqid, L_.DCA_F32)
                finally:
                    del os.environ["DCA_WEIGHTS_ORDER"]
                assert np.array_equal(w.astype(np.float32), oracle_plm.weights(X, seqid, np.float32)), (N, L, q, seqid, order)
                ctx.close()


@pytest.mark.parametrize("tag", PLM_TAGS)
def test_init_x(L_, oracle_plm, tag):
    G = golden("plm_" + tag)
    L, q = int(G["L"]), int(G["q"])
    ctx = make_ctx(L_, G["X"], q, L_.DCA_F32, float(G["seqid"]))
    ctx.plm_configure(float(G["lambda_h"]), float(G["lambda_J"]))
    ctx.plm_init_x()
    x0 = ctx.plm_get_x(np.float32)
    ref = oracle_plm.init_x(G["X"], G["w"], q)
    np.testing.assert_allclose(x0, ref, rtol=2e-6, atol=2e-6)
    ctx.close()


@pytest.mark.parametrize("tag", PLM_TAGS)
@pytest.mark.parametrize("mode", ["chunked", "serial"])
def test_gradient_float32_vs_reference_and_oracle(L_, oracle_plm, tag, mode):
    """Fixed-x (fx, g) of PlmDCA::gradient (plmdca_numerics.cpp:436-607).  float32 kernels:
    <= 1e-5 relative against the reference's own float32 output (golden) and against the
    float64 oracle."""
    G = golden("plm_" + tag)
    L, q = int(G["L"]), int(G["q"])
    lh, lJ = float(G["lambda_h"]), float(G["lambda_J"])
    ctx = make_ctx(L_, G["X"], q, L_.DCA_F32, float(G["seqid"]))
    ctx.plm_configure(lh, lJ, L_.CARRY_CHUNKED if mode == "chunked" else L_.CARRY_SERIAL)
    x0 = oracle_plm.init_x(G["X"], G["w"], q)
    full = "g0" in G.files
    for x, fk, gk in ((x0, "fx0", "g0"), (perturbed(x0, L, q), "fx1", "g1")):
        ctx.plm_set_x(x)
        fx = ctx.plm_gradient()
        g = ctx.plm_get_g(np.float32)
        fx64, g64 = oracle_plm.gradient(G["X"], G["w"].astype(np.float64), q, lh, lJ, x.astype(np.float64), carry=True)
        assert abs(fx - fx64) <= 2e-6 * abs(fx64)
        assert rel_err(g, g64) < 1e-5
        if full:
            assert rel_err(g, G[gk]) < 1e-5
        else:
            assert rel_err(g[G["idx"]], G[gk + "_sub"]) < 1e-5
            assert abs(np.linalg.norm(g.astype(np.float64)) - float(G[gk + "_norm"])) < 1e-5 * float(G[gk + "_norm"])
        assert abs(fx - float(G[fk])) <= 5e-5 * abs(float(G[fk]))
    ctx.close()


@pytest.mark.parametrize("tag", ["toy_rna", "toy_protein", "rf71"])
def test_gradient_float64_vs_oracle(L_, oracle_plm, tag):
    """float64 kernels vs the float64 oracle AT THE LAST BITS.  Round 4: the float64 mode forms every sum in the oracle's
    order (coupling rows in ascending j, then field, then carry; one chain per (site, state, column) over the sequences in
    ascending order; (2 lambda x + view_i) + view_j) or order-independently (objective and field gradients in
    double-double against the oracle's compensated sums), and its chunked scan warms up for 80 steps, which makes it
    bit-identical to the serial chain.  What is left are the last-place differences of the two exp() implementations
    (device library / glibc): most gradient elements are EQUAL, the vector agrees to a few 1e-16 (1e-12 before)."""
    G = golden("plm_" + tag)
    L, q = int(G["L"]), int(G["q"])
    lh, lJ = float(G["lambda_h"]), float(G["lambda_J"])
    w64 = oracle_plm.weights(G["X"], 0.8, np.float64)
    x0 = oracle_plm.init_x(G["X"], w64, q)
    got = {}
    for name, x in (("x0", x0), ("x1", perturbed(x0, L, q))):
        for mode, carry in ((L_.CARRY_SERIAL, True), (L_.CARRY_CHUNKED, True), (L_.CARRY_EXACT, False)):
            ctx = make_ctx(L_, G["X"], q, L_.DCA_F64, 0.8, L_.DCA_F64)
            ctx.plm_configure(lh, lJ, mode)
            ctx.plm_set_x(x)
            fx = ctx.plm_gradient()
            g = ctx.plm_get_g(np.float64)
            fx_o, g_o = oracle_plm.gradient(G["X"], w64, q, lh, lJ, x, carry=carry)
            assert abs(fx - fx_o) <= 4e-16 * abs(fx_o), (name, mode, fx, fx_o)           # equal or one unit in the last place
            assert rel_err(g, g_o) < 5e-15, (name, mode, rel_err(g, g_o))
            assert np.mean(g != g_o) < 0.3, (name, mode, float(np.mean(g != g_o)))
            got[(name, mode)] = (fx, g)
            ctx.close()
        # the chunked scan (80 warm-up steps in float64) and the serial chain: the same bits
        assert got[(name, L_.CARRY_CHUNKED)][0] == got[(name, L_.CARRY_SERIAL)][0]
        assert np.array_equal(got[(name, L_.CARRY_CHUNKED)][1], got[(name, L_.CARRY_SERIAL)][1])


@pytest.mark.parametrize("q,L,N", [(5, 30, 36000), (21, 20, 50000)])
def test_float64_canonical_blocks_both_geometries(L_, oracle_plm, monkeypatch, tmp_path, q, L, N):
    """Round 5: the float64 mode's per-slot chains run over blocks of 16384 sequences whose sums are added in ascending order
    (the oracle's ORACLE_CANONICAL_BLOCK).  The device has two launch geometries for that order -- one workgroup per (strip,
    site group) that adds each finished block to the running sum in G, or one workgroup and slab per block with an ordered
    slab sum -- and picks by a cost model; forced one after the other they give the SAME BITS.  Against the oracle the
    gradient agrees to the last place or two (the two exp() implementations differ by an ulp, and with chains of thousands
    of addends most sums see such a term), so the ORDER is pinned by comparison: the device's blocked sums are equal to
    the blocked oracle's in far more elements than to a single-chain build of the same oracle, and the device forced to
    round 4's single chain (a forced split of one leaves the canonical order) the other way round."""
    import subprocess
    here = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    so = str(tmp_path / "liboracle_chain.so")
    subprocess.check_call(["gcc", "-O3", "-fopenmp", "-fno-fast-math", "-ffp-contract=off", "-shared", "-fPIC",
                           "-DORACLE_CANONICAL_BLOCK=1000000000", "-o", so, os.path.join(here, "oracle", "plm_oracle.c"), "-lm"])
    chain_f = C.CDLL(so).oracle_gradient_f64
    chain_f.restype = C.c_double
    dp = np.ctypeslib.ndpointer(np.float64, flags="C_CONTIGUOUS")
    chain_f.argtypes = [np.ctypeslib.ndpointer(np.uint8, flags="C_CONTIGUOUS"), dp, C.c_int, C.c_int, C.c_int, C.c_double, C.c_double,
                        dp, dp, C.c_int, C.c_int]
    rng = np.random.default_rng(50 + q)
    X = np.ascontiguousarray(rng.integers(0, q, size=(N, L), dtype=np.uint8))          # three / four canonical blocks
    w64 = np.ascontiguousarray(rng.uniform(0.05, 1.0, size=N))
    x = perturbed(oracle_plm.init_x(X, w64, q), L, q)
    fx_o, g_o = oracle_plm.gradient(X, w64, q, 0.7, 3.0, x, carry=True)
    g_c = np.zeros_like(x)
    chain_f(X, w64, N, L, q, 0.7, 3.0, x, g_c, 1, os.cpu_count())
    got = {}
    for name, env in (("one_workgroup", {"DCA_SCATTER_CANON": "1"}), ("slab_per_block", {"DCA_SCATTER_CANON": "2"}), ("picked", {}),
                      ("single_chain", {"DCA_SCATTER_SPLIT": "1"})):
        for k in ("DCA_SCATTER_CANON", "DCA_SCATTER_SPLIT"):
            monkeypatch.delenv(k, raising=False)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        ctx = L_.Context(0, L_.DCA_F64)
        ctx.set_msa(X, q)
        ctx.set_weights(w64)
        ctx.plm_configure(0.7, 3.0, L_.CARRY_CHUNKED)
        ctx.plm_set_x(x)
        fx = ctx.plm_gradient()
        got[name] = (fx, ctx.plm_get_g(np.float64))
        ctx.close()
    pairs_part = slice(L * q, None)                      # the coupling gradient: the sums whose order is at stake
    for name in ("one_workgroup", "slab_per_block", "picked"):
        fx, g = got[name]
        assert abs(fx - fx_o) <= 4e-16 * abs(fx_o), (name, fx, fx_o)
        assert rel_err(g, g_o) < 5e-15, (name, rel_err(g, g_o))
        assert np.array_equal(g, got["one_workgroup"][1]), name
        same, other = float(np.mean(g[pairs_part] != g_o[pairs_part])), float(np.mean(g[pairs_part] != g_c[pairs_part]))
        print("float64 canonical blocks q=%d %s: %.3f of the coupling gradient differs from the blocked oracle, %.3f from the single-chain oracle"
              % (q, name, same, other))
        assert same + 0.15 < other, (name, same, other)
    chain = got["single_chain"][1]
    assert rel_err(chain, g_o) < 1e-12
    same, other = float(np.mean(chain[pairs_part] != g_c[pairs_part])), float(np.mean(chain[pairs_part] != g_o[pairs_part]))
    assert same + 0.15 < other, ("single_chain", same, other)


def test_chunked_scan_equals_serial_chain(L_, oracle_plm):
    """The chunk-parallel scan with 40 warm-up steps must reproduce the strictly serial
    carry chain (start-up error <= 2^-40)."""
    G = golden("plm_rf71")
    L, q = int(G["L"]), int(G["q"])
    x = perturbed(oracle_plm.init_x(G["X"], G["w"], q), L, q)
    out = {}
    for name, mode, chunk in (("serial", L_.CARRY_SERIAL, 0), ("c128", L_.CARRY_CHUNKED, 128), ("c64", L_.CARRY_CHUNKED, 64),
                              ("c32", L_.CARRY_CHUNKED, 32), ("c16", L_.CARRY_CHUNKED, 16)):
        ctx = make_ctx(L_, G["X"], q, L_.DCA_F64, 0.8, L_.DCA_F64)
        ctx.plm_configure(1.0, 20.0, mode, chunk, 40)
        ctx.plm_set_x(x.astype(np.float64))
        fx = ctx.plm_gradient()
        out[name] = (fx, ctx.plm_get_g(np.float64))
        ctx.close()
    # chunks shorter than the warm-up (32, 16 < 40) reach back over several predecessors: the scan must not be in place
    for name in ("c128", "c64", "c32", "c16"):
        assert abs(out[name][0] - out["serial"][0]) <= 1e-12 * abs(out["serial"][0])
        assert rel_err(out[name][1], out["serial"][1]) < 1e-12


def test_gradient_synthetic_multi_tile(L_, oracle_plm):
    """A shape that spans several column tiles, sequence blocks and scatter chunks
    (L*q = 1008 -> 8 column tiles of 128; N = 1500 -> 3 logits blocks, 12 chunks)."""
    from tools.gen_msa import dedup, generate
    X = dedup(generate(48, 1500, 21, 99))
    q = 21
    w = oracle_plm.weights(X, 0.8, np.float32)
    x = perturbed(oracle_plm.init_x(X, w, q), X.shape[1], q)
    fx_o, g_o = oracle_plm.gradient(X, w.astype(np.float64), q, 1.0, 50.0, x.astype(np.float64), carry=True)
    for prec, tol in ((L_.DCA_F32, 1e-5), (L_.DCA_F64, 1e-6)):
        ctx = make_ctx(L_, X, q, prec, 0.8, L_.DCA_F32)
        ctx.plm_configure(1.0, 50.0)
        ctx.plm_set_x(x)
        fx = ctx.plm_gradient()
        g = ctx.plm_get_g(np.float64)
        assert abs(fx - fx_o) <= tol * abs(fx_o)
        assert rel_err(g, g_o) < tol
        ctx.close()
    # RNA-shaped: q = 5
    X = dedup(generate(70, 900, 5, 98))
    w = oracle_plm.weights(X, 0.8, np.float32)
    x = perturbed(oracle_plm.init_x(X, w, 5), X.shape[1], 5)
    fx_o, g_o = oracle_plm.gradient(X, w.astype(np.float64), 5, 13.8, 13.8, x.astype(np.float64), carry=True)
    ctx = make_ctx(L_, X, 5, L_.DCA_F32, 0.8)
    ctx.plm_configure(13.8, 13.8)
    ctx.plm_set_x(x)
    fx = ctx.plm_gradient()
    assert abs(fx - fx_o) <= 1e-5 * abs(fx_o)
    assert rel_err(ctx.plm_get_g(np.float64), g_o) < 1e-5
    ctx.close()


def _topL_same(a, b, L):
    return list(np.argsort(-a, kind="stable")[:L]) == list(np.argsort(-b, kind="stable")[:L])


_ORACLE_RUNS = {}


def _oracle_run(oracle_plm, tag, iters):
    """The float64 oracle's L-BFGS run of a golden alignment to the given cap (made once per module run)."""
    if (tag, iters) not in _ORACLE_RUNS:
        G = golden("plm_" + tag)
        q = int(G["q"])
        w64 = oracle_plm.weights(G["X"], 0.8, np.float64)
        _ORACLE_RUNS[(tag, iters)] = oracle_plm.lbfgs(G["X"], w64, q, float(G["lambda_h"]), float(G["lambda_J"]), iters,
                                                     oracle_plm.init_x(G["X"], w64, q), carry=True)
    return _ORACLE_RUNS[(tag, iters)]


# (the strictly serial chain walks the sequences one by one on a single wave per 64 sites: 100 iterations of it take about a
#  minute on the two larger alignments, so those run the shipped chunked scan only)
@pytest.mark.parametrize("tag,mode", [("toy_rna", "serial"), ("toy_rna", "chunked"), ("toy_protein", "serial"), ("toy_protein", "chunked"),
                                      ("rf71", "serial"), ("rf71", "chunked"), ("rf00167", "chunked"), ("pf02826", "chunked")])
def test_lbfgs_float64_matches_oracle_at_equal_iteration_cap(L_, oracle_plm, oracle_mf, tag, mode):
    """P3 of SURVEY 8c4 / north_star AT THE REFERENCE'S DEFAULT CAP (max_iterations = 100, plmdca.py:72): same
    restated optimiser, same semantics, same cap => same exit status / iterations / evaluations, FN and FN_APC
    within 1e-4 relative and identical top-L order (float64), with the strictly serial carry chain and with the
    chunk-parallel scan the product ships.  PF02826 (the reference's own protein test input) runs into the cap
    (-997); the small RNA sets stop earlier in the line search, at the same iteration on both sides."""
    iters = 100
    G = golden("plm_" + tag)
    L, q = int(G["L"]), int(G["q"])
    lh, lJ = float(G["lambda_h"]), float(G["lambda_J"])
    ref = _oracle_run(oracle_plm, tag, iters)
    ctx = make_ctx(L_, G["X"], q, L_.DCA_F64, 0.8, L_.DCA_F64)
    ctx.plm_configure(lh, lJ, L_.CARRY_SERIAL if mode == "serial" else L_.CARRY_CHUNKED)
    ctx.plm_init_x()
    ctx.plm_lbfgs_begin(iters)
    st = ctx.plm_lbfgs_iterate(iters)
    assert (st.status, st.iterations) == (ref["status"], ref["iterations"])
    assert st.evaluations == ref["evaluations"]
    fn_ref = oracle_mf.plm_fn(ref["x"], L, q, apc_correct=False)
    for apc in (False, True):
        s_gpu = ctx.plm_scores(apc)
        s_ref = oracle_mf.plm_fn(ref["x"], L, q, apc_correct=apc)
        assert_scores_within(s_gpu, s_ref, fn_ref, 1e-4, top=L)          # FN relative to itself, FN_APC relative to the pair's FN
        assert _topL_same(s_gpu, s_ref, L)
    ctx.close()


@pytest.mark.parametrize("tag", ["toy_rna", "toy_protein"])
def test_exact_gradient_mode_converges_to_the_oracles_minimiser(L_, oracle_plm, oracle_mf, tag):
    """in exact_gradient mode (no carry-over) g IS the gradient of fx, the L2-regularised objective is
    strictly convex and the minimiser unique, so P3 can also be checked at convergence, where trajectories no longer
    matter: device (float64, CARRY_EXACT) and oracle (carry = False) both stop with status 0 (|g| / max(1, |x|) <= 1e-3,
    plmdcaBackend.cpp:68-75) -- not necessarily at the same iteration -- and a second device run restarted from the
    oracle's point needs no further iteration; parameters and scores agree to the stopping tolerance, top-L identical."""
    G = golden("plm_" + tag)
    L, q = int(G["L"]), int(G["q"])
    lh, lJ = float(G["lambda_h"]), float(G["lambda_J"])
    w64 = oracle_plm.weights(G["X"], 0.8, np.float64)
    ref = oracle_plm.lbfgs(G["X"], w64, q, lh, lJ, 2000, oracle_plm.init_x(G["X"], w64, q), carry=False)
    assert ref["status"] == 0, ref["status"]
    ctx = make_ctx(L_, G["X"], q, L_.DCA_F64, 0.8, L_.DCA_F64)
    ctx.plm_configure(lh, lJ, L_.CARRY_EXACT)
    ctx.plm_init_x()
    ctx.plm_lbfgs_begin(2000)
    st = ctx.plm_lbfgs_iterate(2000)
    assert st.status == 0 and st.gnorm / max(1.0, st.xnorm) <= 1e-3
    # on these small alignments the two runs even take the same path (on trimmed-71 they stop at iterations 231 and 234,
    # 1e-3 apart as the stopping tolerance allows: not part of the suite, 90 s of oracle time)
    assert (st.iterations, st.evaluations) == (ref["iterations"], ref["evaluations"])
    x = ctx.plm_get_x(np.float64)
    assert rel_err(x, ref["x"]) < 1e-6
    assert abs(st.fx - ref["fx"]) <= 1e-10 * abs(ref["fx"])
    fn_ref = oracle_mf.plm_fn(ref["x"], L, q, apc_correct=False)
    for apc in (False, True):
        s_gpu = ctx.plm_scores(apc)
        s_ref = oracle_mf.plm_fn(ref["x"], L, q, apc_correct=apc)
        assert_scores_within(s_gpu, s_ref, fn_ref, 1e-4, top=L)
        assert _topL_same(s_gpu, s_ref, L)
    # the oracle's minimiser is a stationary point for the device too: the gradient there passes the same test
    ctx.plm_set_x(ref["x"])
    ctx.plm_gradient()
    g = ctx.plm_get_g(np.float64)
    assert np.linalg.norm(g) / max(1.0, np.linalg.norm(ref["x"])) <= 1e-3 * (1 + 1e-6)
    ctx.close()


def test_lbfgs_float32_default_path_scores_close(L_, oracle_plm, oracle_mf):
    """Fast mode (float32 storage, float64 reductions, chunked scan): report-level check
    against the float64 oracle at the same cap -- expected ~1e-3 (SURVEY 8c4, P4 regime)."""
    G = golden("plm_rf71")
    L, q = 71, 5
    w64 = oracle_plm.weights(G["X"], 0.8, np.float64)
    ref = oracle_plm.lbfgs(G["X"], w64, q, 1.0, 20.0, 40, oracle_plm.init_x(G["X"], w64, q), carry=True)
    ctx = make_ctx(L_, G["X"], q, L_.DCA_F32, 0.8)
    ctx.plm_configure(1.0, 20.0)
    ctx.plm_init_x()
    ctx.plm_lbfgs_begin(40)
    st = ctx.plm_lbfgs_iterate(40)
    assert st.iterations == 40 and st.status == -997
    s_gpu = ctx.plm_scores(True)
    s_ref = oracle_mf.plm_fn(ref["x"], L, q)
    top = np.argsort(-s_ref, kind="stable")[:L]
    assert np.max(np.abs(s_gpu[top] - s_ref[top]) / np.abs(s_ref[top])) < 2e-2
    assert len(set(top) & set(np.argsort(-s_gpu, kind="stable")[:L])) >= L - 3
    ctx.close()


@pytest.mark.parametrize("env", [{"DCA_CHOLINV_LEAF16": "0"}, {"DCA_CHOLINV_LEAF16": "0", "DCA_CHOLINV_LEAF_MFMA": "0"},
                                 {"DCA_CHOLINV_LEAF16": "0", "DCA_CHOLINV_LEAF_MFMA": "0", "DCA_CHOLINV_LEAF128": "0"},
                                 {"DCA_CHOLINV_LEAF16": "0", "DCA_CHOLINV_LEAF128": "0"}, {"DCA_CHOLINV_LEAF16": "64"}, {"DCA_CHOLINV_LEAF16": "256"}])
def test_spd_inverse_alternative_leaves(env):
    """The recursion's other leaf kernels (the four-column MFMA leaf of rounds 2 - 5, the register-block leaf, 64-only recursion, the
    16-column-step leaf up to 64 / 256 columns instead of 128; selected by environment variables that the library reads once,
    hence a subprocess) give the same inverse to 1e-11."""
    import subprocess
    code = (
        "import sys, numpy as np; sys.path.insert(0, %r)\n"
        "from pydca_amd import _lib\n"
        "worst = 0.0\n"
        "for n in (64, 128, 192, 320, 500):\n"
        "    rng = np.random.default_rng(n); B = rng.standard_normal((n, n + 8)); A = B @ B.T / n + 0.5 * np.diag(rng.random(n) + 0.5)\n"
        "    ctx = _lib.Context(0, _lib.DCA_F64); inv = ctx.spd_inverse(A); ctx.close(); ref = np.linalg.inv(A)\n"
        "    worst = max(worst, float(np.linalg.norm(inv - ref) / np.linalg.norm(ref))); assert np.array_equal(inv, inv.T)\n"
        "print(worst)\n" % ROOT)
    p = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300, env=dict(os.environ, **env))
    assert p.returncode == 0, p.stderr[-2000:]
    assert float(p.stdout.strip().splitlines()[-1]) < 1e-11


@pytest.mark.parametrize("env", [{"DCA_CHOLINV_PANEL": "128"}, {"DCA_CHOLINV_PANEL": "128", "DCA_CHOLINV_OVERLAP": "0"},
                                 {"DCA_CHOLINV_PANEL": "256", "DCA_CHOLINV_SIDE_CAP": "7"}, {"DCA_CHOLINV_PANEL": "128", "DCA_CHOLINV_BULK_KW": "2"},
                                 {"DCA_CHOLINV_PANEL": "128", "DCA_CHOLINV_STREAMK": "1", "DCA_CHOLINV_TRSM_SPLIT": "1"},
                                 {"DCA_CHOLINV_PANEL": "256", "DCA_CHOLINV_STREAMK": "1", "DCA_CHOLINV_SIDE_CAP": "12"}])
def test_spd_inverse_blocked_form_at_small_sizes(env):
    """Round 5: the look-ahead factorisation + separate triangular-inverse tree (cholinv_blocked) is what runs for n >= 5000;
    forced here onto small matrices (panels of 128 / 256 columns, split-k from 128 columns on, one stream, tiny launch caps,
    the eight-wave bulk kernel, and the two opt-in forms that were measured slower and stay off: the stream-K bulk products
    with their ordered fix-up, the panel's lower rows on the bulk stream) so that every branch of it -- ragged last panel,
    split-k with its reduction, band splitting, the event chain -- is compared with LAPACK."""
    import subprocess
    code = (
        "import sys, numpy as np; sys.path.insert(0, %r)\n"
        "from pydca_amd import _lib\n"
        "worst = 0.0\n"
        "for n in (320, 448, 500, 1000, 1472, 2100):\n"
        "    rng = np.random.default_rng(n); B = rng.standard_normal((n, n + 8)); A = B @ B.T / n + 0.5 * np.diag(rng.random(n) + 0.5)\n"
        "    ctx = _lib.Context(0, _lib.DCA_F64); inv = ctx.spd_inverse(A); ctx.close(); ref = np.linalg.inv(A)\n"
        "    worst = max(worst, float(np.linalg.norm(inv - ref) / np.linalg.norm(ref))); assert np.array_equal(inv, inv.T)\n"
        "A[7, 7] = -1.0\n"
        "ctx = _lib.Context(0, _lib.DCA_F64)\n"
        "try:\n"
        "    ctx.spd_inverse(A); raise SystemExit('an indefinite matrix was accepted')\n"
        "except _lib.DcaBackendError as e:\n"
        "    assert e.code == _lib.DCA_ERR_NOT_SPD, e\n"
        "print(worst)\n" % ROOT)
    full = dict(os.environ, DCA_SWEEP="0", DCA_CHOLINV_BLOCKED_MIN="0", DCA_CHOLINV_SPLITK_MIN="128", **env)
    p = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600, env=full)
    assert p.returncode == 0, p.stderr[-2000:]
    assert float(p.stdout.strip().splitlines()[-1]) < 1e-11


def test_spd_inverse_default_form_at_n_5056(L_):
    """n = 5056: above the size from which dca_spd_inverse_device takes the block sweep by itself (default settings)."""
    n = 5056
    rng = np.random.default_rng(n)
    B = rng.standard_normal((n, n + 8))
    A = B @ B.T / n + 0.5 * np.diag(rng.random(n) + 0.5)
    ctx = L_.Context(0, L_.DCA_F64)
    inv = ctx.spd_inverse(A)
    ctx.close()
    assert rel_err(inv, np.linalg.inv(A)) < 1e-11
    assert np.array_equal(inv, inv.T)


SWEEP_SMALL = {"DCA_SWEEP_MIN": "256", "DCA_SWEEP_PANEL": "128"}


@pytest.mark.parametrize("env", [{}, {"DCA_SWEEP_CAP": "8", "DCA_SWEEP_PRIO_CAP": "8"}, {"DCA_SWEEP_PER_CU": "1", "DCA_SWEEP_STAGES": "4"},
                                 {"DCA_SWEEP_PER_CU": "1", "DCA_SWEEP_STAGES": "3", "DCA_SWEEP_CAP": "16"}, {"DCA_SWEEP_PANEL": "256"},
                                 {"DCA_SWEEP_MASK": "240"}, {"DCA_SWEEP": "0", "DCA_CHOLINV_BLOCKED_MIN": "5000"},
                                 {"DCA_SWEEP_FACTOR_MAX_N": "0"}, {"DCA_SWEEP_FACTOR_MAX_N": "0", "DCA_SWEEP_COPY": "0"}, {"DCA_SWEEP_COPY": "1"},
                                 {"DCA_SWEEP_RESERVE": "4", "DCA_SWEEP_RESERVE_MAX_N": "100000"}])
def test_spd_inverse_block_sweep_at_small_sizes(env):
    """Round 6: the symmetric block sweep (cholinv_sweep) is what runs from n = 2560 on; forced here onto small matrices (panels of 128 /
    256 columns -- ragged last panel, last tile row of 64, a 64-column last panel --, tiny workgroup caps so that the tile
    hand-out wraps and steals across the XCD chunks, one workgroup per CU with three / four operand stages, CU-masked streams)
    and compared with LAPACK; an indefinite matrix must come back as DCA_ERR_NOT_SPD with the pivot of the Cholesky
    factorisation.  {DCA_SWEEP: 0} is the three-phase form on the same matrices; the sets after it: the next pivot block through
    P instead of from X (what n >= 4500 takes), with / without the copy kernel, and CUs reserved for the chain by the update kernel."""
    import subprocess
    code = (
        "import sys, numpy as np; sys.path.insert(0, %r)\n"
        "from pydca_amd import _lib\n"
        "worst = 0.0\n"
        "for n in (448, 500, 1000, 1088, 1472, 2100):\n"
        "    rng = np.random.default_rng(n); B = rng.standard_normal((n, n + 8)); A = B @ B.T / n + 0.5 * np.diag(rng.random(n) + 0.5)\n"
        "    ctx = _lib.Context(0, _lib.DCA_F64); inv = ctx.spd_inverse(A); ctx.close(); ref = np.linalg.inv(A)\n"
        "    worst = max(worst, float(np.linalg.norm(inv - ref) / np.linalg.norm(ref))); assert np.array_equal(inv, inv.T)\n"
        "A[700, 700] = -1.0\n"
        "ctx = _lib.Context(0, _lib.DCA_F64)\n"
        "try:\n"
        "    ctx.spd_inverse(A); raise SystemExit('an indefinite matrix was accepted')\n"
        "except _lib.DcaBackendError as e:\n"
        "    assert e.code == _lib.DCA_ERR_NOT_SPD and 'pivot 701' in str(e), e\n"
        "print(worst)\n" % ROOT)
    p = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600, env=dict(os.environ, **dict(SWEEP_SMALL, **env)))
    assert p.returncode == 0, p.stderr[-2000:]
    assert float(p.stdout.strip().splitlines()[-1]) < 1e-11


def test_fused_fx_sums_give_the_bits_of_the_separate_kernels():
    """Round 6: on one GPU the optimiser sums fx inside the two launches of the line search's dot products (vec_dot3_fx_kernel,
    vec_final_fx_kernel); DCA_PLM_FUSE_FX=0 keeps the four separate launches.  Same code over the same operands in the same order: x, fx
    and the counters after 12 iterations are equal to the last bit, in float32 and in float64, for q = 5 and q = 21."""
    import subprocess
    code = (
        "import sys, hashlib, numpy as np; sys.path.insert(0, %r)\n"
        "from pydca_amd import _lib\n"
        "for q, L, N, prec in ((5, 37, 900, _lib.DCA_F32), (21, 30, 700, _lib.DCA_F32), (5, 37, 900, _lib.DCA_F64), (21, 30, 700, _lib.DCA_F64)):\n"
        "    rng = np.random.default_rng(q * 100 + L); X = rng.integers(0, q, size=(N, L), dtype=np.uint8)\n"
        "    ctx = _lib.Context(0, prec); ctx.set_msa(X, q); ctx.compute_weights(0.8, prec); ctx.plm_configure(1.0, 10.0); ctx.plm_init_x()\n"
        "    ctx.plm_lbfgs_begin(100); st = ctx.plm_lbfgs_iterate(12)\n"
        "    x = ctx.plm_get_x(np.float64 if prec == _lib.DCA_F64 else np.float32)\n"
        "    print(q, prec, st.iterations, st.evaluations, repr(st.fx), hashlib.sha256(x.tobytes()).hexdigest())\n"
        "    ctx.close()\n" % ROOT)
    outs = []
    for fuse in ("1", "0"):
        p = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600, env=dict(os.environ, DCA_PLM_FUSE_FX=fuse))
        assert p.returncode == 0, p.stderr[-2000:]
        outs.append(p.stdout.strip().splitlines()[-4:])
    assert len(outs[0]) == 4 and outs[0] == outs[1], outs


def test_left_over_strips_in_the_main_scatter_launch_give_the_same_bits():
    """Round 6: the column strips left over after the full sets of eight ride behind the main strips' workgroups in ONE launch
    (DCA_SCATTER_MERGE=0: a launch of their own, after the main one).  Every workgroup does what it did before: fx and the gradient are
    equal to the last bit.  DCA_SCATTER_REM=1 forces the left-over launch wherever there are left-over strips (9 and 12 strips here)."""
    import subprocess
    code = (
        "import sys, hashlib, numpy as np; sys.path.insert(0, %r)\n"
        "from pydca_amd import _lib\n"
        "for q, L, N in ((21, 50, 3000), (5, 300, 4000), (21, 200, 2500)):\n"
        "    rng = np.random.default_rng(q * 100 + L); X = rng.integers(0, q, size=(N, L), dtype=np.uint8)\n"
        "    ctx = _lib.Context(0, _lib.DCA_F32); ctx.set_msa(X, q); ctx.compute_weights(0.8, _lib.DCA_F32); ctx.plm_configure(1.0, 10.0); ctx.plm_init_x()\n"
        "    ctx.plm_lbfgs_begin(100); st = ctx.plm_lbfgs_iterate(4)\n"
        "    fx = ctx.plm_gradient(); g = ctx.plm_get_g()\n"
        "    print(q, L, st.evaluations, repr(st.fx), repr(fx), hashlib.sha256(g.tobytes()).hexdigest())\n"
        "    ctx.close()\n" % ROOT)
    outs = []
    # the third run: the field fold as a launch of its own instead of behind the pair fold's workgroups (DCA_FOLD_MERGE=0)
    for env in ({}, {"DCA_SCATTER_MERGE": "0"}, {"DCA_FOLD_MERGE": "0"}):
        p = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600,
                           env=dict(os.environ, DCA_SCATTER_REM="1", **env))
        assert p.returncode == 0, p.stderr[-2000:]
        outs.append(p.stdout.strip().splitlines()[-3:])
    assert len(outs[0]) == 3 and outs[0] == outs[1] == outs[2], outs


def test_scores_kernel(L_, oracle_mf):
    """FN / FN_APC kernel vs plmdca.py:437-524 restated in numpy (float64)."""
    rng = np.random.default_rng(3)
    for (L, q) in ((9, 5), (12, 21)):
        X = rng.integers(0, q, size=(30, L), dtype=np.uint8)
        ctx = make_ctx(L_, X, q, L_.DCA_F64, 0.8, L_.DCA_F64)
        ctx.plm_configure(1.0, 1.0)
        x = rng.standard_normal(ctx.num_params())
        ctx.plm_set_x(x)
        for apc in (False, True):
            np.testing.assert_allclose(ctx.plm_scores(apc), oracle_mf.plm_fn(x, L, q, apc_correct=apc), rtol=1e-12)
        ctx.close()


DROPIN_TOY_BOUNDS = (1e-6, 2e-6)          # measured: 1.5e-7 on x, 2.8e-7 on FN


def test_dropin_plmdcaBackend_symbol(L_, oracle_plm, oracle_mf):
    """The reference's own FFI (plmdcaBackend.cpp:151-156), bound as plmdca.py:79-89 does."""
    G = golden("plm_toy_rna")
    L, q = int(G["L"]), int(G["q"])
    P = oracle_plm.num_params(L, q)
    lib = L_.lib()
    ptr = lib.plmdcaBackend(2, q, os.fsencode(data_file("toy_rna.fa")), L, 0.8, 1.8, 1.8, 100, 1, False)
    assert ptr, lib.dca_last_error()
    x = np.ctypeslib.as_array(C.cast(ptr, C.POINTER(C.c_float)), shape=(P,)).copy()
    lib.freeFieldsAndCouplings(ptr)
    fn_ref = oracle_mf.plm_fn(G["run_a"], L, q)
    fn_our = oracle_mf.plm_fn(x, L, q)
    # The float32 oracle reproduces the reference's own lbfgs() run on this alignment BIT FOR BIT (plm_runs.npz, exit -1001 at
    # iteration 15: tests/test_oracle_golden.py), so the reference's run is one definite vector here and the device's float32
    # run is compared with it directly.  The device adds the same float32 terms in another order (chunk-parallel scan,
    # tiled gather sums), which the 15 iterations and the line search's exit amplify; DROPIN_TOY_BOUNDS is what was
    # measured on MI355X with a margin of 4 (printed below so that a drift shows up in the log before it fails).
    runs = golden("plm_runs")
    assert np.array_equal(G["run_a"], runs["toy_rna_x"]) or rel_err(G["run_a"], runs["toy_rna_x"]) < 2e-2    # as-run vs recorded run (thread order)
    dx, dfn = rel_err(x, runs["toy_rna_x"]), rel_err(fn_our, oracle_mf.plm_fn(runs["toy_rna_x"], L, q))
    print("\nplmdcaBackend on toy_rna against the reference's recorded run: rel.err(x) %.3e, rel.err(FN) %.3e" % (dx, dfn))
    assert dx < DROPIN_TOY_BOUNDS[0] and dfn < DROPIN_TOY_BOUNDS[1]
    assert rel_err(fn_our, fn_ref) < 2e-2
    # error path: NULL + message instead of a C++ exception across the boundary
    assert not lib.plmdcaBackend(2, q, b"/nonexistent.fa", L, 0.8, 1.0, 1.0, 5, 1, False)
    assert b"Unable to open" in lib.dca_last_error()


# ----------------------------------------------------------------------------- mfDCA
def _mf_ctx(L_, X1, q, seqid):
    ctx = L_.Context(0, L_.DCA_F64)
    ctx.set_msa((X1 - 1).astype(np.uint8), q)
    if seqid < 1.0:
        ctx.compute_weights(seqid, L_.DCA_F64)
    else:
        ctx.set_weights(np.ones(X1.shape[0]))
    return ctx


@pytest.mark.parametrize("tag", ["toy_rna", "toy_protein", "toy_rna_theta02_seqid1"])
def test_mf_stages_vs_reference(L_, tag):
    G = golden("mf_" + tag)
    X1, q = G["X"], int(G["q"])
    theta, seqid = float(G["pseudocount"]), float(G["seqid"])
    ctx = _mf_ctx(L_, X1, q, seqid)
    np.testing.assert_array_equal(ctx.weights(), G["w"])
    np.testing.assert_allclose(ctx.mf_single_site_freqs(), G["fi"], rtol=1e-13, atol=1e-16)
    np.testing.assert_allclose(ctx.mf_pair_site_freqs(), G["fij"], rtol=1e-12, atol=1e-15)   # dominant-state rows come from a complement (rounding ~1e-16 of the column total)
    np.testing.assert_allclose(ctx.mf_corr_mat(theta), G["corr_mat"], rtol=1e-11, atol=1e-15)
    np.testing.assert_allclose(ctx.mf_couplings(), G["couplings"], rtol=1e-8, atol=1e-10)
    np.testing.assert_allclose(ctx.mf_corr_from_freqs(G["reg_fi"], G["reg_fij"], X1.shape[1], q), G["corr_mat"],
                               rtol=1e-13, atol=1e-16)
    ctx.close()


@pytest.mark.parametrize("tag", ["toy_rna", "toy_protein", "toy_rna_theta02_seqid1", "rf71", "rf00167", "pf02826"])
def test_mf_scores_and_full_ranking_vs_reference(L_, oracle_mf, tag):
    """mfdca compute_fn: FN and FN_APC <= 1e-9 relative, identical full ranking
    (SURVEY 8c5); rf71 also covers the notebook's published top-5."""
    G = golden("mf_" + tag)
    X1, q = G["X"], int(G["q"])
    L = X1.shape[1]
    ctx = _mf_ctx(L_, X1, q, float(G["seqid"]))
    for apc, pk, sk in ((False, "fn_pairs", "fn_scores"), (True, "apc_pairs", "apc_scores")):
        scores = ctx.mf_run(float(G["pseudocount"]), apc)
        ranked = oracle_mf.sort_scores(scores, L)
        assert [p for p, _ in ranked] == [tuple(p) for p in G[pk]]
        np.testing.assert_allclose([s for _, s in ranked], G[sk], rtol=1e-9, atol=1e-12)   # atol: APC scores are differences of O(1) numbers
    ctx.close()


def test_mf_singular_matrix_is_an_error(L_):
    """pseudocount 0 on a tiny alignment: not positive definite -> error code, no crash
    (reference: LinAlgError path, meanfield_dca.py:542-548)."""
    X = np.zeros((4, 6), dtype=np.uint8)
    ctx = L_.Context(0, L_.DCA_F64)
    ctx.set_msa(X, 5)
    ctx.set_weights(np.ones(4))
    with pytest.raises(L_.DcaBackendError) as ei:
        ctx.mf_run(0.0, True)
    assert ei.value.code == L_.DCA_ERR_NOT_SPD
    ctx.close()


@pytest.mark.parametrize("n", [1, 64, 100, 128, 129, 192, 200, 256, 320, 500, 1000])
def test_spd_inverse_f64_mfma(L_, n):
    """Blocked Cholesky inverse on v_mfma_f64_16x16x4_f64 vs LAPACK; asymmetric-looking
    test matrices (random SPD, no special structure)."""
    rng = np.random.default_rng(n)
    B = rng.standard_normal((n, n + 8))
    A = B @ B.T / n + 0.5 * np.diag(rng.random(n) + 0.5)
    ctx = L_.Context(0, L_.DCA_F64)
    inv = ctx.spd_inverse(A)
    ref = np.linalg.inv(A)
    assert rel_err(inv, ref) < 1e-11
    assert np.array_equal(inv, inv.T)
    ctx.close()


# ----------------------------------------------------------------------------- DI
DI_TAGS = ["toy_rna", "toy_protein", "rf71"]


@pytest.mark.parametrize("tag", DI_TAGS)
def test_mf_direct_information_vs_reference(L_, oracle_mf, tag):
    """mfdca compute_di: DI and DI_APC vs MeanFieldDCA.compute_sorted_DI[_APC] of the real reference
    (golden).  The couplings differ from LAPACK's at ~1e-10, the fixed point stops at the same
    iteration; tolerance 1e-7 relative per score (rtol) with identical top-L ranking."""
    G, M = golden("di_" + tag), golden("mf_" + tag)
    X1, q = M["X"], int(G["q"])
    L = X1.shape[1]
    ctx = _mf_ctx(L_, X1, q, float(G["seqid"]))
    ctx.mf_run(float(G["pseudocount"]), False)
    di = ctx.mf_di_scores(False)
    np.testing.assert_allclose(di, G["mf_di"], rtol=1e-7, atol=1e-12)
    assert np.array_equal(np.argsort(-di, kind="stable")[:L], np.argsort(-G["mf_di"], kind="stable")[:L])
    np.testing.assert_allclose(ctx.mf_di_scores(True), G["mf_di_apc"], rtol=1e-6, atol=1e-10)
    ctx.close()


@pytest.mark.parametrize("tag", DI_TAGS)
def test_plm_direct_information_vs_reference(L_, tag):
    """plmdca compute_di numerics: DI of the reference's own optimised parameters (plm_<tag>.npz
    run_a, float32) with the reference's regularised frequencies -> plmdca/msa_numerics.py's output
    to 1e-9 relative (float64 arithmetic on both sides; only summation order differs)."""
    G, P = golden("di_" + tag), golden("plm_" + tag)
    L, q = int(G["L"]), int(G["q"])
    ctx = make_ctx(L_, P["X"], q, L_.DCA_F32)
    ctx.plm_configure(1.0, 1.0)
    ctx.plm_set_x(P["run_a"])
    di = ctx.plm_di_scores(G["plm_reg_fi"], False)
    np.testing.assert_allclose(di, G["plm_di"], rtol=1e-9, atol=1e-14)
    assert np.array_equal(np.argsort(-di, kind="stable"), np.argsort(-G["plm_di"], kind="stable"))
    ctx.close()


def test_di_kernel_random_blocks_vs_oracle(L_, oracle_mf):
    """DI / DI_APC kernel on random parameters, float64 storage, protein and RNA sizes."""
    rng = np.random.default_rng(5)
    for (L, q) in ((9, 5), (12, 21)):
        X = rng.integers(0, q, size=(30, L), dtype=np.uint8)
        ctx = make_ctx(L_, X, q, L_.DCA_F64, 0.8, L_.DCA_F64)
        ctx.plm_configure(1.0, 1.0)
        x = 0.7 * rng.standard_normal(ctx.num_params())
        ctx.plm_set_x(x)
        fi = rng.random((L, q)) + 0.05
        fi /= fi.sum(axis=1, keepdims=True)
        for apc in (False, True):
            np.testing.assert_allclose(ctx.plm_di_scores(fi, apc), oracle_mf.plm_di(x, fi, L, q, apc_correct=apc),
                                       rtol=1e-10, atol=1e-13)
        ctx.close()


# ----------------------------------------------------------------------------- fields / params
@pytest.mark.parametrize("tag", ["toy_rna", "toy_protein"])
def test_mf_fields_and_pair_couplings_vs_reference(L_, tag):
    """dca_mf_fields / dca_mf_pair_couplings vs MeanFieldDCA.compute_fields / compute_params of the
    real reference (golden): fields <= 1e-8 relative (the couplings differ from LAPACK's at
    ~1e-10), shifted blocks of the reference's own pair selection <= 1e-7."""
    G, M = golden("params_" + tag), golden("mf_" + tag)
    X1, q = M["X"], int(G["q"])
    ctx = _mf_ctx(L_, X1, q, float(G["seqid"]))
    ctx.mf_run(float(G["pseudocount"]), False)
    assert rel_err(ctx.mf_fields(), G["fields"]) <= 1e-8
    for name in ("default", "fn_ld2_n5", "diapc_ld1_n40"):
        pairs = G[name + "_pairs"]
        blocks = ctx.mf_pair_couplings(pairs, shift=True)
        assert blocks.shape == (len(pairs), q - 1, q - 1)
        if len(pairs):
            assert rel_err(blocks.reshape(len(pairs), -1), G[name + "_couplings"]) <= 1e-7
    raw = ctx.mf_pair_couplings([(0, 1)], shift=False)[0]
    np.testing.assert_allclose(raw, M["couplings"][0:q - 1, q - 1:2 * (q - 1)], rtol=1e-8, atol=1e-10)
    with pytest.raises(L_.DcaBackendError):
        ctx.mf_pair_couplings([(3, 2)])
    ctx.close()


def test_plm_pair_couplings_vs_oracle(L_, oracle_mf):
    P = golden("plm_toy_protein")
    L, q = int(P["L"]), int(P["q"])
    ctx = make_ctx(L_, P["X"], q, L_.DCA_F32)
    ctx.plm_configure(1.0, 1.0)
    ctx.plm_set_x(P["run_a"])
    pairs = [(0, 5), (2, 7), (1, 2)]
    got = ctx.plm_pair_couplings(pairs, shift=True)
    J = oracle_mf.plm_blocks(P["run_a"], L, q).astype(np.float64)
    iu, ju = np.triu_indices(L, k=1)
    idx = {(int(a), int(b)): k for k, (a, b) in enumerate(zip(iu, ju))}
    for k, pr in enumerate(pairs):
        np.testing.assert_allclose(got[k], oracle_mf.shift_couplings(J[idx[pr]]), rtol=1e-12, atol=1e-15)
    np.testing.assert_array_equal(ctx.plm_pair_couplings(pairs, shift=False)[1], J[idx[(2, 7)]])
    ctx.close()


def test_scores_order_matches_stable_argsort(L_, oracle_mf):
    """dca_scores_order (device radix sort) == numpy's stable descending argsort, ties included."""
    rng = np.random.default_rng(9)
    L, q = 40, 5
    X = rng.integers(0, q, size=(60, L), dtype=np.uint8)
    ctx = make_ctx(L_, X, q, L_.DCA_F64, 0.8, L_.DCA_F64)
    ctx.plm_configure(1.0, 1.0)
    x = rng.standard_normal(ctx.num_params())
    x[L * q + 50 * q * q:] = 0.0            # hundreds of exactly tied (zero) scores
    ctx.plm_set_x(x)
    for apc in (False, True):
        s = ctx.plm_scores(apc)
        assert np.array_equal(ctx.scores_order(), np.argsort(-s, kind="stable"))
    ctx.close()
    c2 = L_.Context(0, L_.DCA_F64)
    c2.set_msa(X, q)
    with pytest.raises(L_.DcaBackendError):
        c2.scores_order()
    c2.close()


@pytest.mark.parametrize("N,L,q", [(1, 2, 5), (3, 2, 21), (7, 5, 21), (33, 7, 5), (129, 13, 21), (130, 33, 5),
                                    (513, 6, 21), (640, 25, 5), (257, 31, 21),
                                    (96, 6, 21), (97, 7, 21), (768, 5, 21), (769, 8, 21), (49, 26, 5), (769, 27, 5),
                                    (2, 3, 5), (80, 23, 5), (81, 24, 5), (641, 25, 5), (300, 63, 5), (131, 64, 5), (90, 65, 5), (1281, 49, 5)])
def test_gradient_edge_shapes(L_, oracle_plm, N, L, q):
    """Shapes at and across the tile boundaries of the two gather kernels: fewer sequences than one wave's block of
    the logits kernel (96 for q = 21, 48 for q = 5; 32 earlier) / one 128-row tile / one workgroup's 768 sequences
    (512 earlier), each + 1; fewer sites than one LDS tile of W (6 resp. 25 sites) or one 32-site scatter group (+1), a
    single sequence, the minimum L = 2.  Round 5 (q = 5, float32: site pairs on the 25-state alphabet): odd L (the last
    site pairs with a padding site), one wave's 80 sequences and one workgroup's 640 (+1), 12-pair tiles of W (23 / 24 / 25 /
    49 sites), the scatter kernel's 64-site groups (63 / 64 / 65)."""
    rng = np.random.default_rng(1000 * N + 10 * L + q)
    X = rng.integers(0, q, size=(N, L), dtype=np.uint8)
    X = np.unique(X, axis=0)                      # the library expects de-duplicated rows like the reader yields
    rng.shuffle(X, axis=0)
    w = oracle_plm.weights(X, 0.8, np.float32)
    x = perturbed(oracle_plm.init_x(X, w, q), L, q)
    fx_o, g_o = oracle_plm.gradient(X, w.astype(np.float64), q, 0.7, 3.0, x.astype(np.float64), carry=True)
    for prec, tol in ((L_.DCA_F32, 2e-5), (L_.DCA_F64, 1e-10)):
        ctx = make_ctx(L_, X, q, prec, 0.8, L_.DCA_F32)
        ctx.plm_configure(0.7, 3.0)
        ctx.plm_set_x(x)
        fx = ctx.plm_gradient()
        g = ctx.plm_get_g(np.float64)
        assert abs(fx - fx_o) <= tol * abs(fx_o), (prec, fx, fx_o)
        assert rel_err(g, g_o) < tol, (prec, rel_err(g, g_o))
        ctx.close()


@pytest.mark.parametrize("tag", ["toy_rna", "rf71", "rf00167"])
def test_site_pair_alphabet_against_per_site_blocks(L_, oracle_plm, monkeypatch, tag):
    """q = 5, float32 (round 5): both gather kernels walk PAIRS of sites on the combined 25-state alphabet -- the logits kernel
    adds W[(j1, b1)] + W[(j2, b2)] formed once per wave and pair, the scatter kernel sums into 25 accumulators per pair and
    marginalises them when it stores.  Same sums, re-associated: against the per-site blocks (DCA_PLM_PAIRS=0) the objective
    and the gradient agree to float32 rounding (and are not the same bits), and both stay within the bound the per-site
    path has against the compiled reference's own (fx, g) (test_gradient_float32_vs_reference: 1e-5)."""
    G = golden("plm_" + tag)
    L, q = int(G["L"]), int(G["q"])
    assert q == 5
    lh, lJ = float(G["lambda_h"]), float(G["lambda_J"])
    w = oracle_plm.weights(G["X"], 0.8, np.float32)
    x = perturbed(oracle_plm.init_x(G["X"], w, q), L, q)
    out = {}
    for name, env in (("pairs", None), ("sites", "0")):
        if env is None:
            monkeypatch.delenv("DCA_PLM_PAIRS", raising=False)
        else:
            monkeypatch.setenv("DCA_PLM_PAIRS", env)
        ctx = make_ctx(L_, G["X"], q, L_.DCA_F32, 0.8, L_.DCA_F32)
        ctx.plm_configure(lh, lJ)
        ctx.plm_set_x(x)
        fx = ctx.plm_gradient()
        out[name] = (fx, ctx.plm_get_g(np.float64))
        ctx.close()
    fx_o, g_o = oracle_plm.gradient(G["X"], w.astype(np.float64), q, lh, lJ, x.astype(np.float64), carry=True)
    assert abs(out["pairs"][0] - out["sites"][0]) <= 2e-6 * abs(fx_o)
    assert rel_err(out["pairs"][1], out["sites"][1]) < 5e-6
    assert not np.array_equal(out["pairs"][1], out["sites"][1])
    for name in out:
        assert abs(out[name][0] - fx_o) <= 1e-5 * abs(fx_o), name
        assert rel_err(out[name][1], g_o) < 1e-5, (name, rel_err(out[name][1], g_o))


def test_device_block_cache_reuse_and_release(L_):
    """Device blocks of >= 1 MiB are recycled across contexts (csrc/capi.cpp dca_dev_malloc): a second context that
    gets the first one's (dirty) blocks must produce the same bits, and dca_release_cached_memory hands the cache back."""
    rng = np.random.default_rng(17)
    N, L, q = 3000, 120, 21                      # (L q)^2 doubles = 50 MB, L x N list = 1.4 MB: well above the 1 MiB floor
    X = rng.integers(0, q, size=(N, L), dtype=np.uint8)
    X[N // 2:] = X[:N - N // 2]                  # duplicates give non-trivial weights
    X[N // 2:, ::7] = rng.integers(0, q, size=(N - N // 2, len(range(0, L, 7))), dtype=np.uint8)
    L_.release_cached_memory()

    def run():
        ctx = L_.Context(0, L_.DCA_F64)
        ctx.set_msa(X, q)
        ctx.compute_weights(0.8, L_.DCA_F64)
        scores, J = ctx.mf_run(0.5, True, want_couplings=True)
        order = ctx.scores_order()
        ctx.close()
        return scores, J, order

    s1, J1, o1 = run()
    s2, J2, o2 = run()                           # served from the cache
    assert np.array_equal(s1, s2) and np.array_equal(J1, J2) and np.array_equal(o1, o2)
    freed = L_.release_cached_memory()
    assert freed >= (L * q) ** 2 * 8
    assert L_.release_cached_memory() == 0
    s3, _, _ = run()                             # fresh blocks again
    assert np.array_equal(s1, s3)
