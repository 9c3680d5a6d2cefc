"""Shapes that none of BASELINE.json's configurations reaches (round-5 verdict, item 3): Pfam families do.

  * N * L * q > 2^31 elements of the per-sequence tables (L = 500, q = 21, N = 250 000)
  * L >= 1000 at q = 21 (parameter vectors of 0.9 - 3.5 GB, more column strips than the XCD rounds were tuned for)
  * long RNA (L = 1500, q = 5: the site-pair alphabet with hundreds of pairs per tile row)
  * fewer sequences than a wave has lanes (N = 1 ... 63)
  * mfDCA at n = L (q - 1) = 20 000 (twice config D's matrix: 3.2 GB, 8e12 flop)

Too large for the CPU oracle: the size-independent properties of tests/test_full_size_properties.py -- marginal sums of the
data gradient (plmdca_numerics.cpp:436-607), float32 against float64 at the same x, sampled rows of the weights recomputed with
numpy (plmdca_numerics.cpp:611-671), C (J v) = -v for the inverse (msa_numerics.py:321-342) -- and, below N = 64, the oracle
itself."""
import sys

import numpy as np
import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu

sys.path.insert(0, ROOT)
from tools.gen_msa import dedup, generate  # noqa: E402


@pytest.fixture(scope="module")
def L_():
    from pydca_amd import _lib
    _lib.lib()
    return _lib


def perturbed_start(ctx, L, q, dtype):
    ctx.plm_init_x()
    x = ctx.plm_get_x(dtype)
    k = np.arange(x.size - L * q, dtype=np.float32)
    x[L * q:] = (0.02 * np.sin(0.37 * k)).astype(dtype)
    ctx.plm_set_x(x)
    return x


def check_marginals(g, L, q, scale, tol):
    """sum_a dJ_ij(a, b) = dh_j(b), sum_b dJ_ij(a, b) = dh_i(a), sum_a dh_i(a) = 0 -- pair block by pair block (float32 host
    arrays: the vectors of the large shapes are GB-sized)"""
    gh = g[:L * q].reshape(L, q).astype(np.float64)
    gJ = g[L * q:].reshape(-1, q, q)
    assert np.abs(gh.sum(axis=1)).max() / scale < tol
    worst = 0.0
    first = 0
    for i in range(L - 1):                       # the pairs (i, i + 1 ..) are contiguous in the packed order
        blk = gJ[first:first + L - 1 - i].astype(np.float64)
        first += L - 1 - i
        worst = max(worst, np.abs(blk.sum(axis=1) - gh[i + 1:]).max(), np.abs(blk.sum(axis=2) - gh[i]).max())
    assert first == gJ.shape[0]
    assert worst / scale < tol, worst / scale


def sampled_weight_rows(L_, X, q, rows):
    L = X.shape[1]
    ctx = L_.Context(0, L_.DCA_F32)
    ctx.set_msa(X, q)
    ctx.compute_weights(0.8)
    counts = ctx.weight_counts()
    ctx.close()
    thr = np.float32(0.8)
    for n in rows:
        ident = (X == X[n]).sum(axis=1)
        ref = int(((ident.astype(np.float32) / np.float32(L)) > thr).sum())      # plmdca_numerics.cpp:636-640
        assert counts[n] == ref, (int(n), int(counts[n]), ref)
    assert counts.min() >= 1


@pytest.mark.parametrize("L,N,q,lh,lJ", [(500, 250000, 21, 1.0, 50.0), (1024, 5000, 21, 1.0, 50.0), (2000, 5000, 21, 1.0, 50.0),
                                         (1500, 5000, 5, 29.8, 29.8)])
def test_plm_gradient_beyond_baseline_shapes(L_, L, N, q, lh, lJ):
    X = dedup(generate(L, N, q, 4242 + L))
    assert X.shape[0] * L * q > 2**31 or L > 500
    ctx = L_.Context(0, L_.DCA_F32)
    ctx.set_msa(X, q)
    ctx.compute_weights(0.8)
    meff = ctx.meff()
    ctx.plm_configure(lh, lJ, L_.CARRY_CHUNKED, add_regulariser=0)
    x32 = perturbed_start(ctx, L, q, np.float32)
    fx = ctx.plm_gradient()
    assert np.isfinite(fx) and fx > 0
    g = ctx.plm_get_g(np.float32)
    ctx.close()
    assert np.isfinite(g).all()
    check_marginals(g, L, q, meff, 4e-6)
    if L <= 1024:
        # float64 path at the same x: independent accumulation order, same numbers
        ctx64 = L_.Context(0, L_.DCA_F64)
        ctx64.set_msa(X, q)
        ctx64.compute_weights(0.8, L_.DCA_F32)
        ctx64.plm_configure(lh, lJ, L_.CARRY_CHUNKED, add_regulariser=0)
        ctx64.plm_set_x(x32.astype(np.float64))
        fx64 = ctx64.plm_gradient()
        g64 = ctx64.plm_get_g(np.float64)
        ctx64.close()
        assert abs(fx - fx64) / fx64 < 1e-6
        assert np.linalg.norm(g - g64) / np.linalg.norm(g64) < 2e-4
    rng = np.random.default_rng(5)
    sampled_weight_rows(L_, X, q, np.concatenate([[0, X.shape[0] - 1], rng.integers(0, X.shape[0], size=6)]))


@pytest.mark.parametrize("N", [1, 2, 33, 63])
@pytest.mark.parametrize("q,L", [(21, 37), (5, 40)])
def test_plm_fewer_sequences_than_lanes(L_, oracle_plm, N, q, L):
    """N < 64: every tile of sequences is ragged; against the CPU oracle (float64, exact mode) element by element."""
    rng = np.random.default_rng(100 * N + q)
    X = rng.integers(0, q, size=(N, L), dtype=np.uint8)
    X = X[np.sort(np.unique(X, axis=0, return_index=True)[1])]
    ctx = L_.Context(0, L_.DCA_F64)
    ctx.set_msa(X, q)
    w = ctx.compute_weights(0.8, L_.DCA_F32)
    ctx.plm_configure(1.0, 5.0, L_.CARRY_EXACT)
    x = perturbed_start(ctx, L, q, np.float64)
    fx = ctx.plm_gradient()
    g = ctx.plm_get_g(np.float64)
    ctx.close()
    w_ref = oracle_plm.weights(X, 0.8)
    assert np.array_equal(w.astype(np.float32), w_ref.astype(np.float32))
    fx_ref, g_ref = oracle_plm.gradient(X, w_ref.astype(np.float64), q, 1.0, 5.0, x, carry=False)
    assert abs(fx - fx_ref) / abs(fx_ref) < 1e-12
    assert np.linalg.norm(g - g_ref) / np.linalg.norm(g_ref) < 1e-12


def test_mf_inverse_at_n_20000(L_):
    """mfDCA at L = 1000, q = 21: n = 20 000 (the block sweep with 40 panels of 512 columns, 157 tile rows)."""
    L, N, q = 1000, 8000, 21
    X = dedup(generate(L, N, q, 99))
    ctx = L_.Context(0, L_.DCA_F64)
    ctx.set_msa(X, q)
    ctx.compute_weights(0.8, L_.DCA_F64)
    scores, J = ctx.mf_run(0.5, True, want_couplings=True)
    order = ctx.scores_order()
    Cm = ctx.mf_corr_mat(0.5)
    ctx.close()
    n = L * (q - 1)
    assert J.shape == (n, n) and np.array_equal(J, J.T)
    rng = np.random.default_rng(11)
    V = rng.standard_normal((n, 3))
    resid = Cm @ (J @ V) + V
    assert np.linalg.norm(resid) / np.linalg.norm(V) < 1e-9
    assert np.isfinite(scores).all() and np.array_equal(order, np.argsort(-scores, kind="stable"))
