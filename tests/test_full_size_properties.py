"""Full-size checks of the HIP path at BASELINE.json's configurations C (protein L=200 N=10 000 q=21; compared
element-wise with the oracle in tests/test_gpu_configs.py as well), D (protein L=500 N=50 000 q=21) and
E (RNA L=150 N=200 000 q=5), where the CPU oracle needs about a minute per evaluation even on all host
cores: size-independent properties of the algorithm instead of element-wise comparison.

  plmDCA (plmdca_numerics.cpp:436-607)
    * marginal sums of the data gradient: the site-i term of dJ_ij(a, x_nj) sums to zero over a, the site-j term
      of dJ_ij(x_ni, b) sums over a to the data gradient of h_j(b); so for every pair
      sum_a dJ_ij(a, b) = dh_j(b) and sum_b dJ_ij(a, b) = dh_i(a)       (holds with the carry-over as well)
    * float32 and float64 device paths agree at full size (independent accumulation orders)
    * exact mode: central differences of fx along a random direction equal g . d   (float64)
    * the evaluation is additive over sequence blocks (exact mode): two half alignments sum to the whole
  sequence weights (plmdca_numerics.cpp:611-671): sampled rows recomputed with numpy, bit for bit
  mfDCA (msa_numerics.py:53-342, meanfield_dca.py:902-988)
    * C * (J v) = -v for random v (J = -inv(C)), J bit-symmetric
    * ranking: scores[order] never increases, ties in ascending pair index, order is a permutation
"""
import os
import sys

import numpy as np
import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu

sys.path.insert(0, ROOT)
from tools.gen_msa import SEEDS, dedup, generate  # noqa: E402

CONFIGS = {"C": (200, 10000, 21, 1.0, 50.0), "D": (500, 50000, 21, 1.0, 50.0), "E": (150, 200000, 5, 29.8, 29.8)}


@pytest.fixture(scope="module")
def L_():
    from pydca_amd import _lib
    _lib.lib()
    return _lib


_msa_cache = {}


def msa(tag):
    if tag not in _msa_cache:
        L, N, q = CONFIGS[tag][:3]
        _msa_cache[tag] = dedup(generate(L, N, q, SEEDS[tag]))
    return _msa_cache[tag]


def split(v, L, q):
    """packed parameter vector (plmdca_numerics.cpp:467-480) -> h[L, q], J[pairs, q, q]"""
    return v[:L * q].reshape(L, q), v[L * q:].reshape(-1, q, q)


def perturbed_start(ctx, L, q, dtype):
    ctx.plm_init_x()
    x = ctx.plm_get_x(dtype)
    k = np.arange(x.size - L * q, dtype=np.float64)
    x[L * q:] = (0.02 * np.sin(0.37 * k)).astype(dtype)
    ctx.plm_set_x(x)
    return x


@pytest.mark.parametrize("tag", ["C", "D", "E"])
def test_plm_gradient_marginals_full_size(L_, tag):
    L, _, q, lh, lJ = CONFIGS[tag]
    X = msa(tag)
    ctx = L_.Context(0, L_.DCA_F32)
    ctx.set_msa(X, q)
    ctx.compute_weights(0.8)
    meff = ctx.meff()
    ctx.plm_configure(lh, lJ, L_.CARRY_CHUNKED, add_regulariser=0)
    perturbed_start(ctx, L, q, np.float32)
    fx = ctx.plm_gradient()
    assert np.isfinite(fx) and fx > 0
    g = ctx.plm_get_g(np.float64)
    gh, gJ = split(g, L, q)
    ii, jj = np.triu_indices(L, 1)            # pair order (0,1), (0,2), ... = the packed order
    scale = meff                               # |dh| is bounded by the summed weights
    err_j = np.abs(gJ.sum(axis=1) - gh[jj]).max() / scale
    err_i = np.abs(gJ.sum(axis=2) - gh[ii]).max() / scale
    assert err_j < 2e-6 and err_i < 2e-6, (err_i, err_j)
    # every site's field gradient sums to zero over the states: sum_a (p(a) - delta) = 0
    assert np.abs(gh.sum(axis=1)).max() / scale < 2e-6

    # float64 path, same x: independent accumulation order, same numbers
    x32 = ctx.plm_get_x(np.float32)
    ctx.close()
    ctx64 = L_.Context(0, L_.DCA_F64)
    ctx64.set_msa(X, q)
    ctx64.compute_weights(0.8, L_.DCA_F32)
    ctx64.plm_configure(lh, lJ, L_.CARRY_CHUNKED, add_regulariser=0)
    ctx64.plm_set_x(x32.astype(np.float64))
    fx64 = ctx64.plm_gradient()
    g64 = ctx64.plm_get_g(np.float64)
    ctx64.close()
    assert abs(fx - fx64) / fx64 < 1e-6
    # float32 sums of up to 200 000 addends per entry (as in the reference): rounding grows like sqrt(N) * 2^-24
    assert np.linalg.norm(g - g64) / np.linalg.norm(g64) < 1e-4
    gh64, gJ64 = split(g64, L, q)
    assert np.abs(gJ64.sum(axis=1) - gh64[jj]).max() / scale < 1e-12


def test_plm_exact_mode_directional_derivative_and_additivity_full_size(L_):
    """Exact mode at config D in float64: g is the gradient of fx (central differences along a random
    direction), and fx, g are sums over sequences (two half alignments with the full alignment's weights)."""
    L, _, q, lh, lJ = CONFIGS["D"]
    X = msa("D")
    N = X.shape[0]
    ctx = L_.Context(0, L_.DCA_F64)
    ctx.set_msa(X, q)
    w = ctx.compute_weights(0.8, L_.DCA_F32)
    ctx.plm_configure(lh, lJ, L_.CARRY_EXACT)
    x = perturbed_start(ctx, L, q, np.float64)
    fx0 = ctx.plm_gradient()
    g = ctx.plm_get_g(np.float64)
    rng = np.random.default_rng(7)
    d = rng.standard_normal(x.size)
    d = d / np.linalg.norm(d) + g / np.linalg.norm(g)     # a direction along which fx changes by much more than its rounding noise
    d /= np.linalg.norm(d)
    eps = 1e-3
    ctx.plm_set_x(x + eps * d)
    fp = ctx.plm_gradient()
    ctx.plm_set_x(x - eps * d)
    fm = ctx.plm_gradient()
    num = (fp - fm) / (2 * eps)
    ana = float(g @ d)
    assert abs(num - ana) / max(abs(ana), 1e-12) < 2e-5, (num, ana)
    # additivity over sequence blocks (data term only)
    ctx.plm_configure(lh, lJ, L_.CARRY_EXACT, add_regulariser=0)
    ctx.plm_set_x(x)
    f_all = ctx.plm_gradient()
    g_all = ctx.plm_get_g(np.float64)
    ctx.close()
    f_sum, g_sum = 0.0, np.zeros_like(g_all)
    cut = N // 2 + 13
    for lo, hi in ((0, cut), (cut, N)):
        c = L_.Context(0, L_.DCA_F64)
        c.set_msa(X[lo:hi], q)
        c.set_weights(w[lo:hi])
        c.plm_configure(lh, lJ, L_.CARRY_EXACT, add_regulariser=0)
        c.plm_set_x(x)
        f_sum += c.plm_gradient()
        g_sum += c.plm_get_g(np.float64)
        c.close()
    assert abs(f_sum - f_all) / abs(f_all) < 1e-12
    assert np.linalg.norm(g_sum - g_all) / np.linalg.norm(g_all) < 1e-12
    del fx0


@pytest.mark.parametrize("tag", ["C", "D", "E"])
def test_weights_sampled_rows_full_size(L_, tag):
    L, _, q = CONFIGS[tag][:3]
    X = msa(tag)
    N = X.shape[0]
    ctx = L_.Context(0, L_.DCA_F32)
    ctx.set_msa(X, q)
    ctx.compute_weights(0.8)
    counts = ctx.weight_counts()
    w = ctx.weights()
    ctx.close()
    rng = np.random.default_rng(3)
    rows = np.concatenate([[0, 1, N - 1], rng.integers(0, N, size=29)])
    thr = np.float32(0.8)
    for n in rows:
        ident = (X == X[n]).sum(axis=1)
        ref = int(((ident.astype(np.float32) / np.float32(L)) > thr).sum())      # plmdca_numerics.cpp:636-640
        assert counts[n] == ref, (tag, int(n), int(counts[n]), ref)
    assert counts.min() >= 1
    assert np.array_equal(w.astype(np.float32), (np.float32(1.0) / counts.astype(np.float32)))


def test_mf_inverse_and_ranking_full_size(L_):
    L, _, q = CONFIGS["D"][:3]
    X = msa("D")
    ctx = L_.Context(0, L_.DCA_F64)
    ctx.set_msa(X, q)
    ctx.compute_weights(0.8, L_.DCA_F64)
    scores, J = ctx.mf_run(0.5, True, want_couplings=True)
    order = ctx.scores_order()
    Cm = ctx.mf_corr_mat(0.5)
    ctx.close()
    n = L * (q - 1)
    assert J.shape == (n, n) and np.array_equal(J, J.T)
    assert np.array_equal(Cm, Cm.T)
    rng = np.random.default_rng(11)
    V = rng.standard_normal((n, 4))
    resid = Cm @ (J @ V) + V
    assert np.linalg.norm(resid) / np.linalg.norm(V) < 1e-9
    # ranking = Python's sorted(..., reverse=True) on the pair-ordered list (meanfield_dca.py:941, :986)
    assert np.array_equal(np.sort(order), np.arange(scores.size))
    s = scores[order]
    assert np.all(s[:-1] >= s[1:])
    ties = s[:-1] == s[1:]
    assert np.all(order[:-1][ties] < order[1:][ties])
    assert np.array_equal(order, np.argsort(-scores, kind="stable"))
    # APC scores of a symmetric-in-(i, j) construction average to zero-mean corrections: finite and not all equal
    assert np.isfinite(scores).all() and scores.std() > 0
