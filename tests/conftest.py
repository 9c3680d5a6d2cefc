import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")
DATA = os.path.join(GOLDEN, "data")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


def golden(name):
    return np.load(os.path.join(GOLDEN, name + ".npz"))


def data_file(name):
    return os.path.join(DATA, name)


@pytest.fixture(scope="session")
def oracle_plm():
    from oracle import plm
    plm.lib()
    return plm


@pytest.fixture(scope="session")
def oracle_mf():
    from oracle import mf
    return mf


def perturbed(x0, L, q):
    """Same perturbation tests/golden/make_golden.py applied before calling the reference."""
    x = x0.copy()
    k = np.arange(x.size - L * q, dtype=np.float64)
    x[L * q:] = (0.05 * np.sin(0.37 * k)).astype(x.dtype)
    return x


def rel_err(a, b):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-300))


def assert_scores_within(s_gpu, s_ref, fn_ref, rtol=1e-4, top=None):
    """The north-star tolerance ("FN / DI scores within 1e-4 relative") for a score vector and its APC-corrected form.
    FN and DI are positive quantities and are compared relative to themselves.  An APC-corrected score is the
    DIFFERENCE FN_ij - av_i av_j / av of two O(FN) numbers and crosses zero, so "relative to itself" is unbounded there
    for ANY two float64 implementations (PF02826 after 100 iterations: 7e-7 absolute on scores of 2e-5); it is compared
    relative to the uncorrected score of the same pair: |d FN_APC_ij| <= rtol * FN_ij.  With `top` (= L) the `top` highest
    reference scores -- the contacts a user consumes, far from zero -- must ALSO be within rtol of THEMSELVES, so the
    north-star tolerance is enforced as written where it is meaningful.  Returns the largest ratio."""
    s_gpu, s_ref, fn_ref = (np.asarray(v, dtype=np.float64) for v in (s_gpu, s_ref, fn_ref))
    ratio = np.abs(s_gpu - s_ref) / np.maximum(np.abs(fn_ref), 1e-300)
    worst = int(np.argmax(ratio))
    assert ratio[worst] <= rtol, "pair %d: %r vs %r (uncorrected score %r): %.3e > %.1e" % (
        worst, s_gpu[worst], s_ref[worst], fn_ref[worst], ratio[worst], rtol)
    if top:
        idx = np.argsort(-s_ref, kind="stable")[:top]
        self_ratio = np.abs(s_gpu[idx] - s_ref[idx]) / np.maximum(np.abs(s_ref[idx]), 1e-300)
        k = int(np.argmax(self_ratio))
        assert self_ratio[k] <= rtol, "top-%d pair %d: %r vs %r: %.3e > %.1e (relative to itself)" % (
            top, int(idx[k]), s_gpu[idx[k]], s_ref[idx[k]], self_ratio[k], rtol)
    return float(ratio[worst])
