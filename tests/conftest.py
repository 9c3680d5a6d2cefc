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
