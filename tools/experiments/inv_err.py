import sys, numpy as np
sys.path.insert(0, '/root/repo')
from pydca_amd import _lib
ctx = _lib.Context(0, _lib.DCA_F64)
for n in (64, 200, 500, 1000, 2000):
    rng = np.random.default_rng(n)
    B = rng.standard_normal((n, n + 8))
    A = B @ B.T / n + 0.5 * np.diag(rng.random(n) + 0.5)
    inv = ctx.spd_inverse(A)
    ref = np.linalg.inv(A)
    print(n, "rel err vs LAPACK %.3e" % (np.linalg.norm(inv - ref) / np.linalg.norm(ref)), "resid %.3e" % (np.linalg.norm(A @ inv - np.eye(n)) / np.sqrt(n)), "lapack resid %.3e" % (np.linalg.norm(A @ ref - np.eye(n)) / np.sqrt(n)))
