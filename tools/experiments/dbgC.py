import sys, numpy as np
sys.path.insert(0, ".")
from oracle import plm as oplm
from pydca_amd import _lib as L_
from tools.gen_msa import generate, dedup, SEEDS
X = dedup(generate(200, 10000, 21, SEEDS["C"])); q = 21; L = 200
w = oplm.weights(X, 0.8, np.float64)
x0 = oplm.init_x(X, w, q)
fo, go = oplm.gradient(X, w, q, 1.0, 50.0, x0, carry=True)
fe, ge = oplm.gradient(X, w, q, 1.0, 50.0, x0, carry=False)
print("oracle carry fx", repr(fo), " exact fx", repr(fe))
def run(mode, chunk=0, warm=0, reg=1, N=None):
    Xs = X if N is None else X[:N]
    ctx = L_.Context(0, L_.DCA_F64); ctx.set_msa(Xs, q); ctx.set_weights(w[:len(Xs)])
    ctx.plm_configure(1.0, 50.0, mode, chunk, warm, add_regulariser=reg)
    ctx.plm_set_x(x0); fx = ctx.plm_gradient(); g = ctx.plm_get_g(np.float64); ctx.close(); return fx, g
for name, mode, chunk, warm in (("serial", L_.CARRY_SERIAL, 0, 0), ("chunk32", L_.CARRY_CHUNKED, 32, 40), ("chunk64", L_.CARRY_CHUNKED, 64, 40),
                                ("chunk128", L_.CARRY_CHUNKED, 128, 40), ("chunk256", L_.CARRY_CHUNKED, 256, 40), ("chunk32w80", L_.CARRY_CHUNKED, 32, 80), ("default", L_.CARRY_CHUNKED, 0, 0)):
    fx, g = run(mode, chunk, warm)
    print("%-10s fx %r  dfx %.3e  rel g err %.3e  max|dg| %.3e" % (name, fx, fx - fo, np.linalg.norm(g - go) / np.linalg.norm(go), np.abs(g - go).max()))
fx, g = run(L_.CARRY_EXACT)
print("exact      fx %r dfx %.3e rel g err %.3e" % (fx, fx - fe, np.linalg.norm(g - ge) / np.linalg.norm(ge)))
for N in (500, 2000, 5000):
    fo2, go2 = oplm.gradient(X[:N], w[:N], q, 1.0, 50.0, x0, carry=True)
    fx, g = run(L_.CARRY_SERIAL, N=N)
    print("serial N=%d dfx %.3e rel g %.3e" % (N, fx - fo2, np.linalg.norm(g - go2) / np.linalg.norm(go2)))
