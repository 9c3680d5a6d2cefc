"""mf_inverse (ms, median of 5 passes) at config D inside the mfDCA chain with a plmDCA context alive, as bench.py has it."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from pydca_amd import _lib
from tools.gen_msa import dedup, generate
X = dedup(generate(500, 50000, 21, 12346))
full = _lib.Context(0, _lib.DCA_F32); full.set_msa(X, 21); full.compute_weights(0.8, _lib.DCA_F32); full.plm_configure(1.0, 50.0); full.plm_init_x()
full.plm_lbfgs_begin(10); full.plm_lbfgs_iterate(3)
ts = []
for rep in range(6):
    m = _lib.Context(0, _lib.DCA_F64); m.set_msa(X, 21); m.set_profiling(True); m.compute_weights(0.8, _lib.DCA_F64)
    m.mf_run(0.5, True)
    if rep: ts.append(m.kernel_time("mf_inverse")[0])
    m.close()
print("%s mf_inverse median %.2f min %.2f ms" % (sys.argv[1] if len(sys.argv) > 1 else "", float(np.median(ts)), min(ts)))
