import os, sys, time
import numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from pydca_amd import _lib
from tools.gen_msa import dedup, generate
X = dedup(generate(500, 50000, 21, 12346))
# mimic bench: a plm context alive
full = _lib.Context(0, _lib.DCA_F32); full.set_msa(X, 21); full.compute_weights(0.8, _lib.DCA_F32); full.plm_configure(1.0, 50.0); full.plm_init_x()
full.plm_lbfgs_begin(10); full.plm_lbfgs_iterate(3)
for rep in range(4):
    t = [time.perf_counter()]
    m = _lib.Context(0, _lib.DCA_F64); t.append(time.perf_counter())
    m.set_msa(X, 21); t.append(time.perf_counter())
    m.set_profiling(True)
    m.compute_weights(0.8, _lib.DCA_F64); t.append(time.perf_counter())
    s = m.mf_run(0.5, True); t.append(time.perf_counter())
    o = m.scores_order(); t.append(time.perf_counter())
    m.close(); t.append(time.perf_counter())
    print("rep", rep, "create %.1f set_msa %.1f weights %.1f mf_run %.1f order %.1f close %.1f ms" % tuple((t[k+1]-t[k])*1e3 for k in range(6)), flush=True)
