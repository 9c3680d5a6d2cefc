import os, sys, time
import numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from pydca_amd import _lib
from tools.gen_msa import dedup, generate
X = dedup(generate(500, 50000, 21, 12346))
ctx = _lib.Context(0, _lib.DCA_F64)
ctx.set_msa(X, 21)
ctx.compute_weights(0.8, _lib.DCA_F64)
ctx.mf_run(0.5, True)
for name, fn in (("mf_di", lambda: ctx.mf_di_scores(False)), ("mf_di_apc", lambda: ctx.mf_di_scores(True)), ("mf_fn_apc", lambda: ctx.mf_scores(True)),
                 ("mf_fields", ctx.mf_fields), ("order", ctx.scores_order),
                 ("pair_couplings(500)", lambda: ctx.mf_pair_couplings([(i, i + 7) for i in range(490)]))):
    fn()
    t0 = time.perf_counter(); fn(); print("%-22s %.2f ms" % (name, (time.perf_counter() - t0) * 1e3))
