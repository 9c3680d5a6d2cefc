import os, sys, time
import numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from pydca_amd import _lib
from tools.gen_msa import dedup, generate
import argparse
ap = argparse.ArgumentParser(); ap.add_argument('--L', type=int, default=500); ap.add_argument('--N', type=int, default=50000); ap.add_argument('--q', type=int, default=21); ap.add_argument('--seed', type=int, default=12346)
a = ap.parse_args()
X = dedup(generate(a.L, a.N, a.q, a.seed))
for rep in range(3):
    t = [time.perf_counter()]
    ctx = _lib.Context(0, _lib.DCA_F64); t.append(time.perf_counter())
    ctx.set_msa(X, a.q); t.append(time.perf_counter())
    ctx.compute_weights(0.8, _lib.DCA_F64); t.append(time.perf_counter())
    scores = ctx.mf_run(0.5, True); t.append(time.perf_counter())
    order = ctx.scores_order(); t.append(time.perf_counter())
    print("rep", rep, "create %.1f set_msa %.1f weights %.1f mf_run %.1f argsort %.1f ms" % tuple((t[k+1]-t[k])*1e3 for k in range(5)))
    ctx.set_profiling(True)
    t0=time.perf_counter(); scores = ctx.mf_run(0.5, True); print("  second mf_run %.1f ms" % ((time.perf_counter()-t0)*1e3), {k: round(ctx.kernel_time(k)[0], 2) for k in ("mf_sort", "mf_counts", "mf_inverse", "scores")})
    ctx.close()
