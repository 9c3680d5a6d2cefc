"""Round 6: where the host time of `mfdca compute_fn` at config D goes (context, upload, weights, ranking), piece by piece."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from pydca_amd import _lib, _ranking
from tools.gen_msa import dedup, generate
X = dedup(generate(500, 50000, 21, 12346))
print("fastrank:", _ranking._fast)
for rep in range(4):
    t = [time.perf_counter()]
    c = _lib.Context(0, _lib.DCA_F64); t.append(time.perf_counter())
    c.set_msa(X, 21); t.append(time.perf_counter())
    w = c.compute_weights(0.8, _lib.DCA_F64); t.append(time.perf_counter())
    s = c.mf_run(0.5, True); t.append(time.perf_counter())
    o = c.scores_order(); t.append(time.perf_counter())
    r = _ranking.ranked(s, 500, o); t.append(time.perf_counter())
    c.close(); t.append(time.perf_counter())
    print("rep %d: create %.2f set_msa %.2f weights %.2f mf_run %.2f order %.2f ranked %.2f close %.2f ms" % ((rep,) + tuple((t[k + 1] - t[k]) * 1e3 for k in range(7))))
keep = _ranking._fast
_ranking._fast = None
t0 = time.perf_counter(); r2 = _ranking.ranked(s, 500, o); print("python ranked %.2f ms, same %s" % ((time.perf_counter() - t0) * 1e3, r2 == r))
