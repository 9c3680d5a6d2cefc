import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from pydca_amd import _ranking as R
L = 500; n = L * (L - 1) // 2
rng = np.random.default_rng(0); s = rng.random(n); order = np.argsort(-s, kind="stable").astype(np.int32)
fast = R._fast
res = {"fast": [], "python": []}
for rep in range(12):
    for name in ("fast", "python"):
        R._fast = fast if name == "fast" else None
        t0 = time.perf_counter(); x = R.ranked(s, L, order); res[name].append((time.perf_counter() - t0) * 1e3)
        if rep % 2: del x          # with and without the previous list alive
for k, v in res.items(): print(k, "min %.2f median %.2f max %.2f ms" % (min(v), sorted(v)[len(v) // 2], max(v)))
