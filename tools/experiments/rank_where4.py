"""Round 6: is the slow list construction after a fresh context transient (driver work still going on) or a state?"""
import gc, os, resource, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from pydca_amd import _lib, _ranking
from tools.gen_msa import dedup, generate
X = dedup(generate(500, 50000, 21, 12346))
def timed(s, o):
    t0 = time.perf_counter(); r = _ranking.ranked(s, 500, o); return r, (time.perf_counter() - t0) * 1e3
c = _lib.Context(0, _lib.DCA_F64); c.set_msa(X, 21); c.compute_weights(0.8, _lib.DCA_F64)
for rep in range(3):
    s = c.mf_run(0.5, True); o = c.scores_order(); r, t = timed(s, o); r = None
for mode in ("plain", "sleep 30 ms", "twice", "spin 5 ms", "no weights readback"):
    for rep in range(3):
        r = r2 = None
        c2 = _lib.Context(0, _lib.DCA_F64); c2.set_msa(X, 21); c2.compute_weights(0.8, _lib.DCA_F64)
        s = c2.mf_run(0.5, True); o = c2.scores_order()
        if mode == "sleep 30 ms": time.sleep(0.03)
        if mode == "spin 5 ms":
            t0 = time.perf_counter()
            while time.perf_counter() - t0 < 0.005: pass
        r, t = timed(s, o)
        extra = ""
        if mode == "twice":
            r2, t2 = timed(s, o); extra = " then %.2f" % t2
        print("%-20s list %.2f ms%s   threads %d" % (mode, t, extra, len(os.listdir("/proc/self/task"))))
        c2.close()
# the same context reused, but its big buffers dropped from the pool each pass?
for rep in range(3):
    r = None
    s = c.mf_run(0.5, True); o = c.scores_order(); r, t = timed(s, o)
    print("reused context       list %.2f ms" % t)
