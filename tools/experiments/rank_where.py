"""Round 6: the ranking stage of the class path, piece by piece (order from the device, list construction), in the process state the
class path has (alignment arrays alive, a context open)."""
import gc, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from pydca_amd import _lib, _ranking
from tools.gen_msa import dedup, generate
X = dedup(generate(500, 50000, 21, 12346))
ms = lambda a, b: (b - a) * 1e3
c = _lib.Context(0, _lib.DCA_F64); c.set_msa(X, 21); c.compute_weights(0.8, _lib.DCA_F64)
for rep in range(8):
    r = None
    s = c.mf_run(0.5, True)
    t0 = time.perf_counter(); o = c.scores_order(); t1 = time.perf_counter()
    r = _ranking.ranked(s, 500, o); t2 = time.perf_counter()
    r = None; t3 = time.perf_counter()
    print("rep %d order %.2f ranked %.2f free %.2f ms   gc %s counts %s" % (rep, ms(t0, t1), ms(t1, t2), ms(t2, t3), gc.isenabled(), gc.get_count()))
import ctypes
try:
    libc = ctypes.CDLL("libc.so.6")
    libc.mallopt(-1, 1 << 30)      # M_TRIM_THRESHOLD
    libc.mallopt(-3, 1 << 30)      # M_MMAP_THRESHOLD
    print("mallopt set")
except Exception as e:
    print("mallopt", e)
for rep in range(4):
    r = None
    s = c.mf_run(0.5, True)
    t0 = time.perf_counter(); o = c.scores_order(); t1 = time.perf_counter()
    r = _ranking.ranked(s, 500, o); t2 = time.perf_counter()
    print("rep %d order %.2f ranked %.2f ms" % (rep, ms(t0, t1), ms(t1, t2)))
