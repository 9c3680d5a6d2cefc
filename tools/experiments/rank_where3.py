"""Round 6: why the list construction costs 10 - 15 ms in the class path and 4.6 ms in a bare loop: page faults of the list build under
(a) nothing else, (b) a 25 MB array allocated and freed per pass, (c) a context made and closed per pass, (d) the class path."""
import gc, os, resource, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from pydca_amd import _lib, _ranking
from pydca_amd.meanfield_dca.meanfield_dca import MeanFieldDCA
from tools.gen_msa import dedup, generate, write_fasta
X = dedup(generate(500, 50000, 21, 12346))
path = "/tmp/rank_where3.fa"
write_fasta(path, X, 21)
flt = lambda: resource.getrusage(resource.RUSAGE_SELF).ru_minflt
def build(c, tag):
    s = c.mf_run(0.5, True)
    o = c.scores_order()
    f0 = flt(); t0 = time.perf_counter(); r = _ranking.ranked(s, 500, o); t1 = time.perf_counter(); f1 = flt()
    print("%-28s list %.2f ms  %d page faults" % (tag, (t1 - t0) * 1e3, f1 - f0))
    return r
c = _lib.Context(0, _lib.DCA_F64); c.set_msa(X, 21); c.compute_weights(0.8, _lib.DCA_F64)
for rep in range(4):
    r = None; r = build(c, "bare")
for rep in range(4):
    r = None
    Y = np.empty_like(X); Y[...] = X
    r = build(c, "+ 25 MB array per pass"); del Y
for rep in range(4):
    r = None
    c2 = _lib.Context(0, _lib.DCA_F64); c2.set_msa(X, 21); c2.compute_weights(0.8, _lib.DCA_F64)
    r = build(c2, "+ context per pass"); c2.close()
orig = _ranking.ranked
def ranked(scores, L, order=None):
    f0 = flt(); t0 = time.perf_counter(); r = orig(scores, L, order); t1 = time.perf_counter(); f1 = flt()
    print("%-28s list %.2f ms  %d page faults" % ("class path", (t1 - t0) * 1e3, f1 - f0))
    return r
_ranking.ranked = ranked
for rep in range(4):
    r = m = None
    m = MeanFieldDCA(path, "protein", pseudocount=0.5, seqid=0.8, device=0)
    r = m.compute_sorted_FN_APC()
import ctypes
libc = ctypes.CDLL("libc.so.6")
print("mallopt", libc.mallopt(-1, 1 << 30), libc.mallopt(-3, 1 << 30), libc.mallopt(-2, 1 << 28))   # trim threshold, mmap threshold, top pad
for rep in range(4):
    r = m = None
    m = MeanFieldDCA(path, "protein", pseudocount=0.5, seqid=0.8, device=0)
    r = m.compute_sorted_FN_APC()
