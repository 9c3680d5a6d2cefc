"""Round 6: the ranking stage inside the class path (MeanFieldDCA from a file), with the pieces timed by wrappers."""
import gc, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from pydca_amd import _lib, _ranking
from pydca_amd.meanfield_dca import meanfield_dca as mod
from pydca_amd.meanfield_dca.meanfield_dca import MeanFieldDCA
from tools.gen_msa import dedup, generate, write_fasta
X = dedup(generate(500, 50000, 21, 12346))
path = "/tmp/rank_where2.fa"
write_fasta(path, X, 21)
del X
T = {}
orig_ranked = _ranking.ranked
def ranked(scores, L, order=None):
    t0 = time.perf_counter(); r = orig_ranked(scores, L, order); T["list"] = (time.perf_counter() - t0) * 1e3
    T["scores_type"] = (type(scores).__name__, scores.dtype.str, scores.flags["C_CONTIGUOUS"], type(order).__name__, order.dtype.str)
    return r
_ranking.ranked = ranked
orig_order = _lib.Context.scores_order
def scores_order(self):
    t0 = time.perf_counter(); o = orig_order(self); T["order"] = (time.perf_counter() - t0) * 1e3
    return o
_lib.Context.scores_order = scores_order
for rep in range(6):
    r = m = None
    t0 = time.perf_counter()
    m = MeanFieldDCA(path, "protein", pseudocount=0.5, seqid=0.8, device=0)
    r = m.compute_sorted_FN_APC()
    t1 = time.perf_counter()
    print("class path: %.1f ms  %s  %s" % ((t1 - t0) * 1e3, {k: round(v * 1e3, 2) for k, v in m.last_timings.items()}, T))
