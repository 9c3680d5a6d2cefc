"""The ranked list both classes return: [((i, j), score), ...] by descending score, ties in (i, j) order -- what Python's stable
sorted(..., reverse=True) gives the reference (meanfield_dca.py:940, plmdca.py:479).  The order comes from the device
(csrc/rank.hip) or from numpy; what is left on the host is building L(L-1)/2 tuples, which at L = 500 used to cost more than
the whole GPU chain (15 - 60 ms for 124 750 pairs).  Two things cost that time and neither is needed: the (i, j) tuples
were re-made on every call although they depend on L alone (cached here, picked in rank order by one C-level itemgetter
call), and the cyclic garbage collector ran several full passes over the quarter of a million new tuples, none of which
can be part of a cycle (paused for the construction)."""
import gc
import importlib.machinery
import importlib.util
import operator
import os
import threading

import numpy as np


def _load_fastrank():
    """csrc/fastrank.c (built by `make -C pydca_amd/csrc` into lib/_fastrank.so): the same list in one C loop; None if it is not
    there (the Python construction below is then used -- host logic either way)."""
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "lib", "_fastrank.so")
    if os.environ.get("DCA_FASTRANK") == "0" or not os.path.exists(path):
        return None
    try:
        loader = importlib.machinery.ExtensionFileLoader("_fastrank", path)
        spec = importlib.util.spec_from_loader("_fastrank", loader)
        mod = importlib.util.module_from_spec(spec)
        loader.exec_module(mod)
        return mod
    except Exception:            # a NumPy / CPython ABI the module was not built for
        return None


_fast = _load_fastrank()

_pair_tuples = {}
_seen_once = set()
_lock = threading.Lock()


def pair_tuples(L):
    """[(0, 1), (0, 2), ..., (L-2, L-1)]: the pair order of the score vectors (msa_numerics.py:220); cached per L."""
    with _lock:
        pairs = _pair_tuples.get(L)
        if pairs is None:
            if _fast is not None:
                pairs = _fast.pair_tuples(int(L))
            else:
                iu, ju = np.triu_indices(L, k=1)
                pairs = list(zip(iu.tolist(), ju.tolist()))
            if len(_pair_tuples) >= 4:
                _pair_tuples.clear()
            _pair_tuples[L] = pairs
        return pairs


def ranked(scores, L, order=None):
    """scores: the pair-ordered score vector; order: its descending stable order (device) or None (numpy).  The scores stay
    numpy scalars, as in the reference."""
    if order is None:
        order = np.argsort(-scores, kind='stable')
    if _fast is not None and L not in _pair_tuples:
        # the FIRST list of this process for L: the (i, j) tuples are made on the fly (a command line ranks once: building the
        # cache first would cost as much again); the cache is made when a second list is asked for
        with _lock:
            seen = L in _seen_once
            _seen_once.add(L)
        if not seen:
            was_enabled = gc.isenabled()
            gc.disable()
            try:
                return _fast.ranked(None, np.ascontiguousarray(order, dtype=np.int32), np.ascontiguousarray(scores, dtype=np.float64), int(L))
            finally:
                if was_enabled:
                    gc.enable()
    pairs = pair_tuples(L)
    if len(pairs) == 0:
        return []
    was_enabled = gc.isenabled()
    gc.disable()
    try:
        if _fast is not None:
            return _fast.ranked(pairs, np.ascontiguousarray(order, dtype=np.int32), np.ascontiguousarray(scores, dtype=np.float64))
        idx = order.tolist()
        picked = operator.itemgetter(*idx)(pairs) if len(idx) > 1 else (pairs[idx[0]],)
        return list(zip(picked, list(scores[order])))
    finally:
        if was_enabled:
            gc.enable()
