"""Round 6: what the upload of a FRESH 25 MB alignment costs (config D) -- pageable as the reader returns it, registered in place,
from pinned memory -- and what the class path's 'upload_and_weights' stage consists of."""
import ctypes, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from pydca_amd import _lib
from tools.gen_msa import dedup, generate
hip = ctypes.CDLL("libamdhip64.so")
X = dedup(generate(500, 50000, 21, 12346))
ms = lambda a, b: (b - a) * 1e3
for rep in range(4):
    Y = X.copy()
    t0 = time.perf_counter(); c = _lib.Context(0, _lib.DCA_F64); t1 = time.perf_counter()
    c.set_msa(Y, 21); t2 = time.perf_counter()
    w = c.compute_weights(0.8, _lib.DCA_F64); t3 = time.perf_counter()
    c.close(); t4 = time.perf_counter()
    print("fresh pageable: create %.2f set_msa %.2f weights %.2f close %.2f ms" % (ms(t0, t1), ms(t1, t2), ms(t2, t3), ms(t3, t4)))
for rep in range(3):
    Y = X.copy()
    t0 = time.perf_counter()
    rc = hip.hipHostRegister(ctypes.c_void_p(Y.ctypes.data), ctypes.c_size_t(Y.nbytes), 0); t1 = time.perf_counter()
    c = _lib.Context(0, _lib.DCA_F64); c.set_msa(Y, 21); t2 = time.perf_counter()
    hip.hipHostUnregister(ctypes.c_void_p(Y.ctypes.data)); t3 = time.perf_counter()
    c.close()
    print("registered in place: register %.2f (rc %d) create + set_msa %.2f unregister %.2f ms" % (ms(t0, t1), rc, ms(t1, t2), ms(t2, t3)))
for rep in range(3):
    p = ctypes.c_void_p()
    t0 = time.perf_counter(); rc = hip.hipHostMalloc(ctypes.byref(p), ctypes.c_size_t(X.nbytes), 0); t1 = time.perf_counter()
    Y = np.ctypeslib.as_array(ctypes.cast(p, ctypes.POINTER(ctypes.c_uint8)), shape=(X.nbytes,)).reshape(X.shape)
    Y[...] = X; t2 = time.perf_counter()
    c = _lib.Context(0, _lib.DCA_F64); c.set_msa(Y, 21); t3 = time.perf_counter()
    c.close(); del Y
    t4 = time.perf_counter(); hip.hipHostFree(p); t5 = time.perf_counter()
    print("pinned: hipHostMalloc %.2f (rc %d) fill %.2f create + set_msa %.2f free %.2f ms" % (ms(t0, t1), rc, ms(t1, t2), ms(t2, t3), ms(t4, t5)))
from pydca_amd.meanfield_dca.meanfield_dca import MeanFieldDCA
from tools.gen_msa import write_fasta
path = "/tmp/upload_paths.fa"
write_fasta(path, X, 21)
for rep in range(6):
    r = m = None
    t0 = time.perf_counter()
    m = MeanFieldDCA(path, "protein", pseudocount=0.5, seqid=0.8, device=0)
    r = m.compute_sorted_FN_APC()
    t1 = time.perf_counter()
    print("class path: %.1f ms  %s" % (ms(t0, t1), {k: round(v * 1e3, 2) for k, v in m.last_timings.items()}))
