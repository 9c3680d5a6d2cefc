"""TEST INFRASTRUCTURE ONLY -- ctypes front-end of the CPU oracle (oracle/plm_oracle.c)
and, when present, of the compiled reference (oracle/_ref/libpydca_ref.so).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this
module.  The product package `pydca_amd` must never import anything from oracle/.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "_build", "liboracle_plm.so")
_REF_PATH = os.path.join(_HERE, "_ref", "libpydca_ref.so")

_u8p = np.ctypeslib.ndpointer(np.uint8, flags="C_CONTIGUOUS")


def build(ref=True):
    """Compile the C restatement (and the reference, when its sources are present)."""
    subprocess.check_call(["make", "-C", _HERE, "-s", "all"])
    if ref:
        subprocess.check_call(["make", "-C", _HERE, "-s", "ref"])


def _load():
    if not os.path.exists(_LIB_PATH):
        build(ref=False)
    lib = C.CDLL(_LIB_PATH)
    lib.oracle_num_params.restype = C.c_size_t
    lib.oracle_num_params.argtypes = [C.c_int, C.c_int]
    lib.oracle_residue_code.restype = C.c_int
    lib.oracle_residue_code.argtypes = [C.c_int, C.c_int]
    lib.oracle_read_msa.restype = C.c_int
    lib.oracle_read_msa.argtypes = [C.c_char_p, C.c_int, C.c_int, _u8p, C.c_int, C.POINTER(C.c_int)]
    for sfx, ct, dt in (("_f32", C.c_float, np.float32), ("_f64", C.c_double, np.float64)):
        rp = np.ctypeslib.ndpointer(dt, flags="C_CONTIGUOUS")
        f = getattr(lib, "oracle_weights" + sfx)
        f.restype = None
        f.argtypes = [_u8p, C.c_int, C.c_int, ct, rp, C.c_int]
        f = getattr(lib, "oracle_meff" + sfx)
        f.restype = ct
        f.argtypes = [rp, C.c_int]
        f = getattr(lib, "oracle_init_x" + sfx)
        f.restype = None
        f.argtypes = [_u8p, rp, C.c_int, C.c_int, C.c_int, rp]
        f = getattr(lib, "oracle_gradient" + sfx)
        f.restype = ct
        f.argtypes = [_u8p, rp, C.c_int, C.c_int, C.c_int, ct, ct, rp, rp, C.c_int, C.c_int]
        f = getattr(lib, "oracle_lbfgs" + sfx)
        f.restype = C.c_int
        f.argtypes = [_u8p, rp, C.c_int, C.c_int, C.c_int, ct, ct, C.c_int, C.c_int, C.c_int,
                      rp, C.POINTER(ct), C.POINTER(C.c_int * 3), C.c_void_p, C.c_int,
                      C.c_void_p, C.c_int, C.c_void_p]
    return lib


_lib = None


def lib():
    global _lib
    if _lib is None:
        _lib = _load()
    return _lib


def _sfx(dtype):
    dtype = np.dtype(dtype)
    if dtype == np.float32:
        return "_f32", C.c_float
    if dtype == np.float64:
        return "_f64", C.c_double
    raise ValueError(dtype)


def num_params(L, q):
    return int(lib().oracle_num_params(L, q))


def read_msa(path, biomolecule, L, capacity=None):
    """-> (uint8[N', L] 0-based, gap = q-1; raw_count).  biomolecule: 1 protein, 2 RNA."""
    if capacity is None:
        with open(path, "rb") as fh:
            capacity = sum(1 for ln in fh if ln.strip() and not ln.startswith(b">")) + 1
    out = np.zeros((capacity, L), dtype=np.uint8)
    raw = C.c_int(0)
    n = lib().oracle_read_msa(os.fsencode(path), biomolecule, L, out, capacity, C.byref(raw))
    if n < 0:
        raise RuntimeError("oracle_read_msa failed with code %d" % n)
    return np.ascontiguousarray(out[:n]), raw.value


def weights(X, seqid, dtype=np.float32, threads=0):
    X = np.ascontiguousarray(X, dtype=np.uint8)
    sfx, ct = _sfx(dtype)
    w = np.zeros(X.shape[0], dtype=dtype)
    getattr(lib(), "oracle_weights" + sfx)(X, X.shape[0], X.shape[1], ct(seqid), w, threads or os.cpu_count())
    return w


def meff(w):
    sfx, _ = _sfx(w.dtype)
    return float(getattr(lib(), "oracle_meff" + sfx)(np.ascontiguousarray(w), w.shape[0]))


def init_x(X, w, q):
    X = np.ascontiguousarray(X, dtype=np.uint8)
    sfx, _ = _sfx(w.dtype)
    N, L = X.shape
    x = np.zeros(num_params(L, q), dtype=w.dtype)
    getattr(lib(), "oracle_init_x" + sfx)(X, np.ascontiguousarray(w), N, L, q, x)
    return x


def gradient(X, w, q, lambda_h, lambda_J, x, carry=True, threads=0):
    """-> (fx, g) with the reference's semantics (carry=True) or the exact gradient."""
    X = np.ascontiguousarray(X, dtype=np.uint8)
    sfx, ct = _sfx(x.dtype)
    N, L = X.shape
    g = np.zeros_like(x)
    fx = getattr(lib(), "oracle_gradient" + sfx)(
        X, np.ascontiguousarray(w, dtype=x.dtype), N, L, q, ct(lambda_h), ct(lambda_J),
        np.ascontiguousarray(x), g, int(bool(carry)), threads or os.cpu_count())
    return float(fx), g


def lbfgs(X, w, q, lambda_h, lambda_J, max_iterations, x0, carry=True, threads=0, trace_cap=0, snapshots=()):
    """-> dict(x, fx, status, iterations, evaluations, trace[, snapshots = {iteration: x after it}])."""
    X = np.ascontiguousarray(X, dtype=np.uint8)
    sfx, ct = _sfx(x0.dtype)
    N, L = X.shape
    x = np.array(x0, copy=True)
    fx = ct(0)
    stats = (C.c_int * 3)()
    trace = np.zeros((max(trace_cap, 1), 4), dtype=x0.dtype)
    snap_it = np.ascontiguousarray(sorted(int(k) for k in snapshots), dtype=np.int32)
    snap_x = np.zeros((len(snap_it), x.shape[0]), dtype=x0.dtype) if len(snap_it) else None
    getattr(lib(), "oracle_lbfgs" + sfx)(
        X, np.ascontiguousarray(w, dtype=x0.dtype), N, L, q, ct(lambda_h), ct(lambda_J),
        int(max_iterations), int(bool(carry)), threads or os.cpu_count(), x, C.byref(fx),
        C.byref(stats), trace.ctypes.data_as(C.c_void_p) if trace_cap else None, trace_cap,
        snap_it.ctypes.data_as(C.c_void_p) if len(snap_it) else None, len(snap_it),
        snap_x.ctypes.data_as(C.c_void_p) if len(snap_it) else None)
    out = dict(x=x, fx=float(fx.value), status=stats[0], iterations=stats[1],
               evaluations=stats[2], trace=trace[:min(trace_cap, stats[1])])
    if len(snap_it):
        out["snapshots"] = {int(k): snap_x[i] for i, k in enumerate(snap_it) if k <= stats[1]}
    return out


# ----------------------------------------------------------------------------
# compiled reference (only where oracle/_ref/libpydca_ref.so exists)
# ----------------------------------------------------------------------------
def have_reference():
    return os.path.exists(_REF_PATH)


class Reference:
    """The reference's own PlmDCA C++ object (plmdca/include/plmdca.h:16-84) via
    oracle/ref_driver.cpp."""

    def __init__(self, msa_file, biomolecule, L, q, seqid, lambda_h, lambda_J, threads=1):
        self._lib = C.CDLL(_REF_PATH)
        fp = np.ctypeslib.ndpointer(np.float32, flags="C_CONTIGUOUS")
        self._lib.ref_open.restype = C.c_void_p
        self._lib.ref_open.argtypes = [C.c_char_p, C.c_uint, C.c_uint, C.c_uint, C.c_float,
                                       C.c_float, C.c_float, C.c_uint]
        self._lib.ref_close.argtypes = [C.c_void_p]
        self._lib.ref_read_seqs.restype = C.c_int
        self._lib.ref_read_seqs.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int]
        self._lib.ref_weights.restype = C.c_int
        self._lib.ref_weights.argtypes = [C.c_void_p, fp, C.c_int]
        self._lib.ref_init_x.argtypes = [C.c_void_p, fp]
        self._lib.ref_gradient.restype = C.c_float
        self._lib.ref_gradient.argtypes = [C.c_void_p, fp, fp]
        self.L, self.q = L, q
        # the reference object keeps the caller's char* (plmdca.h:69), so keep it alive
        self._path = C.create_string_buffer(os.fsencode(msa_file))
        self._h = self._lib.ref_open(self._path, biomolecule, L, q, seqid,
                                     lambda_h, lambda_J, threads)
        if not self._h:
            raise RuntimeError("reference PlmDCA constructor threw")
        self.N = self._lib.ref_read_seqs(self._h, None, 0, L)

    def close(self):
        if self._h:
            self._lib.ref_close(self._h)
            self._h = None

    def seqs(self):
        out = np.zeros((self.N, self.L), dtype=np.uint8)
        self._lib.ref_read_seqs(self._h, out.ctypes.data_as(C.c_void_p), self.N, self.L)
        return out

    def weights(self):
        w = np.zeros(self.N, dtype=np.float32)
        self._lib.ref_weights(self._h, w, self.N)
        return w

    def init_x(self):
        x = np.zeros(num_params(self.L, self.q), dtype=np.float32)
        self._lib.ref_init_x(self._h, x)
        return x

    def gradient(self, x):
        g = np.zeros_like(x)
        fx = self._lib.ref_gradient(self._h, np.ascontiguousarray(x, dtype=np.float32), g)
        return float(fx), g

    def lbfgs_run(self, max_iterations):
        """The reference's own lbfgs() (lbfgs.cpp:248-644) on its own PlmDCA::gradient with the
        backend's parameters (plmdcaBackend.cpp:67-77), recorded by oracle/ref_driver.cpp:
        -> dict(x, fx, status, iterations, evaluations, trace[iterations, 5] = fx, xnorm, gnorm, step, ls)."""
        f = self._lib.ref_lbfgs_run
        fp = np.ctypeslib.ndpointer(np.float32, flags="C_CONTIGUOUS")
        f.restype = C.c_int
        f.argtypes = [C.c_void_p, C.c_int, C.c_uint, fp, C.POINTER(C.c_float), C.POINTER(C.c_int),
                      C.POINTER(C.c_int), fp, C.c_int]
        P = num_params(self.L, self.q)
        x = np.zeros(P, dtype=np.float32)
        cap = max(1, int(max_iterations))
        trace = np.zeros((cap, 5), dtype=np.float32)
        fx, it, ev = C.c_float(0), C.c_int(0), C.c_int(0)
        status = f(self._h, P, int(max_iterations), x, C.byref(fx), C.byref(it), C.byref(ev), trace, cap)
        return dict(x=x, fx=float(fx.value), status=int(status), iterations=it.value, evaluations=ev.value,
                    trace=trace[:it.value].copy())

    def backend(self, msa_file, biomolecule, seqid, lambda_h, lambda_J, max_iterations, threads=1):
        """Full run through the reference's own extern "C" plmdcaBackend
        (plmdcaBackend.cpp:151-201), bound exactly as plmdca.py:79-89 does."""
        P = num_params(self.L, self.q)
        f = self._lib.plmdcaBackend
        f.argtypes = (C.c_ushort, C.c_ushort, C.c_char_p, C.c_uint, C.c_float, C.c_float,
                      C.c_float, C.c_uint, C.c_uint, C.c_bool)
        f.restype = C.POINTER(C.c_float * P)
        ptr = f(biomolecule, self.q, os.fsencode(msa_file), self.L, seqid, lambda_h, lambda_J,
                max_iterations, threads, False)
        x = np.frombuffer(ptr.contents, dtype=np.float32).copy()
        # the reference frees a malloc'd block with delete[] (plmdcaBackend.cpp:218-222);
        # pair it with free() here instead of repeating the mismatch.
        libc = C.CDLL(None)
        libc.free.argtypes = [C.c_void_p]
        libc.free(C.cast(ptr, C.c_void_p))
        return x
