"""ctypes binding of libdca_hip.so (include/dca_hip.h).

The product has no CPU fallback: if the shared library is missing or no gfx950 device
is visible, every compute entry point raises -- it never routes to another
implementation.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# DCA_LIB_PATH: an alternative build of the SAME library (kernel experiments); never a different backend
LIB_PATH = os.environ.get("DCA_LIB_PATH") or os.path.join(_HERE, "lib", "libdca_hip.so")

DCA_OK = 0
DCA_ERR_ARG, DCA_ERR_IO, DCA_ERR_RESIDUE = -1, -2, -3
DCA_ERR_NOMEM, DCA_ERR_HIP, DCA_ERR_NO_DEVICE, DCA_ERR_NOT_SPD, DCA_ERR_STATE = -4, -5, -6, -7, -8
DCA_F32, DCA_F64 = 32, 64
DCA_BIOMOLECULE_PROTEIN, DCA_BIOMOLECULE_RNA = 1, 2
CARRY_EXACT, CARRY_CHUNKED, CARRY_SERIAL = 0, 1, 2
PROTEIN, RNA = 1, 2


class DcaBackendError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("libdca_hip error %d: %s" % (code, msg))
        self.code = code


class PlmStats(C.Structure):
    _fields_ = [("status", C.c_int), ("iterations", C.c_int), ("evaluations", C.c_int), ("finished", C.c_int),
                ("fx", C.c_double), ("xnorm", C.c_double), ("gnorm", C.c_double), ("step", C.c_double),
                ("seconds", C.c_double)]


COMM_HOOK = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_size_t, C.c_int)
COMM_ALL_REDUCE, COMM_REDUCE_SCATTER, COMM_ALL_GATHER = 0, 1, 2
REDUCE_HOOK = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p)


def build(verbose=False):
    """Compile libdca_hip.so in-tree with hipcc for gfx950 (cross-compiles without a GPU)."""
    cmd = ["make", "-C", os.path.join(_HERE, "csrc"), "-j", str(min(8, os.cpu_count() or 1))]
    subprocess.check_call(cmd, stdout=None if verbose else subprocess.DEVNULL)
    return LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise DcaBackendError(-100, "%s not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                                    "(pydca_amd has no CPU fallback)" % LIB_PATH)
    # the SPD inverse overlaps three streams; HIP's default of four hardware queues makes the streams of a process share queues
    # (read by the runtime at its first call; the library sets the same default when it is loaded and probes its streams anyway)
    os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
    L = C.CDLL(LIB_PATH)
    vp, i, d, sz = C.c_void_p, C.c_int, C.c_double, C.c_size_t
    sig = {
        "dca_last_error": (C.c_char_p, []),
        "dca_version": (C.c_char_p, []),
        "dca_device_count": (i, []),
        "dca_release_cached_memory": (sz, []),
        "dca_read_msa": (i, [C.c_char_p, i, i, vp, i, C.POINTER(i)]),
        "dca_count_msa_lines": (i, [C.c_char_p]),
        "dca_fasta_shape": (i, [C.c_char_p, C.POINTER(i), C.POINTER(i)]),
        "dca_read_fasta": (i, [C.c_char_p, i, i, vp, i, C.POINTER(i)]),
        "dca_read_fasta_alloc": (i, [C.c_char_p, i, C.POINTER(vp), C.POINTER(i), C.POINTER(i)]),
        "dca_host_free": (None, [vp]),
        "dca_read_msa_alloc": (i, [C.c_char_p, i, i, C.POINTER(vp), C.POINTER(i)]),
        "dca_create": (i, [C.POINTER(vp), i, i]),
        "dca_destroy": (None, [vp]),
        "dca_set_msa": (i, [vp, vp, i, i, i]),
        "dca_compute_weights": (i, [vp, d, i]),
        "dca_weights_work": (i, [vp, vp]),
        "dca_compute_weights_sharded": (i, [vp, d, i]),
        "dca_weights_partial_counts": (i, [vp, d, i, i, i, vp]),
        "dca_set_weight_counts": (i, [vp, vp]),
        "dca_comm_unique_id": (i, [C.c_char_p, vp]),
        "dca_comm_init": (i, [vp, C.c_char_p, vp, i, i]),
        "dca_comm_destroy": (i, [vp]),
        "dca_comm_info": (i, [vp, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
        "dca_plm_set_native_comm": (i, [vp, i]),
        "dca_mf_set_native_comm": (i, [vp, i]),
        "dca_set_weights": (i, [vp, vp]),
        "dca_get_weights": (i, [vp, vp]),
        "dca_get_weight_counts": (i, [vp, vp]),
        "dca_get_meff": (i, [vp, C.POINTER(d)]),
        "dca_plm_configure": (i, [vp, d, d, i, i, i, i, i]),
        "dca_plm_configure_strips": (i, [vp, d, d, i, i, i]),
        "dca_plm_num_params": (sz, [i, i]),
        "dca_plm_init_x": (i, [vp]),
        "dca_plm_set_x": (i, [vp, vp, i]),
        "dca_plm_get_x": (i, [vp, vp, i]),
        "dca_plm_release": (i, [vp]),
        "dca_plm_gradient": (i, [vp, C.POINTER(d)]),
        "dca_plm_get_g": (i, [vp, vp, i]),
        "dca_plm_set_reduce_hook": (i, [vp, REDUCE_HOOK, vp]),
        "dca_mf_set_reduce_hook": (i, [vp, REDUCE_HOOK, vp]),
        "dca_di_from_arrays": (i, [vp, vp, i, vp, i, i, vp, vp]),
        "dca_di_from_fields": (i, [vp, vp, i, vp, vp, i, i, vp]),
        "dca_plm_set_vector_sharding": (i, [vp, i, i, COMM_HOOK, vp]),
        "dca_plm_lbfgs_begin": (i, [vp, i, i]),
        "dca_plm_lbfgs_iterate": (i, [vp, i, C.POINTER(PlmStats)]),
        "dca_plm_lbfgs_end": (i, [vp]),
        "dca_mf_set_row_window": (i, [vp, i, i]),
        "dca_comm_allgather_host": (i, [vp, vp, i, vp]),
        "dca_plm_scores": (i, [vp, i, vp]),
        "dca_plm_di_scores": (i, [vp, vp, i, vp]),
        "dca_mf_di_scores": (i, [vp, i, vp]),
        "dca_plm_pair_couplings": (i, [vp, vp, i, i, vp]),
        "dca_mf_fields": (i, [vp, vp]),
        "dca_mf_pair_couplings": (i, [vp, vp, i, i, vp]),
        "dca_mf_single_site_freqs": (i, [vp, vp]),
        "dca_mf_pair_site_freqs": (i, [vp, vp]),
        "dca_mf_corr_mat": (i, [vp, d, vp]),
        "dca_mf_couplings": (i, [vp, vp]),
        "dca_mf_scores": (i, [vp, i, vp]),
        "dca_mf_run": (i, [vp, d, i, vp, vp]),
        "dca_mf_corr_from_freqs": (i, [vp, vp, vp, i, i, vp]),
        "dca_spd_inverse": (i, [vp, vp, i, vp]),
        "dca_comm_abort": (i, [vp]),
        "dca_plm_run": (i, [vp, vp, i, vp]),
        "dca_scores_order": (i, [vp, vp, i]),
        "dca_sw_scores": (i, [C.c_char_p, i, C.c_char_p, vp, i, vp, i, i, vp]),
        "dca_sw_align": (i, [C.c_char_p, i, C.c_char_p, i, vp, i, i, C.POINTER(i), C.POINTER(i), C.POINTER(i), C.c_char_p,
                             C.c_char_p, C.POINTER(i)]),
        "dca_set_profiling": (i, [vp, i]),
        "dca_set_profiling_only": (i, [vp, C.c_char_p]),
        "dca_get_kernel_time": (i, [vp, C.c_char_p, C.POINTER(d), C.POINTER(i)]),
        "dca_reset_kernel_times": (i, [vp]),
        "plmdcaBackend": (C.c_void_p, [C.c_ushort, C.c_ushort, C.c_char_p, C.c_uint, C.c_float, C.c_float, C.c_float,
                                       C.c_uint, C.c_uint, C.c_bool]),
        "freeFieldsAndCouplings": (None, [vp]),
    }
    for name, (res, args) in sig.items():
        f = getattr(L, name)
        f.restype = res
        f.argtypes = args
    _lib = L
    return L


EXPORTS = ["dca_weights_work", "dca_compute_weights_sharded", "dca_weights_partial_counts", "dca_set_weight_counts", "dca_comm_unique_id",
           "dca_comm_init", "dca_comm_destroy", "dca_comm_abort", "dca_comm_info", "dca_plm_set_native_comm", "dca_mf_set_native_comm",
           "dca_last_error", "dca_version", "dca_device_count", "dca_release_cached_memory", "dca_read_msa", "dca_count_msa_lines", "dca_read_msa_alloc", "dca_mf_set_row_window", "dca_comm_allgather_host", "dca_fasta_shape", "dca_read_fasta", "dca_read_fasta_alloc", "dca_host_free", "dca_create",
           "dca_destroy", "dca_set_msa", "dca_compute_weights", "dca_set_weights", "dca_get_weights",
           "dca_get_weight_counts", "dca_get_meff", "dca_plm_configure", "dca_plm_configure_strips", "dca_plm_num_params", "dca_plm_init_x",
           "dca_plm_set_x", "dca_plm_get_x", "dca_plm_release", "dca_plm_gradient", "dca_plm_get_g", "dca_plm_set_reduce_hook", "dca_mf_set_reduce_hook", "dca_di_from_arrays", "dca_di_from_fields", "dca_plm_set_vector_sharding",
           "dca_plm_lbfgs_begin", "dca_plm_lbfgs_iterate", "dca_plm_lbfgs_end", "dca_plm_scores", "dca_plm_di_scores",
           "dca_mf_di_scores", "dca_plm_pair_couplings", "dca_mf_fields", "dca_mf_pair_couplings",
           "dca_mf_single_site_freqs",
           "dca_mf_pair_site_freqs", "dca_mf_corr_mat", "dca_mf_couplings", "dca_mf_scores", "dca_mf_run",
           "dca_mf_corr_from_freqs", "dca_spd_inverse", "dca_sw_scores", "dca_sw_align", "dca_scores_order", "dca_set_profiling", "dca_set_profiling_only", "dca_get_kernel_time",
           "dca_reset_kernel_times", "dca_plm_run", "plmdcaBackend", "freeFieldsAndCouplings"]


class PlmArgs(C.Structure):
    """dca_plm_args (include/dca_hip.h)"""
    _fields_ = [("biomolecule", C.c_int), ("msa_file", C.c_char_p), ("msa", C.c_void_p), ("num_seqs", C.c_int), ("seqs_len", C.c_int),
                ("seqid", C.c_float), ("lambda_h", C.c_float), ("lambda_J", C.c_float), ("max_iterations", C.c_int), ("precision", C.c_int),
                ("devices", C.POINTER(C.c_int)), ("num_devices", C.c_int), ("exchange_scheme", C.c_int), ("rccl_path", C.c_char_p),
                ("verbose", C.c_int)]


def plm_run(biomolecule, seqs_len, msa_file=None, msa=None, seqid=0.8, lambda_h=1.0, lambda_J=20.0, max_iterations=100, precision=DCA_F32,
            devices=None, exchange_scheme=0, rccl_path=None, verbose=False, dtype=np.float32):
    """dca_plm_run, the library's one-call plmDCA entry (what a C host would call) -> (x, PlmStats).  msa: uint8[N, L] codes."""
    q = 21 if biomolecule == DCA_BIOMOLECULE_PROTEIN else 5
    a = PlmArgs()
    a.biomolecule, a.seqs_len = int(biomolecule), int(seqs_len)
    a.msa_file = os.fsencode(msa_file) if msa_file else None
    keep = None
    if msa is not None:
        keep = np.ascontiguousarray(msa, dtype=np.uint8)
        a.msa, a.num_seqs = keep.ctypes.data, int(keep.shape[0])
    a.seqid, a.lambda_h, a.lambda_J = float(seqid), float(lambda_h), float(lambda_J)
    a.max_iterations, a.precision, a.exchange_scheme, a.verbose = int(max_iterations), int(precision), int(exchange_scheme), int(bool(verbose))
    devs = (C.c_int * len(devices))(*devices) if devices else None
    a.devices, a.num_devices = devs, len(devices) if devices else 0
    a.rccl_path = os.fsencode(rccl_path) if rccl_path else None
    L = int(seqs_len)
    x = np.zeros(L * q + L * (L - 1) // 2 * q * q, dtype=dtype)
    st = PlmStats()
    check(lib().dca_plm_run(C.byref(a), _ptr(x), DCA_F64 if np.dtype(dtype) == np.float64 else DCA_F32, C.byref(st)))
    del keep
    return x, st


def comm_unique_id(rccl_path=None):
    """128-byte RCCL unique id (rank 0 makes it, every rank passes it to Context.comm_init)."""
    buf = C.create_string_buffer(128)
    check(lib().dca_comm_unique_id(os.fsencode(rccl_path) if rccl_path else None, buf))
    return buf.raw


def release_cached_memory():
    """Return the library's cached device blocks (>= 1 MiB, kept across contexts) to the driver -> bytes released."""
    return int(lib().dca_release_cached_memory())


def check(rc):
    if rc != DCA_OK:
        raise DcaBackendError(rc, lib().dca_last_error().decode("utf-8", "replace"))


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p)


def read_msa(path, biomolecule, L):
    """Reference C++ reader semantics (plmdca_numerics.cpp:685-767) -> (uint8[N',L], raw_count)."""
    l = lib()
    rows, raw = C.c_void_p(), C.c_int(0)
    n = l.dca_read_msa_alloc(os.fsencode(path), int(biomolecule), int(L), C.byref(rows), C.byref(raw))     # one pass over the file
    try:
        if n < 0:
            check(n)
        out = np.empty((n, int(L)), dtype=np.uint8)
        if n:
            C.memmove(out.ctypes.data, rows, out.nbytes)
        return out, raw.value
    finally:
        if rows:
            l.dca_host_free(rows)


def fasta_shape(path):
    """(records with residues, their common length) of a FASTA file, natively (dca_fasta_shape)."""
    l = lib()
    n, L = C.c_int(0), C.c_int(0)
    rc = l.dca_fasta_shape(os.fsencode(path), C.byref(n), C.byref(L))
    if rc == DCA_ERR_ARG:
        raise ValueError(l.dca_last_error().decode("utf-8", "replace"))
    check(rc)
    return n.value, L.value


def read_fasta(path, biomolecule):
    """The mfDCA path's FASTA semantics (fasta_reader.py:81-163) natively -> (uint8[N', L] with 0-based codes, gap = q-1;
    number of records read).  Raises DcaBackendError(DCA_ERR_RESIDUE) for files with non-ASCII bytes (the caller then
    reads in text mode itself), ValueError for empty files / unequal lengths like the Python reader."""
    l = lib()
    rows, L, raw = C.c_void_p(), C.c_int(0), C.c_int(0)
    k = l.dca_read_fasta_alloc(os.fsencode(path), int(biomolecule), C.byref(rows), C.byref(L), C.byref(raw))     # one pass
    try:
        if k == DCA_ERR_ARG:
            raise ValueError(l.dca_last_error().decode("utf-8", "replace"))
        if k < 0:
            check(k)
        if raw.value == 0:
            raise ValueError("No sequences found in %s" % path)
        out = np.empty((k, L.value), dtype=np.uint8)
        C.memmove(out.ctypes.data, rows, out.nbytes)
    finally:
        if rows:
            l.dca_host_free(rows)
    return out, raw.value


class Context:
    """One GPU, one stream, one alignment (include/dca_hip.h `dca_ctx`)."""

    def __init__(self, device=0, precision=DCA_F32):
        self._l = lib()
        self._h = C.c_void_p()
        check(self._l.dca_create(C.byref(self._h), int(device), int(precision)))
        self.precision = precision
        self.N = self.L = self.q = 0
        self._hook = None

    def close(self):
        if self._h:
            self._l.dca_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- alignment / weights
    def set_msa(self, X, q):
        X = np.ascontiguousarray(X, dtype=np.uint8)
        self.N, self.L, self.q = X.shape[0], X.shape[1], int(q)
        check(self._l.dca_set_msa(self._h, _ptr(X), self.N, self.L, self.q))

    def compute_weights(self, seqid, compare_precision=DCA_F32):
        check(self._l.dca_compute_weights(self._h, float(seqid), int(compare_precision)))
        return self.weights()

    def weights_work(self):
        """(wave x 32-site groups the last compute_weights launch compared, the same without the early exit, bit planes per group)"""
        out = np.zeros(3, dtype=np.uint64)
        check(self._l.dca_weights_work(self._h, _ptr(out)))
        return int(out[0]), int(out[1]), int(out[2])

    def set_weights(self, w):
        w = np.ascontiguousarray(w, dtype=np.float64)
        assert w.shape == (self.N,)
        check(self._l.dca_set_weights(self._h, _ptr(w)))

    def compute_weights_sharded(self, seqid, compare_precision=DCA_F32):
        """Weights with the comparisons divided over the ranks of the context's communicator (comm_init first).
        Same default compare precision as compute_weights, so the two give the same counts at threshold ties."""
        check(self._l.dca_compute_weights_sharded(self._h, float(seqid), int(compare_precision)))
        return self.weights()

    def weights_partial_counts(self, seqid, compare_precision, part, parts):
        """Part `part` of `parts` of the identity comparisons -> partial integer counts (sum the parts, then set_weight_counts)."""
        c = np.zeros(self.N, dtype=np.uint32)
        check(self._l.dca_weights_partial_counts(self._h, float(seqid), int(compare_precision), int(part), int(parts), _ptr(c)))
        return c

    def set_weight_counts(self, counts):
        counts = np.ascontiguousarray(counts, dtype=np.uint32)
        if counts.shape != (self.N,):
            raise ValueError("counts must have one entry per sequence")
        check(self._l.dca_set_weight_counts(self._h, _ptr(counts)))

    # ---- native collectives (RCCL on the context's stream)
    def comm_init(self, unique_id, world, rank, rccl_path=None):
        self._comm_id = bytes(unique_id)
        check(self._l.dca_comm_init(self._h, os.fsencode(rccl_path) if rccl_path else None, self._comm_id, int(world), int(rank)))

    def comm_destroy(self):
        check(self._l.dca_comm_destroy(self._h))

    def comm_abort(self):
        """From a watchdog thread: a peer died -- make this context's pending collectives fail (ncclCommAbort)."""
        check(self._l.dca_comm_abort(self._h))

    def comm_info(self):
        """(world, rank) as the communicator itself reports them (ncclCommCount / ncclCommUserRank)."""
        w, r = C.c_int(0), C.c_int(0)
        check(self._l.dca_comm_info(self._h, C.byref(w), C.byref(r)))
        return w.value, r.value

    def plm_set_native_comm(self, mode):
        """0 off, 1 all-reduce of g and fx per evaluation, 2 sharded optimiser vectors (RCCL reduce-scatter / all-gather),
        3 sharded optimiser vectors by direct exchange (grouped send / recv + rank-ordered local sum)."""
        check(self._l.dca_plm_set_native_comm(self._h, int(mode)))

    def mf_set_row_window(self, first, count):
        """Count only the sequences [first, first + count) of the (whole) alignment this context holds; count < 0: all."""
        check(self._l.dca_mf_set_row_window(self._h, int(first), int(count)))

    def comm_allgather(self, values):
        """Every rank's `values` (a few doubles) on every rank, in rank order: array [world, len(values)].  Collective."""
        mine = np.ascontiguousarray(values, dtype=np.float64).reshape(-1)
        world = self.comm_info()[0]
        out = np.zeros((world, mine.size), dtype=np.float64)
        check(self._l.dca_comm_allgather_host(self._h, _ptr(mine), mine.size, _ptr(out)))
        return out

    def mf_set_native_comm(self, on=True):
        check(self._l.dca_mf_set_native_comm(self._h, int(bool(on))))

    def weights(self):
        w = np.zeros(self.N, dtype=np.float64)
        check(self._l.dca_get_weights(self._h, _ptr(w)))
        return w

    def weight_counts(self):
        c = np.zeros(self.N, dtype=np.uint32)
        check(self._l.dca_get_weight_counts(self._h, _ptr(c)))
        return c

    def meff(self):
        v = C.c_double(0)
        check(self._l.dca_get_meff(self._h, C.byref(v)))
        return v.value

    # ---- plmDCA
    def num_params(self):
        return int(self._l.dca_plm_num_params(self.L, self.q))

    def plm_configure(self, lambda_h, lambda_J, carry_mode=CARRY_CHUNKED, chunk=0, warmup=0, halo=0, add_regulariser=1):
        check(self._l.dca_plm_configure(self._h, float(lambda_h), float(lambda_J), int(carry_mode), int(chunk),
                                        int(warmup), int(halo), int(add_regulariser)))

    def plm_configure_strips(self, lambda_h, lambda_J, carry_mode=CARRY_CHUNKED, chunk=0, warmup=0):
        """Column-strip decomposition over the context's communicator (comm_init first; the context holds the whole
        alignment): this rank takes the columns of its share of the sites.  plm_get_x / plm_get_g / plm_scores are collective."""
        check(self._l.dca_plm_configure_strips(self._h, float(lambda_h), float(lambda_J), int(carry_mode), int(chunk), int(warmup)))

    def plm_init_x(self):
        check(self._l.dca_plm_init_x(self._h))

    def plm_set_x(self, x):
        x = np.ascontiguousarray(x)
        dt = DCA_F32 if x.dtype == np.float32 else DCA_F64
        if dt == DCA_F64:
            x = np.ascontiguousarray(x, dtype=np.float64)
        check(self._l.dca_plm_set_x(self._h, _ptr(x), dt))

    def plm_get_x(self, dtype=np.float32):
        x = np.zeros(self.num_params(), dtype=dtype)
        check(self._l.dca_plm_get_x(self._h, _ptr(x), DCA_F32 if x.dtype == np.float32 else DCA_F64))
        return x

    def plm_release(self):
        """Frees the plmDCA engine of this context (tables, vectors); alignment, weights and communicator stay."""
        check(self._l.dca_plm_release(self._h))

    def plm_get_g(self, dtype=np.float32):
        g = np.zeros(self.num_params(), dtype=dtype)
        check(self._l.dca_plm_get_g(self._h, _ptr(g), DCA_F32 if g.dtype == np.float32 else DCA_F64))
        return g

    def plm_gradient(self):
        fx = C.c_double(0)
        check(self._l.dca_plm_gradient(self._h, C.byref(fx)))
        return fx.value

    def plm_set_reduce_hook(self, pyfunc):
        """pyfunc(g_dev_ptr:int, count:int, dtype:int, fx_dev_ptr:int) -> 0 on success."""
        def tramp(user, g_dev, count, dtype, fx_dev):
            try:
                return int(pyfunc(g_dev, count, dtype, fx_dev) or 0)
            except Exception:   # never let an exception cross the C boundary
                import traceback
                traceback.print_exc()
                return 1
        self._hook = REDUCE_HOOK(tramp) if pyfunc is not None else C.cast(None, REDUCE_HOOK)
        check(self._l.dca_plm_set_reduce_hook(self._h, self._hook, None))

    def plm_set_vector_sharding(self, rank, world, pyfunc):
        """Shard the optimiser's P-vectors over `world` ranks; pyfunc(op, buf_dev_ptr, count, dtype) -> 0
        runs the collectives (COMM_ALL_REDUCE / COMM_REDUCE_SCATTER / COMM_ALL_GATHER, in place on
        device memory).  pyfunc=None switches back to replicated vectors."""
        def tramp(user, op, buf, count, dtype):
            try:
                return int(pyfunc(op, buf, count, dtype) or 0)
            except Exception:   # never let an exception cross the C boundary
                import traceback
                traceback.print_exc()
                return 1
        self._comm = COMM_HOOK(tramp) if pyfunc is not None else C.cast(None, COMM_HOOK)
        check(self._l.dca_plm_set_vector_sharding(self._h, int(rank), int(world), self._comm, None))

    def mf_set_reduce_hook(self, pyfunc):
        """Same hook protocol for the mfDCA pair counts: pyfunc(craw_dev_ptr, count, 64, meff_dev_ptr)."""
        def tramp(user, g_dev, count, dtype, fx_dev):
            try:
                return int(pyfunc(g_dev, count, dtype, fx_dev) or 0)
            except Exception:   # never let an exception cross the C boundary
                import traceback
                traceback.print_exc()
                return 1
        self._mf_hook = REDUCE_HOOK(tramp) if pyfunc is not None else C.cast(None, REDUCE_HOOK)
        check(self._l.dca_mf_set_reduce_hook(self._h, self._mf_hook, None))

    def plm_lbfgs_begin(self, max_iterations, verbose=False):
        check(self._l.dca_plm_lbfgs_begin(self._h, int(max_iterations), int(bool(verbose))))

    def plm_lbfgs_end(self):
        check(self._l.dca_plm_lbfgs_end(self._h))

    def plm_lbfgs_iterate(self, iterations):
        st = PlmStats()
        check(self._l.dca_plm_lbfgs_iterate(self._h, int(iterations), C.byref(st)))
        return st

    def plm_scores(self, apc=True):
        out = np.zeros(self.L * (self.L - 1) // 2, dtype=np.float64)
        check(self._l.dca_plm_scores(self._h, int(bool(apc)), _ptr(out)))
        return out

    def plm_di_scores(self, reg_fi, apc=False):
        reg_fi = np.ascontiguousarray(reg_fi, dtype=np.float64)
        if reg_fi.shape != (self.L, self.q):
            raise ValueError("reg_fi must be L x q")
        out = np.zeros(self.L * (self.L - 1) // 2, dtype=np.float64)
        check(self._l.dca_plm_di_scores(self._h, _ptr(reg_fi), int(bool(apc)), _ptr(out)))
        return out

    def di_from_arrays(self, couplings, layout, reg_fi, L, q, want_fields=False, want_di=True):
        """layout 1: n x n couplings matrix; 2: gap-stripped 1-D blocks.  -> (fields [pairs,2,q] | None, di [pairs] | None)"""
        couplings = np.ascontiguousarray(couplings, dtype=np.float64)
        reg_fi = np.ascontiguousarray(reg_fi, dtype=np.float64)
        npairs = L * (L - 1) // 2
        fields = np.zeros((npairs, 2, q), dtype=np.float64) if want_fields else None
        di = np.zeros(npairs, dtype=np.float64) if want_di else None
        check(self._l.dca_di_from_arrays(self._h, _ptr(couplings), int(layout), _ptr(reg_fi), int(L), int(q),
                                         _ptr(fields) if want_fields else None, _ptr(di) if want_di else None))
        return fields, di

    def di_from_fields(self, couplings, layout, reg_fi, fields_ij, L, q):
        """DI from the caller's two-site model fields [pairs, 2, q] (no fixed point is run)."""
        couplings = np.ascontiguousarray(couplings, dtype=np.float64)
        reg_fi = np.ascontiguousarray(reg_fi, dtype=np.float64)
        npairs = L * (L - 1) // 2
        fields_ij = np.ascontiguousarray(fields_ij, dtype=np.float64)
        if fields_ij.shape != (npairs, 2, q):
            raise ValueError("fields_ij must have shape (L(L-1)/2, 2, q)")
        di = np.zeros(npairs, dtype=np.float64)
        check(self._l.dca_di_from_fields(self._h, _ptr(couplings), int(layout), _ptr(reg_fi), _ptr(fields_ij), int(L), int(q), _ptr(di)))
        return di

    def scores_order(self):
        """Pair indices of the last score vector, best first (device radix sort, stable)."""
        out = np.zeros(self.L * (self.L - 1) // 2, dtype=np.int32)
        check(self._l.dca_scores_order(self._h, _ptr(out), int(out.size)))
        return out

    def _pair_couplings(self, fn, pairs, shift):
        pairs = np.ascontiguousarray(pairs, dtype=np.int32).reshape(-1, 2)
        qm = self.q - 1
        out = np.zeros((pairs.shape[0], qm, qm), dtype=np.float64)
        if pairs.shape[0]:
            check(fn(self._h, _ptr(pairs), int(pairs.shape[0]), int(bool(shift)), _ptr(out)))
        return out

    def plm_pair_couplings(self, pairs, shift=True):
        return self._pair_couplings(self._l.dca_plm_pair_couplings, pairs, shift)

    # ---- mfDCA
    def mf_pair_couplings(self, pairs, shift=True):
        return self._pair_couplings(self._l.dca_mf_pair_couplings, pairs, shift)

    def mf_fields(self):
        out = np.zeros((self.L, self.q - 1), dtype=np.float64)
        check(self._l.dca_mf_fields(self._h, _ptr(out)))
        return out

    def mf_di_scores(self, apc=False):
        out = np.zeros(self.L * (self.L - 1) // 2, dtype=np.float64)
        check(self._l.dca_mf_di_scores(self._h, int(bool(apc)), _ptr(out)))
        return out

    def mf_single_site_freqs(self):
        out = np.zeros((self.L, self.q), dtype=np.float64)
        check(self._l.dca_mf_single_site_freqs(self._h, _ptr(out)))
        return out

    def mf_pair_site_freqs(self):
        qm = self.q - 1
        out = np.zeros((self.L * (self.L - 1) // 2, qm, qm), dtype=np.float64)
        check(self._l.dca_mf_pair_site_freqs(self._h, _ptr(out)))
        return out

    def mf_corr_mat(self, pseudocount, want=True):
        n = self.L * (self.q - 1)
        out = np.zeros((n, n), dtype=np.float64) if want else None
        check(self._l.dca_mf_corr_mat(self._h, float(pseudocount), _ptr(out) if want else None))
        return out

    def mf_couplings(self, want=True):
        n = self.L * (self.q - 1)
        out = np.zeros((n, n), dtype=np.float64) if want else None
        check(self._l.dca_mf_couplings(self._h, _ptr(out) if want else None))
        return out

    def mf_scores(self, apc=True):
        out = np.zeros(self.L * (self.L - 1) // 2, dtype=np.float64)
        check(self._l.dca_mf_scores(self._h, int(bool(apc)), _ptr(out)))
        return out

    def mf_run(self, pseudocount, apc=True, want_couplings=False):
        n = self.L * (self.q - 1)
        scores = np.zeros(self.L * (self.L - 1) // 2, dtype=np.float64)
        J = np.zeros((n, n), dtype=np.float64) if want_couplings else None
        check(self._l.dca_mf_run(self._h, float(pseudocount), int(bool(apc)), _ptr(scores),
                                 _ptr(J) if want_couplings else None))
        return (scores, J) if want_couplings else scores

    def mf_corr_from_freqs(self, reg_fi, reg_fij, L, q):
        reg_fi = np.ascontiguousarray(reg_fi, dtype=np.float64)
        reg_fij = np.ascontiguousarray(reg_fij, dtype=np.float64)
        n = L * (q - 1)
        out = np.zeros((n, n), dtype=np.float64)
        check(self._l.dca_mf_corr_from_freqs(self._h, _ptr(reg_fi), _ptr(reg_fij), int(L), int(q), _ptr(out)))
        return out

    def spd_inverse(self, A):
        A = np.ascontiguousarray(A, dtype=np.float64)
        out = np.zeros_like(A)
        check(self._l.dca_spd_inverse(self._h, _ptr(A), A.shape[0], _ptr(out)))
        return out

    # ---- timing
    def set_profiling(self, on=True):
        check(self._l.dca_set_profiling(self._h, int(bool(on))))

    def set_profiling_only(self, stage):
        """Clock ONE stage only (None: profiling off): two event records per launch of it instead of two per stage."""
        check(self._l.dca_set_profiling_only(self._h, stage.encode() if stage else None))

    def reset_kernel_times(self):
        check(self._l.dca_reset_kernel_times(self._h))

    def kernel_time(self, tag):
        ms, n = C.c_double(0), C.c_int(0)
        check(self._l.dca_get_kernel_time(self._h, tag.encode(), C.byref(ms), C.byref(n)))
        return ms.value, n.value


def sw_scores(ref, seqs, sub, gap_open, gap_extend):
    """Best local alignment score of `ref` against every string of `seqs` (host code in libdca_hip.so)."""
    L = lib()
    enc = [s.encode("ascii") for s in seqs]
    offs = np.zeros(len(enc) + 1, dtype=np.int32)
    offs[1:] = np.cumsum([len(e) for e in enc])
    blob = b"".join(enc)
    sub = np.ascontiguousarray(sub, dtype=np.int32)
    out = np.zeros(len(enc), dtype=np.int32)
    r = ref.encode("ascii")
    check(L.dca_sw_scores(r, len(r), blob, _ptr(offs), len(enc), _ptr(sub), int(gap_open), int(gap_extend), _ptr(out)))
    return out


def sw_align(a, b, sub, gap_open, gap_extend):
    """One optimal local alignment -> (aligned_a, aligned_b, score, start_a, start_b)."""
    L = lib()
    ea, eb = a.encode("ascii"), b.encode("ascii")
    sub = np.ascontiguousarray(sub, dtype=np.int32)
    cap = len(ea) + len(eb) + 1
    ba, bb = C.create_string_buffer(cap), C.create_string_buffer(cap)
    score, sa, sb, n = C.c_int(), C.c_int(), C.c_int(), C.c_int()
    check(L.dca_sw_align(ea, len(ea), eb, len(eb), _ptr(sub), int(gap_open), int(gap_extend), C.byref(score), C.byref(sa),
                         C.byref(sb), ba, bb, C.byref(n)))
    return ba.value.decode("ascii"), bb.value.decode("ascii"), score.value, sa.value, sb.value
