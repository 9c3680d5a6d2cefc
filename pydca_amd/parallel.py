"""Sequence sharding of the plmDCA evaluation across GPUs (one process per GPU).

The de-duplicated, file-ordered sequences are cut into `world` contiguous blocks.  Each
rank holds its block plus a halo of the `warmup` preceding sequences, which only warm up
the carry-over scan (see csrc/plm_engine.hip, plm_softmax_kernel) and do not contribute
to fx or g.  x is replicated; every evaluation ends with ONE exchange step: an
all-reduce(sum) of the P gradient entries and of fx over RCCL/xGMI, issued through
torch.distributed from the library's reduce hook.  The regulariser is added on rank 0
only.  Weights depend on the whole alignment, so they are computed once on the full MSA
(it is tens of MB) and the shard's slice is handed to the shard context.

On top of that the optimiser state can be sharded by parameter range (`TorchVectorComm` +
`Context.plm_set_vector_sharding`): the evaluation then ends with a reduce-scatter of the gradients,
a step with an all-gather of x, and the L-BFGS dot products are all-reduced as a few doubles --
the same bytes on the wire as the all-reduce, but the vector work (Gram pass, direction, differences)
is divided by the number of ranks.  `bench.py --gpus N` uses this mode.

mfDCA shards the same way for its one N-dependent stage, the weighted pair counts: every rank
counts its block (global weights), ONE all-reduce(sum) of the (L q)^2 raw counts and of Meff through
the same hook protocol (`Context.mf_set_reduce_hook`), after which frequencies, correlation
matrix, inverse and scores are computed identically on every rank (`make_sharded_mf_context`).

No reference counterpart: pydca is single-process.
"""
import ctypes as C

import numpy as np


def shard_bounds(n_seqs, world, rank):
    """Contiguous, balanced [start, stop) of rank's owned sequences."""
    base, rem = divmod(n_seqs, world)
    start = rank * base + min(rank, rem)
    stop = start + base + (1 if rank < rem else 0)
    return start, stop


def shard_with_halo(n_seqs, world, rank, warmup):
    """-> (first_row, stop_row, halo): rows [first_row, stop_row) go to the rank, the
    leading `halo` of them are warm-up only."""
    start, stop = shard_bounds(n_seqs, world, rank)
    halo = min(warmup, start)
    return start - halo, stop, halo


def init_native_comm(ctx, lib_mod, rank, world, dist=None, group=None, rccl_path=None):
    """Give `ctx` its own RCCL communicator (csrc/comm_rccl.cpp): rank 0 makes the unique id, it travels through the
    process group `dist` already has (any backend: it is 128 bytes), every rank joins.  After this the context's
    exchange steps run as RCCL collectives on its own stream -- no Python callback, no device synchronisation.
    Call order on every rank:  ctx.compute_weights_sharded(seqid)  (or set_weights), then  ctx.plm_configure(...),
    then  ctx.plm_set_native_comm(1 | 2 | 3)  /  ctx.mf_set_native_comm().  The exchange scheme survives later weight
    changes and re-configurations (the engines are invalidated, not dropped); a new comm_init resets it."""
    # rccl_path None: the library picks the librccl that belongs to the HIP runtime it is bound to (comm_rccl.cpp)
    path = rccl_path
    if world == 1 or dist is None:
        uid = lib_mod.comm_unique_id(path)
    else:
        box = [lib_mod.comm_unique_id(path) if rank == 0 else None]
        dist.broadcast_object_list(box, src=0, group=group)
        uid = box[0]
    ctx.comm_init(uid, world, rank, path)
    return uid


_hip = None


def _hip_rt():
    global _hip
    if _hip is None:
        _hip = C.CDLL("libamdhip64.so")
        _hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
        _hip.hipMemcpy.restype = C.c_int
        _hip.hipDeviceSynchronize.restype = C.c_int
    return _hip


def _copies_done():
    """Device-to-device hipMemcpy does not wait for the copy on the host side; the library's
    stream is non-blocking and would not wait for it either, so the hooks finish with this."""
    return _hip_rt().hipDeviceSynchronize()


_D2D = 3  # hipMemcpyDeviceToDevice


class _DevArray:
    """`__cuda_array_interface__` view of `count` elements at a raw device address, so that torch can run a
    collective directly on the library's buffer (torch.as_tensor does not copy)."""

    def __init__(self, ptr, count, dtype_bits):
        self.__cuda_array_interface__ = {"shape": (int(count),), "typestr": "<f4" if dtype_bits == 32 else "<f8",
                                         "data": (int(ptr), False), "version": 2}


def wrap_device_buffer(torch, ptr, count, dtype_bits, device):
    """torch tensor aliasing library memory, or None when this torch build cannot do it (the hooks then stage
    through their own tensors).  Set DCA_COMM_STAGED=1 to force staging."""
    import os
    if os.environ.get("DCA_COMM_STAGED") == "1":
        return None
    try:
        t = torch.as_tensor(_DevArray(ptr, count, dtype_bits), device=device)
        if t.data_ptr() != int(ptr) or t.numel() != count:
            return None
        return t
    except Exception:
        return None


class TorchAllReduceHook:
    """Reduce hook for `Context.plm_set_reduce_hook`: sums g and fx over the process group.

    The library hands over raw device pointers; torch.distributed (backend "nccl" = RCCL on ROCm) runs the
    collective in place on them through a `__cuda_array_interface__` view (`wrap_device_buffer`), or, where
    that is not available, on a staging tensor (two device-to-device copies of P elements).
    """

    def __init__(self, device, group=None):
        import torch
        import torch.distributed as dist
        self.torch, self.dist, self.group = torch, dist, group
        self.device = torch.device("cuda", device)
        self.gbuf = None
        self.fbuf = torch.zeros(1, dtype=torch.float64, device=self.device)
        self.calls = 0
        self.direct_calls = 0      # calls that ran in place on the library's buffers
        self.seconds = 0.0

    def __call__(self, g_dev, count, dtype, fx_dev):
        import time
        torch, dist = self.torch, self.dist
        t0 = time.perf_counter()
        tdt = torch.float32 if dtype == 32 else torch.float64
        hip = _hip_rt()
        nbytes = count * (4 if dtype == 32 else 8)
        direct = wrap_device_buffer(torch, g_dev, count, dtype, self.device)
        fdirect = wrap_device_buffer(torch, fx_dev, 1, 64, self.device) if direct is not None else None
        if direct is not None and fdirect is not None:
            # the collectives run in place on the library's buffers: no staging copies
            dist.all_reduce(direct, op=dist.ReduceOp.SUM, group=self.group)
            dist.all_reduce(fdirect, op=dist.ReduceOp.SUM, group=self.group)
            torch.cuda.synchronize(self.device)
            self.calls += 1
            self.direct_calls += 1
            self.seconds += time.perf_counter() - t0
            return 0
        if self.gbuf is None or self.gbuf.numel() != count or self.gbuf.dtype != tdt:
            self.gbuf = torch.empty(count, dtype=tdt, device=self.device)
        if hip.hipMemcpy(self.gbuf.data_ptr(), g_dev, nbytes, _D2D) != 0:
            return 1
        if hip.hipMemcpy(self.fbuf.data_ptr(), fx_dev, 8, _D2D) != 0:
            return 1
        dist.all_reduce(self.gbuf, op=dist.ReduceOp.SUM, group=self.group)
        dist.all_reduce(self.fbuf, op=dist.ReduceOp.SUM, group=self.group)
        torch.cuda.synchronize(self.device)
        if hip.hipMemcpy(g_dev, self.gbuf.data_ptr(), nbytes, _D2D) != 0:
            return 1
        if hip.hipMemcpy(fx_dev, self.fbuf.data_ptr(), 8, _D2D) != 0:
            return 1
        if _copies_done() != 0:
            return 1
        self.calls += 1
        self.seconds += time.perf_counter() - t0
        return 0


class TorchVectorComm:
    """Comm hook for `Context.plm_set_vector_sharding` on torch.distributed (backend "nccl" = RCCL):
    in-place all-reduce / reduce-scatter / all-gather on the library's device buffers, which torch sees through a
    `__cuda_array_interface__` view (only the rank's slice is copied); staged through one torch tensor where that
    view is not available and for the gloo self-test."""

    def __init__(self, device, rank, world, group=None):
        import torch
        import torch.distributed as dist
        self.torch, self.dist, self.group = torch, dist, group
        self.rank, self.world = rank, world
        # gloo has no reduce-scatter: emulate both slice collectives with all-reduce (self-test of the
        # multi-process path on a single GPU, see bench.py; the product path is RCCL)
        self.only_all_reduce = dist.get_backend(group) != "nccl"
        import os
        self.trace = os.environ.get("DCA_COMM_TRACE") == "1"
        self.device = torch.device("cuda", device)
        self.buf = {}
        self.calls = [0, 0, 0]
        self.direct_calls = 0      # calls that ran in place on the library's buffers
        self.seconds = 0.0

    def _tensor(self, count, dtype, slot="full"):
        tdt = self.torch.float32 if dtype == 32 else self.torch.float64
        t = self.buf.get((slot, dtype))
        if t is None or t.numel() < count:
            t = self.torch.empty(count, dtype=tdt, device=self.device)
            self.buf[(slot, dtype)] = t
        return t[:count]

    def __call__(self, op, dev, count, dtype):
        import time
        t0 = time.perf_counter()
        dist, hip = self.dist, _hip_rt()
        esz = 4 if dtype == 32 else 8
        if self.trace:
            import sys
            print("[comm rank %d] op %d count %d dtype %d" % (self.rank, op, count, dtype), file=sys.stderr, flush=True)
        direct = None if self.only_all_reduce else wrap_device_buffer(self.torch, dev, count, dtype, self.device)
        if direct is not None:
            # in place on the library's buffer: all-reduce as is; reduce-scatter into a slice buffer (its input and
            # output must not alias) and one slice-sized copy back; all-gather from a copy of the own slice
            if op == 0:
                dist.all_reduce(direct, op=dist.ReduceOp.SUM, group=self.group)
            else:
                ps = count // self.world
                lo = self.rank * ps
                mine = self._tensor(ps, dtype, slot="slice")
                if op == 1:
                    dist.reduce_scatter_tensor(mine, direct, op=dist.ReduceOp.SUM, group=self.group)
                    direct[lo:lo + ps].copy_(mine)
                else:
                    mine.copy_(direct[lo:lo + ps])
                    dist.all_gather_into_tensor(direct, mine, group=self.group)
            self.torch.cuda.synchronize(self.device)
            self.calls[op] += 1
            self.direct_calls += 1
            self.seconds += time.perf_counter() - t0
            return 0
        t = self._tensor(count, dtype)
        if op == 0:                                         # all-reduce (scalars)
            if hip.hipMemcpy(t.data_ptr(), dev, count * esz, _D2D) != 0:
                return 1
            dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group)
            self.torch.cuda.synchronize(self.device)
            if hip.hipMemcpy(dev, t.data_ptr(), count * esz, _D2D) != 0:
                return 1
        else:
            ps = count // self.world
            lo = self.rank * ps
            if self.only_all_reduce:
                if hip.hipMemcpy(t.data_ptr(), dev, count * esz, _D2D) != 0:
                    return 1
                if op == 2:                                 # all-gather as a sum of vectors that are zero off-slice
                    t[:lo].zero_()
                    t[lo + ps:].zero_()
                dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group)
                self.torch.cuda.synchronize(self.device)
                if hip.hipMemcpy(dev, t.data_ptr(), count * esz, _D2D) != 0:
                    return 1
            else:
                # separate slice buffer: no aliasing between the input and output of the collective
                mine = self._tensor(ps, dtype, slot="slice")
                if op == 1:                                 # reduce-scatter: whole vector in, own slice out
                    if hip.hipMemcpy(t.data_ptr(), dev, count * esz, _D2D) != 0:
                        return 1
                    dist.reduce_scatter_tensor(mine, t, op=dist.ReduceOp.SUM, group=self.group)
                    self.torch.cuda.synchronize(self.device)
                    if hip.hipMemcpy(dev + lo * esz, mine.data_ptr(), ps * esz, _D2D) != 0:
                        return 1
                else:                                       # all-gather: own slice in, whole vector out
                    if hip.hipMemcpy(mine.data_ptr(), dev + lo * esz, ps * esz, _D2D) != 0:
                        return 1
                    dist.all_gather_into_tensor(t, mine, group=self.group)
                    self.torch.cuda.synchronize(self.device)
                    if hip.hipMemcpy(dev, t.data_ptr(), count * esz, _D2D) != 0:
                        return 1
        if _copies_done() != 0:
            return 1
        self.calls[op] += 1
        self.seconds += time.perf_counter() - t0
        return 0


class ThreadComm:
    """The same three collectives between `world` THREADS of one process (one context each, any
    devices), staged through host memory -- the stand-in for RCCL that lets the sharded optimiser be
    exercised on a single GPU (tests/test_api_gpu.py).  One instance is shared; every thread calls
    `hook(rank)` to get its comm function."""

    def __init__(self, world):
        import threading
        self.world = world
        self.barrier = threading.Barrier(world)
        self.slots = [None] * world
        self.full = None

    def hook(self, rank):
        hip = _hip_rt()

        def comm(op, dev, count, dtype):
            dt = np.float32 if dtype == 32 else np.float64
            esz = np.dtype(dt).itemsize
            ps = count // self.world if op else 0
            lo = rank * ps
            if op in (0, 1):
                part = np.empty(count, dtype=dt)
                if hip.hipMemcpy(part.ctypes.data, dev, count * esz, 2) != 0:
                    return 1
                self.slots[rank] = part
                self.barrier.wait()
                total = self.slots[0].copy()
                for r in range(1, self.world):              # rank order: every thread gets the same bits
                    total += self.slots[r]
                self.barrier.wait()
                if op == 0:
                    return hip.hipMemcpy(dev, total.ctypes.data, count * esz, 1)
                mine = np.ascontiguousarray(total[lo:lo + ps])
                return hip.hipMemcpy(dev + lo * esz, mine.ctypes.data, ps * esz, 1)
            mine = np.empty(ps, dtype=dt)
            if hip.hipMemcpy(mine.ctypes.data, dev + lo * esz, ps * esz, 2) != 0:
                return 1
            if rank == 0:
                self.full = np.empty(count, dtype=dt)
            self.barrier.wait()
            self.full[lo:lo + ps] = mine
            self.barrier.wait()
            rc = hip.hipMemcpy(dev, self.full.ctypes.data, count * esz, 1)
            self.barrier.wait()
            return rc
        return comm


def make_sharded_plm_context(lib_mod, X, q, weights, lambda_h, lambda_J, rank, world, device,
                             precision=32, carry_mode=1, chunk=0, warmup=40):
    """Build the rank's shard context (alignment slice + halo, sliced weights, hook unset)."""
    N = X.shape[0]
    first, stop, halo = shard_with_halo(N, world, rank, warmup if carry_mode != 0 else 0)
    ctx = lib_mod.Context(device, precision)
    ctx.set_msa(np.ascontiguousarray(X[first:stop]), q)
    ctx.set_weights(np.ascontiguousarray(weights[first:stop], dtype=np.float64))
    ctx.plm_configure(lambda_h, lambda_J, carry_mode, chunk, warmup, halo, 1 if rank == 0 else 0)
    return ctx


def make_strip_plm_context(lib_mod, X, q, weights, lambda_h, lambda_J, rank, world, device, dist=None, group=None,
                           precision=32, carry_mode=1, chunk=0, warmup=0, rccl_path=None):
    """Context of one rank of the COLUMN-STRIP decomposition (csrc/plm_engine.hip, exchange mode 4): the whole alignment and
    all weights on every rank, the library's own RCCL communicator, and the columns of this rank's share of the sites.
    Per evaluation two grouped point-to-point exchanges (couplings up, gradient-table rows down) instead of a
    reduce-scatter / all-gather of whole parameter vectors; plm_get_x / plm_get_g / plm_scores are collective."""
    ctx = lib_mod.Context(device, precision)
    ctx.set_msa(np.ascontiguousarray(X), q)
    ctx.set_weights(np.ascontiguousarray(weights, dtype=np.float64))
    init_native_comm(ctx, lib_mod, rank, world, dist, group, rccl_path)
    ctx.plm_configure_strips(lambda_h, lambda_J, carry_mode, chunk, warmup)
    return ctx


def initial_x(X, weights, q, dtype=np.float32):
    """PlmDCA::initFieldsAndCouplings (plmdca_numerics.cpp:207-249) on the FULL alignment,
    host side, in `dtype` arithmetic and the reference's accumulation order (ascending n).
    Used by sharded runs, where no single context sees every sequence."""
    X = np.asarray(X)
    N, L = X.shape
    w = np.asarray(weights).astype(dtype)
    meff = dtype(0)
    for n in range(N):
        meff = dtype(meff + w[n])
    h = np.zeros((L, q), dtype=dtype)
    cols = np.arange(L)
    for n in range(N):
        h[cols, X[n]] += w[n]
    h = (h / meff).astype(dtype)
    h = np.log((h * meff + dtype(1)).astype(dtype)).astype(dtype)
    for i in range(L):
        s = dtype(0)
        for a in range(q):
            s = dtype(s + h[i, a])
        h[i] -= dtype(s / dtype(q))
    P = L * q + L * (L - 1) // 2 * q * q
    x = np.zeros(P, dtype=dtype)
    x[:L * q] = h.reshape(-1)
    return x


def make_sharded_mf_context(lib_mod, X, q, weights, rank, world, device):
    """Shard context for the mfDCA pair counts: the rank's block of the alignment (0-based codes) with
    the global weights of those sequences; set a reduce hook (e.g. TorchAllReduceHook) before
    calling mf_run / mf_corr_mat."""
    start, stop = shard_bounds(X.shape[0], world, rank)
    ctx = lib_mod.Context(device, lib_mod.DCA_F64)
    ctx.set_msa(np.ascontiguousarray(X[start:stop]), q)
    ctx.set_weights(np.ascontiguousarray(weights[start:stop], dtype=np.float64))
    return ctx
