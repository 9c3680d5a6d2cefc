"""One process per GPU behind `PlmDCA(..., devices=[...])`, `MeanFieldDCA(..., devices=[...])` and the command lines'
`--devices 0,1,...`.

The reference's one parallel knob is a constructor / command-line argument (`num_threads`, pydca/plmdca_main.py:77-78,131,
consumed by the OpenMP loop over sites at plmdca_numerics.cpp:490).  Its counterpart here is the list of GPUs: the calling
process becomes rank 0 on devices[0] and starts one helper process per further device (`python -m pydca_amd.multi_gpu
<job directory> <rank>`); every rank gives its contexts the library's own RCCL communicators (csrc/comm_rccl.cpp: the
collectives run on the contexts' streams, over xGMI on a real node) and runs the SAME code below in lockstep.  Rank 0
created the communicators' unique ids before it started the helpers and hands them over in the job file, so no other
rendezvous (no torch.distributed, no MPI) is needed.

plmDCA: sequence weights with the comparisons divided over the ranks, then the optimisation under one of the four
exchange schemes of DESIGN.md section 6.  The scheme is chosen DETERMINISTICALLY -- the column strips (scheme 4: 1 / world
of the bytes of the others on the wires, and the scheme whose float64 gradient is bit-identical to the single-GPU run's),
scheme 2 where the strips do not come up -- so that the same command gives the same scores from run to run; `scheme=` /
DCA_EXCHANGE_SCHEME=1..4 forces one, DCA_EXCHANGE_SCHEME=auto times all four on the node at start-up like `bench.py --gpus N`
does (three iterations each, the fastest runs: float32 scores then depend on which scheme won).  The optimised parameters
end up resident in rank 0's single-GPU context, so scores, DI and parameter export are what they are on one GPU.  mfDCA: sharded weights, every rank counts a window of the sequences, ONE all-reduce of the raw
pair counts; correlation matrix, inverse and scores then run on rank 0 (the inverse is not sharded).

Failures: every helper reports 'ready' (alignment on its device, context up) before any collective starts; a helper that
fails -- then or later -- leaves its exception type and message in its status file, and rank 0 re-raises it as the
exception type the caller's class uses (PlmDCAException / MeanFieldDCAException, ValueError, FileNotFoundError).  A helper
that dies LATER leaves rank 0 inside a collective that waits for it: rank 0's share runs in a worker thread, the calling
thread notices the exit, kills the other helpers and ABORTS rank 0's communicators (dca_comm_abort -> ncclCommAbort), which
makes the pending call return an error; the caller gets the helper's error and its process lives on.  Where that does not
release the worker (no ncclCommAbort in the collective library, or the communicator was still being set up) the worker
thread is left behind after 15 s and the error is raised all the same -- never a hang, never os._exit."""
import json
import os
import shutil
import subprocess
import sys
import tempfile
import threading
import time

import numpy as np

from . import _lib, parallel

SCHEMES = {1: "sequences sharded, all-reduce(g)",
           2: "sequences sharded, reduce-scatter(g) + all-gather(x), optimiser vectors sharded",
           3: "sequences sharded, direct exchange (grouped send / recv, rank-ordered local sum), optimiser vectors sharded",
           4: "column strips: every rank all sequences x the columns of its sites, point-to-point exchange"}
ABANDON_S = 15.0     # how long rank 0's worker thread gets to come back after its communicators were aborted
NUM_IDS = 3          # communicators a run may need: weights / resident context, sequence-sharded context, column-strip context


_COMM_CONTEXTS = []      # contexts of THIS process with a live communicator (what the watchdog aborts)


def _comm_up(ctx, unique_id, world, rank, rccl):
    ctx.comm_init(unique_id, world, rank, rccl)
    _COMM_CONTEXTS.append(ctx)


def _comm_forget(ctx):
    if ctx in _COMM_CONTEXTS:
        _COMM_CONTEXTS.remove(ctx)


class MultiGpuError(RuntimeError):
    """A helper rank failed; `.kind` is the helper's exception class name, `.rank` its rank."""

    def __init__(self, rank, kind, message):
        super().__init__("rank %d: %s: %s" % (rank, kind, message))
        self.rank, self.kind, self.message = rank, kind, message


def parse_devices(spec):
    """'0,1,2' / [0, 1, 2] / None -> list of device indices or None (one GPU: the `device` argument decides).  A device may
    be named twice only over a stand-in collective library (DCA_RCCL_PATH: the tests run two ranks on one GPU that way;
    RCCL itself refuses)."""
    if spec is None or spec == "" or spec == []:
        return None
    if isinstance(spec, str):
        try:
            devs = [int(v) for v in spec.replace(" ", "").split(",") if v != ""]
        except ValueError:
            raise ValueError("--devices takes a comma-separated list of GPU indices, e.g. 0,1,2,3 (got %r)" % (spec,))
    else:
        devs = [int(v) for v in spec]
    if not devs or any(d < 0 for d in devs):
        raise ValueError("devices must be non-negative GPU indices (got %r)" % (spec,))
    if len(set(devs)) != len(devs) and not os.environ.get("DCA_RCCL_PATH"):
        raise ValueError("devices names a GPU twice (%r): one rank per GPU" % (devs,))
    if len(devs) > 64:
        raise ValueError("at most 64 ranks")
    return devs


# ------------------------------------------------------------------------------------------------ what every rank runs
def _ids(job):
    return [bytes.fromhex(h) for h in job["ids"]]


def _agree(ctx, ok):
    """True only if every rank says ok (collective over ctx's communicator)."""
    return bool(np.all(ctx.comm_allgather([1.0 if ok else 0.0]) > 0.5))


def plm_rank(job, rank, X, log=None, ctx=None):
    """The plmDCA run of one rank -> (x on rank 0 / None, stats dict, selection dict, resident context on rank 0 / None).
    ctx: this rank's context with the alignment already on the device (a helper makes it before it reports 'ready')."""
    devices, q, prec = job["devices"], job["q"], job["precision"]
    world, dev, ids, rccl = len(devices), devices[rank], _ids(job), job.get("rccl_path")
    lh, lJ, carry, seqid, cap = job["lambda_h"], job["lambda_J"], job["carry_mode"], job["seqid"], job["max_iterations"]
    dtype = np.float64 if prec == _lib.DCA_F64 else np.float32
    full = ctx if ctx is not None else _lib.Context(dev, prec)
    seq_ctx = strip_ctx = None
    try:
        if ctx is None:
            full.set_msa(X, q)
        _comm_up(full, ids[0], world, rank, rccl)
        # every rank counts 1 / world of the identity comparisons, ONE all-reduce of the N integer counts
        full.compute_weights_sharded(seqid, prec)
        counts = full.weight_counts()
        w = (dtype(1.0) / counts.astype(dtype)).astype(dtype)
        # the initial point from the whole alignment, as the single-GPU path forms it (PlmDCA::initFieldsAndCouplings)
        full.plm_configure(lh, lJ, carry)
        full.plm_init_x()
        x0 = full.plm_get_x(dtype)
        # ... and nothing else of that engine: its N x L q tables would sit beside the shard's for the whole run (rank 0 configures
        # `full` again once the run is over)
        full.plm_release()

        def seq_up():
            c = parallel.make_sharded_plm_context(_lib, X, q, w.astype(np.float64), lh, lJ, rank, world, dev,
                                                  precision=prec, carry_mode=carry, warmup=80 if prec == _lib.DCA_F64 else 40)
            _comm_up(c, ids[1], world, rank, rccl)
            return c

        def strips_up():
            c = _lib.Context(dev, prec)
            c.set_msa(X, q)
            c.set_weight_counts(counts)
            _comm_up(c, ids[2], world, rank, rccl)
            c.plm_configure_strips(lh, lJ, carry)
            return c

        def time_mode(c, mode):
            """three iterations (after an untimed one) under an exchange scheme: ms per iteration, max over the ranks; None if the
            scheme did not come up on every rank"""
            ok = True
            try:
                if mode != 4:
                    c.plm_set_native_comm(mode)
                c.plm_set_x(x0)
                c.plm_lbfgs_begin(4)
                c.plm_lbfgs_iterate(1)
            except Exception as exc:               # pragma: no cover (needs a failing collective library)
                print("pydca_amd rank %d: exchange scheme %d unavailable (%r)" % (rank, mode, exc), file=sys.stderr)
                ok = False
            if not _agree(full, ok):
                if ok:
                    c.plm_lbfgs_end()              # the run this rank began must not block the next scheme's set-up
                return None
            full.comm_allgather([0.0])               # barrier
            t0 = time.perf_counter()
            c.plm_lbfgs_iterate(3)
            ms = (time.perf_counter() - t0) / 3.0 * 1e3
            c.plm_lbfgs_end()
            return float(np.max(full.comm_allgather([ms])))

        forced = job.get("scheme")
        timings = {}
        if forced and str(forced).lower() != "auto":
            chosen = int(forced)
        elif not forced:
            # deterministic: the column strips; scheme 2 if they do not come up on every rank (e.g. fewer sites than ranks)
            ok = True
            try:
                strip_ctx = strips_up()
            except Exception as exc:
                print("pydca_amd rank %d: column strips unavailable (%r)" % (rank, exc), file=sys.stderr)
                ok = False
            chosen = 4 if _agree(full, ok) else 2
        else:
            seq_ctx = seq_up()
            for mode in (1, 2, 3):
                t = time_mode(seq_ctx, mode)
                if t is not None:
                    timings[mode] = t
            seq_ctx.close()                          # the two decompositions are never resident together
            _comm_forget(seq_ctx)
            seq_ctx = None
            strip_ctx = strips_up()
            t = time_mode(strip_ctx, 4)
            if t is not None:
                timings[4] = t
            if not timings:
                raise RuntimeError("no exchange scheme came up on every rank")
            chosen = min(timings, key=lambda m: timings[m])
        if chosen not in SCHEMES:
            raise ValueError("exchange scheme must be one of 1, 2, 3, 4 (got %r)" % (chosen,))
        if chosen == 4:
            if seq_ctx is not None:
                seq_ctx.close()
                _comm_forget(seq_ctx)
                seq_ctx = None
            run = strip_ctx if strip_ctx is not None else strips_up()
            strip_ctx = run
        else:
            if strip_ctx is not None:
                strip_ctx.close()
                _comm_forget(strip_ctx)
                strip_ctx = None
            run = seq_ctx if seq_ctx is not None else seq_up()
            seq_ctx = run
            run.plm_set_native_comm(chosen)
        run.plm_set_x(x0)
        if os.environ.get("DCA_MULTI_GPU_TEST_DIE_IN_RUN") == str(rank):        # tests: a rank that dies with every communicator up
            os._exit(9)
        t0 = time.perf_counter()
        run.plm_lbfgs_begin(cap, bool(job.get("verbose")) and rank == 0)
        st = run.plm_lbfgs_iterate(cap if cap else 1 << 30)
        seconds = time.perf_counter() - t0
        x = run.plm_get_x(dtype)                   # collective under the column strips; every rank holds all of x otherwise
        stats = dict(status=st.status, iterations=st.iterations, evaluations=st.evaluations, fx=st.fx, seconds=seconds,
                     unique_sequences=int(X.shape[0]))
        selection = dict(chosen_scheme=chosen, scheme=SCHEMES[chosen], ms_per_iteration={str(k): v for k, v in timings.items()},
                         ranks=run.comm_info()[0], devices=list(devices))
        run.close()
        _comm_forget(run)
        seq_ctx = strip_ctx = None
        if rank != 0:
            full.close()
            _comm_forget(full)
            return None, stats, selection, None
        # rank 0 keeps ONE single-GPU context with the result resident: the first one (whole alignment, all weights)
        full.comm_destroy()
        _comm_forget(full)
        full.plm_configure(lh, lJ, carry)
        full.plm_set_x(x)
        return x, stats, selection, full
    except BaseException:
        for c in (seq_ctx, strip_ctx, full):
            try:
                if c is not None:
                    _comm_forget(c)
                    c.close()
            except Exception:
                pass
        raise


def mf_rank(job, rank, X, ctx=None):
    """The sharded stages of mfDCA on one rank -> resident context with weights and the summed pair counts (rank 0) / None."""
    devices, q, seqid = job["devices"], job["q"], job["seqid"]
    world, dev, ids, rccl = len(devices), devices[rank], _ids(job), job.get("rccl_path")
    made = ctx is None
    if made:
        ctx = _lib.Context(dev, _lib.DCA_F64)
    try:
        if made:
            ctx.set_msa(X, q)
        _comm_up(ctx, ids[0], world, rank, rccl)
        if seqid < 1.0:
            ctx.compute_weights_sharded(seqid, _lib.DCA_F64)
        else:
            ctx.set_weights(np.ones(X.shape[0], dtype=np.float64))          # meanfield_dca.py:111-115
        start, stop = parallel.shard_bounds(X.shape[0], world, rank)
        ctx.mf_set_native_comm(True)
        ctx.mf_set_row_window(start, stop - start)
        ctx.mf_single_site_freqs()                 # counts of the window + ONE all-reduce of the (L q)^2 raw pair counts
        if rank != 0:
            ctx.close()
            _comm_forget(ctx)
            return None
        ctx.comm_destroy()                         # the summed counts stay (dca_mf_set_row_window)
        _comm_forget(ctx)
        return ctx
    except BaseException:
        try:
            _comm_forget(ctx)
            ctx.close()
        except Exception:
            pass
        raise


# ------------------------------------------------------------------------------------------------ rank 0's side
class _Helpers:
    """Ranks 1 .. world-1 as child processes of the calling one."""

    def __init__(self, job, X):
        self.dir = tempfile.mkdtemp(prefix="pydca_amd_ranks_")
        self.world = len(job["devices"])
        self.procs = []
        self.failed = None
        np.save(os.path.join(self.dir, "X.npy"), X)
        with open(os.path.join(self.dir, "job.json"), "w") as fh:
            json.dump(job, fh)
        env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
        env["PYTHONPATH"] = root + (os.pathsep + env["PYTHONPATH"] if env.get("PYTHONPATH") else "")
        for r in range(1, self.world):
            log = open(os.path.join(self.dir, "rank%d.log" % r), "w")
            self.procs.append((r, subprocess.Popen([sys.executable, "-m", "pydca_amd.multi_gpu", self.dir, str(r)], env=env,
                                                   stdout=log, stderr=subprocess.STDOUT), log))

    def _status(self, r):
        try:
            with open(os.path.join(self.dir, "rank%d.status" % r)) as fh:
                return json.load(fh)
        except (OSError, ValueError):
            return None

    def _error_of(self, r, proc):
        st = self._status(r) or {}
        if st.get("state") == "error":
            return MultiGpuError(r, st.get("type", "RuntimeError"), st.get("message", ""))
        tail = ""
        try:
            with open(os.path.join(self.dir, "rank%d.log" % r)) as fh:
                tail = fh.read()[-2000:]
        except OSError:
            pass
        return MultiGpuError(r, "RuntimeError", "helper exited with code %s\n%s" % (proc.returncode, tail))

    def wait_ready(self, timeout=600.0):
        """Every helper has its alignment and its device context: from here on the ranks meet in collectives."""
        t0 = time.time()
        pending = {r for r, _p, _l in self.procs}
        while pending:
            for r, p, _l in self.procs:
                if r not in pending:
                    continue
                st = self._status(r)
                if st and st.get("state") == "ready":
                    pending.discard(r)
                elif (st and st.get("state") == "error") or p.poll() is not None:
                    err = self._error_of(r, p)
                    self.abort()
                    raise err
            if time.time() - t0 > timeout:
                self.abort()
                raise MultiGpuError(min(pending), "TimeoutError", "helper did not come up within %.0f s" % timeout)
            if pending:
                time.sleep(0.02)

    def dead_helper(self):
        """-> MultiGpuError of the first helper that has exited with a failure, or None."""
        for r, p, _l in self.procs:
            if p.poll() not in (None, 0):
                return self._error_of(r, p)
        return None

    def finish(self, timeout=600.0):
        err = None
        for r, p, log in self.procs:
            try:
                p.wait(timeout=timeout)
            except subprocess.TimeoutExpired:
                p.kill()
                err = err or MultiGpuError(r, "TimeoutError", "helper did not finish")
            if p.returncode not in (0, None) and err is None:
                err = self._error_of(r, p)
            log.close()
        shutil.rmtree(self.dir, ignore_errors=True)
        if err is not None:
            raise err

    def abort(self):
        for _r, p, log in self.procs:
            if p.poll() is None:
                p.kill()
            try:
                p.wait(timeout=10)
            except Exception:
                pass
            log.close()
        shutil.rmtree(self.dir, ignore_errors=True)


def _job(kind, devices, q, seqid, **more):
    rccl = os.environ.get("DCA_RCCL_PATH") or None
    job = dict(kind=kind, devices=[int(d) for d in devices], q=int(q), seqid=float(seqid), rccl_path=rccl,
               ids=[_lib.comm_unique_id(rccl).hex() for _ in range(NUM_IDS)])
    job.update(more)
    return job


def _run(job, X, rank_fn):
    """Rank 0's share runs in a worker thread of the calling process; the calling thread watches the helpers meanwhile.  A helper
    that dies after 'ready' leaves rank 0 inside (or on its way into) a collective that waits for it: the other helpers are
    killed, rank 0's communicators aborted (dca_comm_abort -> ncclCommAbort: the pending call fails and the worker ends), and
    the helper's error is raised.  Should the worker not come back within ABANDON_S seconds all the same (a collective library
    without ncclCommAbort, or a rank that died while the communicator was still being set up), it is left behind as a
    daemon thread and the error is raised anyway: the caller is never blocked for ever and never killed."""
    helpers = _Helpers(job, X)
    box = {}

    def work():
        try:
            box["out"] = rank_fn(job, 0, X)
        except BaseException as exc:            # handed to the calling thread
            box["exc"] = exc

    try:
        helpers.wait_ready()
    except MultiGpuError:
        raise
    except BaseException:
        helpers.abort()
        raise
    worker = threading.Thread(target=work, daemon=True)
    worker.start()
    failed = None
    try:
        while worker.is_alive():
            worker.join(0.1)
            if worker.is_alive():
                failed = helpers.dead_helper()
                if failed is not None:
                    break
    except BaseException:                       # KeyboardInterrupt in the calling thread
        for c in list(_COMM_CONTEXTS):
            try:
                c.comm_abort()
            except Exception:
                pass
        helpers.abort()
        raise
    if failed is not None:
        for _r, p, _l in helpers.procs:
            if p.poll() is None:
                p.kill()
        for c in list(_COMM_CONTEXTS):
            try:
                c.comm_abort()
            except Exception:
                pass
        worker.join(ABANDON_S)
        if worker.is_alive():
            sys.stderr.write("pydca_amd: %s\npydca_amd: rank 0's collective could not be aborted; its thread is left behind.\n" % failed)
            del _COMM_CONTEXTS[:]
        helpers.abort()
        raise failed
    if "exc" in box:
        exc = box["exc"]
        if isinstance(exc, MultiGpuError):
            helpers.abort()
            raise exc
        failed = helpers.dead_helper()
        helpers.abort()
        if failed is not None:
            raise failed from exc
        raise exc
    helpers.finish()
    return box["out"]


def run_plm(X, q, seqid, lambda_h, lambda_J, max_iterations, precision, carry_mode, devices, verbose=False, scheme=None):
    """-> (x, stats, selection, resident single-GPU context on devices[0] holding x).  Called by PlmDCA with devices=[...]."""
    job = _job("plm", devices, q, seqid, lambda_h=float(lambda_h), lambda_J=float(lambda_J), max_iterations=int(max_iterations),
               precision=int(precision), carry_mode=int(carry_mode), verbose=bool(verbose),
               scheme=scheme if scheme is not None else (os.environ.get("DCA_EXCHANGE_SCHEME") or None))
    return _run(job, X, plm_rank)


def run_mf_counts(X, q, seqid, devices):
    """-> resident float64 context on devices[0]: whole alignment, weights, summed raw pair counts.  Called by MeanFieldDCA."""
    return _run(_job("mf", devices, q, seqid), X, mf_rank)


# ------------------------------------------------------------------------------------------------ a helper rank
def _helper_main(jobdir, rank):
    status = os.path.join(jobdir, "rank%d.status" % rank)

    def say(**kw):
        tmp = status + ".tmp"
        with open(tmp, "w") as fh:
            json.dump(kw, fh)
        os.replace(tmp, status)

    try:
        with open(os.path.join(jobdir, "job.json")) as fh:
            job = json.load(fh)
        X = np.load(os.path.join(jobdir, "X.npy"))
        # everything that can fail WITHOUT a peer happens before 'ready': the device exists, the library loads, the alignment
        # fits on the device
        ctx = _lib.Context(job["devices"][rank], job["precision"] if job["kind"] == "plm" else _lib.DCA_F64)
        ctx.set_msa(X, job["q"])
        say(state="ready")
        if os.environ.get("DCA_MULTI_GPU_TEST_DIE_AFTER_READY") == str(rank):      # tests: a rank that dies between collectives
            os._exit(9)
        if job["kind"] == "plm":
            plm_rank(job, rank, X, ctx=ctx)
        else:
            mf_rank(job, rank, X, ctx=ctx)
        say(state="done")
        return 0
    except BaseException as exc:
        say(state="error", type=type(exc).__name__, message=str(exc))
        import traceback
        traceback.print_exc()
        return 1


if __name__ == "__main__":
    sys.exit(_helper_main(sys.argv[1], int(sys.argv[2])))
