#!/usr/bin/env python3
"""dca_plm_run (the one-call C entry, include/dca_hip.h) through ctypes, in its own process (the library binds the first librccl
it opens): one device, then devices = {0, 0} -- two ranks as two host THREADS on the one GPU of a test box over the stand-in
tests/fake_rccl/libfake_rccl.so (real RCCL refuses two ranks on one device).  Prints one JSON line."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import data_file, golden  # noqa: E402
from pydca_amd import _lib  # noqa: E402

FAKE = os.path.join(ROOT, "tests", "fake_rccl", "libfake_rccl.so")


def main():
    G = golden("plm_rf71")
    X, q = G["X"], int(G["q"])
    L = X.shape[1]
    out = {}
    # the stage API on one device: what the one call has to reproduce
    ctx = _lib.Context(0, _lib.DCA_F64)
    ctx.set_msa(X, q)
    ctx.compute_weights(0.8, _lib.DCA_F64)
    ctx.plm_configure(1.0, 20.0)
    ctx.plm_init_x()
    ctx.plm_lbfgs_begin(12)
    st_ref = ctx.plm_lbfgs_iterate(12)
    x_ref = ctx.plm_get_x(np.float64)
    ctx.close()
    kw = dict(seqid=0.8, lambda_h=1.0, lambda_J=20.0, max_iterations=12, precision=_lib.DCA_F64, dtype=np.float64)
    x1, st1 = _lib.plm_run(_lib.DCA_BIOMOLECULE_RNA, L, msa=X, **kw)
    out["one_device_bytes_equal"] = bool(np.array_equal(x1, x_ref))
    out["one_device_stats"] = [st1.status, st1.iterations, st1.evaluations] == [st_ref.status, st_ref.iterations, st_ref.evaluations]
    x2, st2 = _lib.plm_run(_lib.DCA_BIOMOLECULE_RNA, L, msa=X, devices=[0, 0], rccl_path=FAKE, **kw)
    out["two_ranks_bytes_equal"] = bool(np.array_equal(x2, x_ref))
    out["two_ranks_max_rel"] = float(np.max(np.abs(x2 - x_ref)) / np.max(np.abs(x_ref)))
    out["two_ranks_stats"] = [st2.status, st2.iterations, st2.evaluations] == [st_ref.status, st_ref.iterations, st_ref.evaluations]
    x3, st3 = _lib.plm_run(_lib.DCA_BIOMOLECULE_RNA, L, msa=X, devices=[0, 0, 0], rccl_path=FAKE, **kw)
    out["three_ranks_max_rel"] = float(np.max(np.abs(x3 - x_ref)) / np.max(np.abs(x_ref)))
    # from a file, float32 (the reference's own call: plmdcaBackend) -- same bytes as the drop-in symbol gives
    f = data_file("toy_rna.fa")
    T = golden("plm_toy_rna")
    Lt = int(T["L"])
    xf, stf = _lib.plm_run(_lib.DCA_BIOMOLECULE_RNA, Lt, msa_file=f, seqid=0.8, lambda_h=1.8, lambda_J=1.8, max_iterations=100)
    import ctypes as C
    lib = _lib.lib()
    ptr = lib.plmdcaBackend(2, 5, os.fsencode(f), Lt, 0.8, 1.8, 1.8, 100, 1, False)
    xb = np.ctypeslib.as_array(C.cast(ptr, C.POINTER(C.c_float)), shape=(xf.size,)).copy()
    lib.freeFieldsAndCouplings(ptr)
    out["file_float32_equals_dropin"] = bool(np.array_equal(xf, xb))
    out["file_status"] = stf.status
    # failures come back as codes, never as a hang: a device that does not exist as the second rank, a missing file
    for name, kwargs in (("bad_device", dict(msa=X, devices=[0, 4097], rccl_path=FAKE)), ("no_file", dict(msa_file="/nonexistent/msa.fa"))):
        try:
            _lib.plm_run(_lib.DCA_BIOMOLECULE_RNA, L, seqid=0.8, max_iterations=2, **kwargs)
            out[name] = "no error"
        except _lib.DcaBackendError as e:
            out[name] = [e.code, str(e)[:120]]
    print(json.dumps(out))


if __name__ == "__main__":
    main()
