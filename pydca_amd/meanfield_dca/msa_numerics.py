"""Drop-in for pydca/meanfield_dca/msa_numerics.py: same function names, keyword
arguments, array shapes and dtypes -- the arithmetic runs on the GPU through
libdca_hip.so (kernels in csrc/weights.hip, csrc/mf_engine.hip, csrc/cholinv.hip).
Alignments use the reference's Python coding (1-based states, gap = q)."""
import numpy as np

from .. import _lib

_DEVICE = 0


def set_device(device):
    global _DEVICE
    _DEVICE = int(device)


def _ctx_for(alignment_data, num_site_states, seqs_weight=None):
    X = np.asarray(alignment_data)
    if X.ndim != 2:
        raise ValueError('alignment_data must be a 2d integer array')
    if X.min() < 1 or X.max() > num_site_states:
        raise ValueError('alignment_data must hold states 1..num_site_states')
    ctx = _lib.Context(_DEVICE, _lib.DCA_F64)
    ctx.set_msa((X - 1).astype(np.uint8), int(num_site_states))
    if seqs_weight is not None:
        ctx.set_weights(np.asarray(seqs_weight, dtype=np.float64))
    return ctx


def compute_sequences_weight(alignment_data=None, seqid=None):
    """msa_numerics.py:13-50 -> float64[N]; float64(ident)/float64(L) > seqid, self included."""
    X = np.asarray(alignment_data)
    q = int(max(2, X.max()))
    ctx = _ctx_for(X, q)
    try:
        return ctx.compute_weights(float(seqid), _lib.DCA_F64)
    finally:
        ctx.close()


def compute_single_site_freqs(alignment_data=None, num_site_states=None, seqs_weight=None):
    """msa_numerics.py:53-89 -> float64[L, q] (gap state last)."""
    ctx = _ctx_for(alignment_data, num_site_states, seqs_weight)
    try:
        return ctx.mf_single_site_freqs()
    finally:
        ctx.close()


def get_reg_single_site_freqs(single_site_freqs=None, seqs_len=None, num_site_states=None, pseudocount=None):
    """msa_numerics.py:92-125.  Like the reference this updates its argument in place
    (an L*q element-wise affine map; no kernel is warranted)."""
    theta_by_q = np.float64(pseudocount) / np.float64(num_site_states)
    single_site_freqs *= (1.0 - pseudocount)
    single_site_freqs += theta_by_q
    return single_site_freqs


def compute_pair_site_freqs(alignment_data=None, num_site_states=None, seqs_weight=None):
    """msa_numerics.py:182-229 -> float64[pairs, q-1, q-1], pair order (0,1),(0,2),..."""
    ctx = _ctx_for(alignment_data, num_site_states, seqs_weight)
    try:
        return ctx.mf_pair_site_freqs()
    finally:
        ctx.close()


def compute_pair_site_freqs_serial(alignment_data=None, num_site_states=None, seqs_weight=None):
    """msa_numerics.py:128-179, the reference's un-parallelised twin of compute_pair_site_freqs (same result, same
    pair order); here both names reach the same device kernel."""
    return compute_pair_site_freqs(alignment_data=alignment_data, num_site_states=num_site_states, seqs_weight=seqs_weight)


def get_reg_pair_site_freqs(pair_site_freqs=None, seqs_len=None, num_site_states=None, pseudocount=None):
    """msa_numerics.py:231-267 (in place, like the reference)."""
    theta_by_qsqrd = pseudocount / float(num_site_states * num_site_states)
    pair_site_freqs *= (1.0 - pseudocount)
    pair_site_freqs += theta_by_qsqrd
    return pair_site_freqs


def construct_corr_mat(reg_fi=None, reg_fij=None, seqs_len=None, num_site_states=None):
    """msa_numerics.py:270-318 -> float64[L(q-1), L(q-1)]."""
    ctx = _lib.Context(_DEVICE, _lib.DCA_F64)
    try:
        return ctx.mf_corr_from_freqs(reg_fi, reg_fij, int(seqs_len), int(num_site_states))
    finally:
        ctx.close()


def compute_couplings(corr_mat=None):
    """msa_numerics.py:321-342 -> -inv(C).  A non positive definite matrix raises
    numpy.linalg.LinAlgError('Singular matrix'), the exception the reference's
    np.linalg.inv path produces."""
    ctx = _lib.Context(_DEVICE, _lib.DCA_F64)
    try:
        inv = ctx.spd_inverse(np.asarray(corr_mat, dtype=np.float64))
    except _lib.DcaBackendError as exc:
        if exc.code == _lib.DCA_ERR_NOT_SPD:
            raise np.linalg.LinAlgError('Singular matrix')
        raise
    finally:
        ctx.close()
    return -1.0 * inv


def slice_couplings(couplings=None, site_pair=None, num_site_states=None):
    """msa_numerics.py:346-374 -> float64[q, q]: the (q-1) x (q-1) block of site pair (i, j) of the L(q-1) x L(q-1)
    couplings matrix, with a zero row and column for the gap state (a host-side slice; nothing to compute)."""
    q = int(num_site_states)
    i, j = int(site_pair[0]), int(site_pair[1])
    block = np.zeros((q, q), dtype=np.float64)
    block[:q - 1, :q - 1] = np.asarray(couplings)[i * (q - 1):(i + 1) * (q - 1), j * (q - 1):(j + 1) * (q - 1)]
    return block


def compute_two_site_model_fields(couplings=None, reg_fi=None, seqs_len=None, num_site_states=None):
    """msa_numerics.py:378-470 -> float64[pairs, 2, q]: the two-site model fields h_i, h_j of every
    pair (fixed point to 1e-4), one workgroup per pair on the device."""
    n = int(seqs_len) * (int(num_site_states) - 1)
    if np.asarray(couplings).shape != (n, n):
        raise ValueError('couplings must be the L(q-1) x L(q-1) matrix')
    ctx = _lib.Context(_DEVICE, _lib.DCA_F64)
    try:
        return ctx.di_from_arrays(couplings, 1, reg_fi, int(seqs_len), int(num_site_states), want_fields=True, want_di=False)[0]
    finally:
        ctx.close()


def compute_direct_info(couplings=None, fields_ij=None, reg_fi=None, seqs_len=None, num_site_states=None):
    """msa_numerics.py:473-533 -> float64[pairs].  fields_ij, when given, is used as it is (the reference
    does); without it the two-site model fields are computed on the device first."""
    ctx = _lib.Context(_DEVICE, _lib.DCA_F64)
    try:
        if fields_ij is not None:
            return ctx.di_from_fields(couplings, 1, reg_fi, fields_ij, int(seqs_len), int(num_site_states))
        return ctx.di_from_arrays(couplings, 1, reg_fi, int(seqs_len), int(num_site_states))[1]
    finally:
        ctx.close()
