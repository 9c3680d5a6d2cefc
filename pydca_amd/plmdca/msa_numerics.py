"""Drop-in for the direct-information part of pydca/plmdca/msa_numerics.py (:156-311): same function
names and keyword arguments; couplings are the gap-stripped 1-D array of
PlmDCA.get_couplings_no_gap_state (pair order, (q-1)^2 values per pair).  The frequency functions of
that module are the ones of the mean-field module and are re-exported from there."""
import numpy as np

from .. import _lib
from ..meanfield_dca.msa_numerics import (compute_sequences_weight as _weights, compute_single_site_freqs,  # noqa: F401
                                          get_reg_single_site_freqs)

_DEVICE = 0


def set_device(device):
    global _DEVICE
    _DEVICE = int(device)


def compute_sequences_weight(alignment_data=None, sequence_identity=None):
    """plmdca/msa_numerics.py:13-49 (keyword `sequence_identity`, float64 comparison)."""
    return _weights(alignment_data=alignment_data, seqid=sequence_identity)


def _check(couplings, seqs_len, num_site_states):
    L, qm1 = int(seqs_len), int(num_site_states) - 1
    c = np.asarray(couplings, dtype=np.float64).reshape(-1)
    if c.size != L * (L - 1) // 2 * qm1 * qm1:
        raise ValueError('couplings must hold (q-1)^2 values for each of the L(L-1)/2 pairs')
    return c


def slice_couplings(couplings=None, site_pair=None, num_site_states=None, seqs_len=None):
    """plmdca/msa_numerics.py:128-152 -> float64[q, q]: the (q-1)^2 gap-stripped couplings of pair (i, j), i < j, cut from
    the 1-D array in pair order, with a zero row and column for the gap state."""
    L, q = int(seqs_len), int(num_site_states)
    i, j = int(site_pair[0]), int(site_pair[1])
    pair = L * (L - 1) // 2 - (L - i) * (L - i - 1) // 2 + j - i - 1
    block = np.zeros((q, q), dtype=np.float64)
    block[:q - 1, :q - 1] = np.asarray(couplings).reshape(-1)[pair * (q - 1) ** 2:(pair + 1) * (q - 1) ** 2].reshape(q - 1, q - 1)
    return block


def compute_two_site_model_fields(couplings=None, reg_fi=None, seqs_len=None, num_site_states=None):
    """plmdca/msa_numerics.py:156-246 -> float64[pairs, 2, q]."""
    c = _check(couplings, seqs_len, num_site_states)
    ctx = _lib.Context(_DEVICE, _lib.DCA_F64)
    try:
        return ctx.di_from_arrays(c, 2, reg_fi, int(seqs_len), int(num_site_states), want_fields=True, want_di=False)[0]
    finally:
        ctx.close()


def compute_direct_info(couplings=None, fields_ij=None, reg_fi=None, seqs_len=None, num_site_states=None):
    """plmdca/msa_numerics.py:249-311 -> float64[pairs].  fields_ij, when given, is used as it is (the
    reference does); without it the two-site model fields are computed on the device first."""
    c = _check(couplings, seqs_len, num_site_states)
    ctx = _lib.Context(_DEVICE, _lib.DCA_F64)
    try:
        if fields_ij is not None:
            return ctx.di_from_fields(c, 2, reg_fi, fields_ij, int(seqs_len), int(num_site_states))
        return ctx.di_from_arrays(c, 2, reg_fi, int(seqs_len), int(num_site_states))[1]
    finally:
        ctx.close()
