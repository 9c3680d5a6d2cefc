"""TEST INFRASTRUCTURE ONLY -- numpy float64 restatement of the reference's mfDCA path
and of the FN / APC scoring shared by both paths.

Reference (relative to /root/reference/pydca/):
  meanfield_dca/msa_numerics.py  (weights :13-50, f_i :53-125, f_ij :182-267,
                                  corr mat :270-318, couplings :321-342)
  meanfield_dca/msa_numerics.py  (two-site fields :378-470, DI :473-533; plmDCA twin
                                  plmdca/msa_numerics.py :156-311)
  meanfield_dca/meanfield_dca.py (FN :902-943, APC :946-988, DI :793-899, fields :588-633,
                                  shift_couplings :636-658, compute_params :661-752)
  plmdca/plmdca.py               (compute_params :345-434)
  plmdca/plmdca.py               (gap stripping :246-268, FN :437-481, APC :484-524)
  fasta_reader/fasta_reader.py   (letter->int :34-45,:122-163)

Alignments here use the reference's Python coding: 1-based states, gap = q.
Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this.
"""
import numpy as np

RES_TO_INT = {
    "PROTEIN": {c: k + 1 for k, c in enumerate("ACDEFGHIKLMNPQRSTVWY")},
    "RNA": {c: k + 1 for k, c in enumerate("ACGU")},
}


def read_fasta(path):
    """Multi-line FASTA -> list of upper-cased sequence strings (what
    fasta_reader.get_alignment_from_fasta_file yields through Biopython)."""
    seqs, cur = [], None
    with open(path) as fh:
        for line in fh:
            line = line.strip()
            if not line:
                continue
            if line.startswith(">"):
                if cur is not None:
                    seqs.append("".join(cur))
                cur = []
            elif cur is not None:
                cur.append(line)
    if cur is not None:
        seqs.append("".join(cur))
    return [s.upper() for s in seqs if s]


def letter2int(seqs, biomolecule):
    """fasta_reader.alignment_letter2int :122-163: unknown letters -> gap (q); exact
    duplicates dropped keeping the first occurrence."""
    biomolecule = biomolecule.strip().upper()
    q = 21 if biomolecule == "PROTEIN" else 5
    table = np.full(256, q, dtype=np.int32)
    for ch, v in RES_TO_INT[biomolecule].items():
        table[ord(ch)] = v
    rows, seen = [], set()
    for s in seqs:
        r = table[np.frombuffer(s.upper().encode("latin-1"), dtype=np.uint8)]
        key = r.tobytes()
        if key not in seen:
            seen.add(key)
            rows.append(r)
    return np.array(rows, dtype=np.int32)


def compute_sequences_weight(alignment_data, seqid):
    """msa_numerics.py:13-50.  float64(ident)/float64(L) > seqid, self included."""
    X = np.ascontiguousarray(alignment_data)
    N, L = X.shape
    counts = np.zeros(N, dtype=np.float64)
    step = max(1, int(2e7 // max(1, N * L)))
    for s in range(0, N, step):
        ident = (X[s:s + step, None, :] == X[None, :, :]).sum(axis=2)
        counts[s:s + step] = (ident.astype(np.float64) / np.float64(L) > seqid).sum(axis=1)
    return 1.0 / counts


def compute_single_site_freqs(alignment_data, num_site_states, seqs_weight):
    """msa_numerics.py:53-89 (gap state included, column q-1)."""
    X = np.asarray(alignment_data)
    N, L = X.shape
    q = num_site_states
    meff = np.sum(seqs_weight)
    fi = np.zeros((L, q), dtype=np.float64)
    for i in range(L):
        fi[i] = np.bincount(X[:, i] - 1, weights=seqs_weight, minlength=q)[:q] / meff
    return fi


def get_reg_single_site_freqs(single_site_freqs, seqs_len, num_site_states, pseudocount):
    """msa_numerics.py:92-125 (returns a new array; the reference mutates its input)."""
    theta_by_q = np.float64(pseudocount) / np.float64(num_site_states)
    return theta_by_q + (1.0 - pseudocount) * np.asarray(single_site_freqs)


def _onehot_nogap(X, q):
    N, L = X.shape
    oh = np.zeros((N, L, q - 1), dtype=np.float64)
    n_idx, i_idx = np.nonzero(X < q)
    oh[n_idx, i_idx, X[n_idx, i_idx] - 1] = 1.0
    return oh.reshape(N, L * (q - 1))


def compute_pair_site_freqs(alignment_data, num_site_states, seqs_weight):
    """msa_numerics.py:182-229: f_ij(a,b) for i<j, non-gap a,b; pair order (0,1),(0,2),..."""
    X = np.asarray(alignment_data)
    N, L = X.shape
    q = num_site_states
    meff = np.sum(seqs_weight)
    oh = _onehot_nogap(X, q)
    full = (oh.T @ (oh * np.asarray(seqs_weight)[:, None])) / meff
    full = full.reshape(L, q - 1, L, q - 1).transpose(0, 2, 1, 3)
    iu, ju = np.triu_indices(L, k=1)
    return np.ascontiguousarray(full[iu, ju])


def get_reg_pair_site_freqs(pair_site_freqs, seqs_len, num_site_states, pseudocount):
    """msa_numerics.py:231-267."""
    theta_by_qsqrd = pseudocount / float(num_site_states * num_site_states)
    return theta_by_qsqrd + (1.0 - pseudocount) * np.asarray(pair_site_freqs)


def construct_corr_mat(reg_fi, reg_fij, seqs_len, num_site_states):
    """msa_numerics.py:270-318."""
    L, q = seqs_len, num_site_states
    qm1 = q - 1
    f = reg_fi[:, :qm1]
    C4 = np.zeros((L, L, qm1, qm1), dtype=np.float64)
    iu, ju = np.triu_indices(L, k=1)
    blocks = reg_fij - f[iu][:, :, None] * f[ju][:, None, :]
    C4[iu, ju] = blocks
    C4[ju, iu] = blocks.transpose(0, 2, 1)
    for i in range(L):
        C4[i, i] = -np.outer(f[i], f[i]) + np.diag(f[i])
    return np.ascontiguousarray(C4.transpose(0, 2, 1, 3).reshape(L * qm1, L * qm1))


def compute_couplings(corr_mat):
    """msa_numerics.py:321-342: -inv(C) (LAPACK getrf/getri like the reference)."""
    return -1.0 * np.linalg.inv(corr_mat)


def frobenius_from_blocks(blocks):
    """Shared FN formula (meanfield_dca.py:933-939, plmdca.py:467-475):
    double-centre each (q-1)x(q-1) block, Frobenius norm.  blocks: [pairs, q-1, q-1]."""
    m1 = blocks.mean(axis=1, keepdims=True)
    m2 = blocks.mean(axis=2, keepdims=True)
    m = blocks.mean(axis=(1, 2), keepdims=True)
    c = blocks - m1 - m2 + m
    return np.sqrt((c * c).sum(axis=(1, 2)))


def mf_blocks(couplings, L, q):
    qm1 = q - 1
    C4 = couplings.reshape(L, qm1, L, qm1).transpose(0, 2, 1, 3)
    iu, ju = np.triu_indices(L, k=1)
    return C4[iu, ju]


def plm_blocks(x, L, q):
    """Gap-stripped coupling blocks of a packed plmDCA vector (plmdca.py:246-268)."""
    npairs = L * (L - 1) // 2
    J = np.asarray(x)[L * q:].reshape(npairs, q, q)
    return J[:, :q - 1, :q - 1]


def apc(fn, L):
    """Average-product correction (meanfield_dca.py:968-984, plmdca.py:507-521):
    av_i = sum_{j!=i} FN_ij/(L-1); av = sum_i av_i / L; FN_ij - av_i*av_j/av."""
    iu, ju = np.triu_indices(L, k=1)
    s = np.zeros(L, dtype=fn.dtype)
    np.add.at(s, iu, fn)
    np.add.at(s, ju, fn)
    av = s / float(L - 1)
    av_all = av.sum() / float(L)
    return fn - av[iu] * (av[ju] / av_all)


def sort_scores(scores, L):
    """Stable descending sort; ties keep (i,j) lexicographic order, as Python's
    sorted(..., reverse=True) does on the reference's pair-ordered list."""
    iu, ju = np.triu_indices(L, k=1)
    order = np.argsort(-scores, kind="stable")
    return [((int(iu[k]), int(ju[k])), scores[k]) for k in order]


def mfdca_fn(alignment_data, num_site_states, pseudocount, seqid, weights=None, apc_correct=True):
    """Whole mfDCA compute_fn chain -> (scores in pair order, couplings)."""
    X = np.asarray(alignment_data)
    N, L = X.shape
    q = num_site_states
    if weights is None:
        weights = compute_sequences_weight(X, seqid) if seqid < 1.0 else np.ones(N)
    fi = get_reg_single_site_freqs(compute_single_site_freqs(X, q, weights), L, q, pseudocount)
    fij = get_reg_pair_site_freqs(compute_pair_site_freqs(X, q, weights), L, q, pseudocount)
    J = compute_couplings(construct_corr_mat(fi, fij, L, q))
    fn = frobenius_from_blocks(mf_blocks(J, L, q))
    return (apc(fn, L) if apc_correct else fn), J


def plm_fn(x, L, q, apc_correct=True, dtype=np.float64):
    fn = frobenius_from_blocks(plm_blocks(np.asarray(x, dtype=dtype), L, q))
    return apc(fn, L) if apc_correct else fn


def two_site_model_fields(blocks, reg_fi, L, q, tol=1.0e-4):
    """compute_two_site_model_fields (meanfield_dca/msa_numerics.py:378-470, plmdca twin
    :156-246): per pair, E = exp(J_ij) with the gap row/column of J zero; iterate
    h_i <- f_i / (E h_j), h_j <- f_j / (E^T h_i) (both from the OLD fields), normalise,
    until the largest absolute change is <= tol.  blocks: [pairs, q-1, q-1] (i<j order).
    Vectorised over pairs with a per-pair 'still iterating' mask so every pair stops at
    its own iteration exactly as the reference's per-pair while loop does."""
    iu, ju = np.triu_indices(L, k=1)
    P = len(iu)
    E = np.ones((P, q, q), dtype=np.float64)
    E[:, :q - 1, :q - 1] = np.exp(np.asarray(blocks, dtype=np.float64))
    fi, fj = reg_fi[iu], reg_fi[ju]
    hi = np.full((P, q), 1.0 / q)
    hj = np.full((P, q), 1.0 / q)
    active = np.arange(P)
    while active.size:
        Ea, hia, hja = E[active], hi[active], hj[active]
        xi = np.einsum("pab,pb->pa", Ea, hja)
        xj = np.einsum("pab,pa->pb", Ea, hia)
        ni = fi[active] / xi
        ni /= ni.sum(axis=1, keepdims=True)
        nj = fj[active] / xj
        nj /= nj.sum(axis=1, keepdims=True)
        change = np.maximum(np.abs(ni - hia).max(axis=1), np.abs(nj - hja).max(axis=1))
        hi[active], hj[active] = ni, nj
        active = active[change > tol]
    return E, hi, hj


def direct_info(blocks, reg_fi, L, q, fields_ij=None):
    """compute_direct_info (meanfield_dca/msa_numerics.py:473-533; plmdca twin :249-311):
    P_dir = E * h_i h_j^T / sum; DI = sum_{a,b<q-1} (P+eps) log((P+eps)/(f_i f_j+eps)).
    fields_ij [pairs, 2, q]: the caller's two-site model fields (the reference uses the array it is given)."""
    eps = 1.0e-20
    iu, ju = np.triu_indices(L, k=1)
    E, hi, hj = two_site_model_fields(blocks, reg_fi, L, q)
    if fields_ij is not None:
        hi, hj = np.asarray(fields_ij)[:, 0, :], np.asarray(fields_ij)[:, 1, :]
    pdir = E * hi[:, :, None] * hj[:, None, :]
    pdir /= pdir.sum(axis=(1, 2), keepdims=True)
    pdir += eps
    fifj = reg_fi[iu][:, :, None] * reg_fi[ju][:, None, :] + eps
    val = pdir * np.log(pdir / fifj)
    return val[:, :q - 1, :q - 1].sum(axis=(1, 2))


def mfdca_di(alignment_data, num_site_states, pseudocount, seqid, weights=None, apc_correct=False):
    """MeanFieldDCA.compute_sorted_DI[_APC] chain (meanfield_dca.py:793-899), pair order."""
    X = np.asarray(alignment_data)
    N, L = X.shape
    q = num_site_states
    if weights is None:
        weights = compute_sequences_weight(X, seqid) if seqid < 1.0 else np.ones(N)
    fi = get_reg_single_site_freqs(compute_single_site_freqs(X, q, weights), L, q, pseudocount)
    fij = get_reg_pair_site_freqs(compute_pair_site_freqs(X, q, weights), L, q, pseudocount)
    J = compute_couplings(construct_corr_mat(fi, fij, L, q))
    di = direct_info(mf_blocks(J, L, q), fi, L, q)
    return apc(di, L) if apc_correct else di


def plm_di(x, reg_fi, L, q, apc_correct=False):
    """PlmDCA.compute_sorted_DI[_APC] (plmdca.py:683-790) on a packed vector; the reference
    widens the float32 couplings to float64 before exp."""
    di = direct_info(plm_blocks(np.asarray(x), L, q).astype(np.float64), reg_fi, L, q)
    return apc(di, L) if apc_correct else di


def compute_fields(couplings, reg_fi, L, q):
    """MeanFieldDCA.compute_fields (meanfield_dca.py:588-633): per site
    log(f_i(a)/f_i(q)) - sum_{j!=i} J_ij f_j (gap state dropped) -> [L, q-1]."""
    qm1 = q - 1
    f = reg_fi[:, :qm1]
    J4 = couplings.reshape(L, qm1, L, qm1)
    total = np.einsum("iajb,jb->ia", J4, f)
    diag = np.einsum("iaib,ib->ia", J4, f)
    return np.log(f / reg_fi[:, qm1:q]) - (total - diag)


def shift_couplings(block):
    """shift_couplings (meanfield_dca.py:636-658, plmdca.py:320-342) in the dtype of `block`."""
    block = np.asarray(block)
    return block - block.mean(axis=1, keepdims=True) - block.mean(axis=0, keepdims=True) + block.mean()


def select_ranked_pairs(sorted_scores, L, linear_dist=4, num_site_pairs=None):
    """Pair selection of compute_params (meanfield_dca.py:715-745, plmdca.py:398-428) without a
    reference sequence: walk the ranked list, keep |i-j| > linear_dist, stop after num_site_pairs
    (default: L)."""
    if num_site_pairs is None:
        num_site_pairs = L
    out = []
    for (i, j), _score in sorted_scores:
        if abs(i - j) > linear_dist:
            if len(out) >= num_site_pairs:
                break
            out.append((i, j))
    return out


def mf_compute_params(couplings, reg_fi, sorted_scores, L, q, linear_dist=4, num_site_pairs=None):
    qm1 = q - 1
    fields = compute_fields(couplings, reg_fi, L, q)
    pairs = select_ranked_pairs(sorted_scores, L, linear_dist, num_site_pairs)
    blocks = [(pr, shift_couplings(couplings[pr[0] * qm1:(pr[0] + 1) * qm1, pr[1] * qm1:(pr[1] + 1) * qm1]).reshape(-1))
              for pr in pairs]
    return tuple((i, fields[i]) for i in range(L)), tuple(blocks)


def plm_compute_params(x, sorted_scores, L, q, linear_dist=4, num_site_pairs=None):
    """PlmDCA.compute_params on a packed float32 vector (fields/couplings keep its dtype)."""
    x = np.asarray(x)
    qm1 = q - 1
    h = x[:L * q].reshape(L, q)[:, :qm1]
    J = plm_blocks(x, L, q)
    iu, ju = np.triu_indices(L, k=1)
    index = {(int(a), int(b)): k for k, (a, b) in enumerate(zip(iu, ju))}
    pairs = select_ranked_pairs(sorted_scores, L, linear_dist, num_site_pairs)
    blocks = [(pr, shift_couplings(J[index[pr]]).reshape(-1)) for pr in pairs]
    return tuple((i, h[i]) for i in range(L)), tuple(blocks)
