"""MeanFieldDCA -- same class surface as pydca/meanfield_dca/meanfield_dca.py for the
`mfdca compute_fn` path; numerics on the GPU (one resident context per instance: the
alignment and the weighted pair counts stay in HBM between calls)."""
import logging
import time

import numpy as np

from .. import _lib, _ranking, multi_gpu
from ..fasta_reader import fasta_reader
from . import msa_numerics

logger = logging.getLogger(__name__)


class MeanFieldDCAException(Exception):
    """Errors related to mean-field DCA computation."""


def _ranked(scores, L, ctx=None):
    """[((i, j), score), ...] sorted by score, descending; ties keep (i, j) order, as Python's
    stable sorted(..., reverse=True) does in the reference (meanfield_dca.py:940, plmdca.py:479).
    With ctx the order comes from the device (stable radix sort of the score vector the context
    just produced, dca_scores_order); without it from numpy."""
    return _ranking.ranked(scores, L, ctx.scores_order() if ctx is not None else None)


class MeanFieldDCA:
    """Mean-field DCA (meanfield_dca.py:43-139)."""

    def __init__(self, msa, biomolecule, pseudocount=None, seqid=None, device=0, devices=None):
        self.__pseudocount = pseudocount if pseudocount is not None else 0.5
        self.__seqid = seqid if seqid is not None else 0.8
        if self.__pseudocount >= 1.0 or self.__pseudocount < 0:
            logger.error('\n\tThe relative pseudocount must satisfy 0 <= pseudocount < 1 (0.5 is customary)')
            raise ValueError
        if self.__seqid > 1.0 or self.__seqid <= 0.0:
            logger.error('\n\tThe sequence identity threshold must satisfy 0 < seqid <= 1 (0.7 to 0.9 are customary)')
            raise ValueError
        biomolecule = biomolecule.strip().upper()
        self.__msa = msa
        if biomolecule == 'RNA':
            self.__num_site_states = 5
        elif biomolecule == 'PROTEIN':
            self.__num_site_states = 21
        else:
            logger.error('\n\tBiomolecule must be PROTEIN or RNA (any case)')
            raise ValueError
        self.__sequences = None           # list-of-lists form of the reference's `alignment` property, made on demand
        self.last_timings = {}            # seconds per stage of the most recent calls (reader, weights, scores, ranking)
        t0 = time.perf_counter()
        self.__read_alignment(msa, biomolecule)
        self.__num_sequences, self.__sequences_len = (int(v) for v in self.__X0.shape)
        self.__biomolecule = biomolecule
        t1 = time.perf_counter()
        devices = multi_gpu.parse_devices(devices)          # ValueError on a malformed list, as for the other arguments
        if devices and len(devices) > 1:
            # one rank per GPU for the two stages that scale with the number of sequences: the weights (N^2 L comparisons divided
            # over the ranks) and the pair counts (a window of the sequences each, ONE all-reduce); what follows runs here
            try:
                self.__ctx = multi_gpu.run_mf_counts(self.__X0, self.__num_site_states, self.__seqid, devices)
            except multi_gpu.MultiGpuError as exc:
                logger.error('\n\tA GPU rank failed: {}'.format(exc))
                raise MeanFieldDCAException(str(exc))
            self.__sequences_weight = self.__ctx.weights()
        else:
            self.__ctx = _lib.Context(int(devices[0] if devices else device), _lib.DCA_F64)
            self.__ctx.set_msa(self.__X0, self.__num_site_states)
            if self.__seqid < 1.0:
                self.__sequences_weight = self.compute_sequences_weight()
            else:
                self.__sequences_weight = np.ones((self.__num_sequences,), dtype=np.float64)
                self.__ctx.set_weights(self.__sequences_weight)
        self.last_timings.update(reader=t1 - t0, upload_and_weights=time.perf_counter() - t1)
        self.__effective_num_sequences = np.sum(self.__sequences_weight)
        self.__couplings = None
        logger.info('\n\tCreated a MeanFieldDCA object: biomolecule {}, states {}, pseudocount {}, seqid {}, '
                    'L {}, unique sequences {}, Meff {}'.format(
                        biomolecule, self.__num_site_states, self.__pseudocount, self.__seqid,
                        self.__sequences_len, self.__num_sequences, self.__effective_num_sequences))

    def __read_alignment(self, msa, biomolecule):
        if isinstance(msa, str):
            self.__X0 = fasta_reader.get_alignment_int_array(msa, biomolecule=biomolecule, zero_based=True)   # uint8 [N', L], device coding
        elif isinstance(msa, (list, tuple)) or hasattr(msa, '__iter__'):
            # an in-memory alignment: records with a .seq attribute (Bio.Align.MultipleSeqAlignment
            # in the reference, :102-104) or plain strings
            seqs = [str(getattr(rec, 'seq', rec)).strip().upper() for rec in msa]
            self.__sequences = fasta_reader.alignment_letter2int([s for s in seqs if s], biomolecule)
            self.__X0 = np.array(self.__sequences, dtype=np.uint8) - np.uint8(1)
        else:
            raise ValueError("Alignment input parameter is invalid")

    def __str__(self):
        return '<instance of MeanFieldDCA>'

    def __call__(self, pseudocount=0.5, seqid=0.8):
        """Resets pseudocount / seqid without recomputing the weights (meanfield_dca.py:160-190)."""
        self.__pseudocount = pseudocount
        self.__seqid = seqid
        logger.warning('\n\tYou have changed one of the parameters (pseudo count or sequence identity)'
                       '\n\tpseudocount: {} \n\tsequence_identity: {}'.format(self.__pseudocount, self.__seqid))
        return None

    # ---- properties (meanfield_dca.py:193-347)
    @property
    def alignment(self):
        if self.__sequences is None:
            self.__sequences = (self.__X0 + np.uint8(1)).tolist()          # the reference's 1-based states (gap = q)
        return self.__sequences

    @property
    def biomolecule(self):
        return self.__biomolecule

    @property
    def sequences_len(self):
        return self.__sequences_len

    @property
    def num_site_states(self):
        return self.__num_site_states

    @property
    def num_sequences(self):
        return self.__num_sequences

    @property
    def sequence_identity(self):
        return self.__seqid

    @property
    def pseudocount(self):
        return self.__pseudocount

    @property
    def sequences_weight(self):
        return self.__sequences_weight

    @property
    def effective_num_sequences(self):
        return np.sum(self.__sequences_weight)

    # ---- stages (meanfield_dca.py:350-553)
    def compute_sequences_weight(self):
        logger.info('\n\tSequence weights (pairwise identity counts on the device)')
        return self.__ctx.compute_weights(self.__seqid, _lib.DCA_F64)

    def get_single_site_freqs(self):
        logger.info('\n\tWeighted single-site frequencies')
        return self.__ctx.mf_single_site_freqs()

    def get_reg_single_site_freqs(self):
        return msa_numerics.get_reg_single_site_freqs(
            single_site_freqs=self.get_single_site_freqs(), seqs_len=self.__sequences_len,
            num_site_states=self.__num_site_states, pseudocount=self.__pseudocount)

    def get_pair_site_freqs(self):
        logger.info('\n\tWeighted pair-site frequencies')
        return self.__ctx.mf_pair_site_freqs()

    def get_reg_pair_site_freqs(self):
        return msa_numerics.get_reg_pair_site_freqs(
            pair_site_freqs=self.get_pair_site_freqs(), seqs_len=self.__sequences_len,
            num_site_states=self.__num_site_states, pseudocount=self.__pseudocount)

    def construct_corr_mat(self, reg_fi, reg_fij):
        logger.info('\n\tCorrelation matrix from the regularised frequencies')
        return msa_numerics.construct_corr_mat(reg_fi=reg_fi, reg_fij=reg_fij, seqs_len=self.__sequences_len,
                                               num_site_states=self.__num_site_states)

    def compute_couplings(self, corr_mat):
        logger.info('\n\tComputing couplings')
        try:
            couplings = msa_numerics.compute_couplings(corr_mat=corr_mat)
        except Exception as e:
            logger.error('\n\tThe correlation matrix cannot be inverted ({}): pseudocount {} is too small for this alignment, raise it.'.format(
                e, self.__pseudocount))
            raise
        self.__couplings = couplings
        return couplings

    # ---- the compute_fn path, fused on the device
    def _device_scores(self, apc):
        try:
            return self.__ctx.mf_run(self.__pseudocount, apc)
        except _lib.DcaBackendError as exc:
            if exc.code == _lib.DCA_ERR_NOT_SPD:
                e = np.linalg.LinAlgError('Singular matrix')
                logger.error('\n\tThe correlation matrix cannot be inverted ({}): pseudocount {} is too small for this alignment, raise it.'.format(
                    e, self.__pseudocount))
                raise e
            raise

    def get_mapped_site_pairs_dca_scores(self, sorted_dca_scores, seqbackmapper):
        """Keeps the site pairs whose two MSA columns map to the reference sequence and renames
        them to reference positions (meanfield_dca.py:755-790, plmdca.py:527-562)."""
        mapping_dict = seqbackmapper.map_to_reference_sequence()
        self.__refseq_mapping_dict = mapping_dict
        sorted_scores_mapped = list()
        for pair, score in sorted_dca_scores:
            try:
                mapped_pair = mapping_dict[pair[0]], mapping_dict[pair[1]]
            except KeyError:
                pass
            else:
                sorted_scores_mapped.append((mapped_pair, score))
        sorted_scores_mapped = sorted(sorted_scores_mapped, key=lambda k: k[1], reverse=True)
        logger.info('\n\tSite pairs mapped onto the reference sequence: {}'.format(len(sorted_scores_mapped)))
        return tuple(sorted_scores_mapped)

    def _maybe_mapped(self, ranked, seqbackmapper):
        return ranked if seqbackmapper is None else self.get_mapped_site_pairs_dca_scores(ranked, seqbackmapper)

    def compute_sorted_FN(self, seqbackmapper=None):
        """meanfield_dca.py:902-943."""
        logger.info('\n\tFrobenius norm of every coupling block, on the device')
        return self._maybe_mapped(_ranked(self._device_scores(False), self.__sequences_len, self.__ctx), seqbackmapper)

    def compute_sorted_FN_APC(self, seqbackmapper=None):
        """meanfield_dca.py:946-988."""
        logger.info('\n\tAverage product correction of the Frobenius-norm scores')
        t0 = time.perf_counter()
        scores = self._device_scores(True)
        t1 = time.perf_counter()
        ranked = _ranked(scores, self.__sequences_len, self.__ctx)
        self.last_timings.update(counts_inverse_scores=t1 - t0, ranking=time.perf_counter() - t1)
        return self._maybe_mapped(ranked, seqbackmapper)

    def get_couplings(self):
        """-inv(C) of the current pseudocount as float64[L(q-1), L(q-1)] (device resident
        result of the last run is copied out)."""
        self.__ctx.mf_corr_mat(self.__pseudocount, want=False)
        self.__couplings = self.__ctx.mf_couplings()
        return self.__couplings

    def compute_fields(self, couplings=None):
        """meanfield_dca.py:588-633 -> {site: float64[q-1]}.  With couplings=None the couplings of
        the current pseudocount are (re)computed on the device and the field sums run there too
        (one workgroup per row of J); a caller-supplied matrix is used as given, on the host."""
        q = self.__num_site_states
        logger.info('\n\tLocal fields of the global model')
        if couplings is None:
            self._device_scores(False)
            f = self.__ctx.mf_fields()
        else:
            reg_fi = self.get_reg_single_site_freqs()
            L, qm1 = self.__sequences_len, q - 1
            J4 = np.asarray(couplings).reshape(L, qm1, L, qm1)
            p = reg_fi[:, :qm1]
            total = np.einsum('iajb,jb->ia', J4, p) - np.einsum('iaib,ib->ia', J4, p)
            f = np.log(p / reg_fi[:, qm1:q]) - total
        return {i: f[i] for i in range(self.__sequences_len)}

    def shift_couplings(self, couplings_ij):
        """meanfield_dca.py:636-658 (zero-sum gauge of one block)."""
        qm1 = self.__num_site_states - 1
        couplings_ij = np.reshape(couplings_ij, (qm1, qm1))
        avx = np.reshape(np.mean(couplings_ij, axis=1), (qm1, 1))
        avy = np.reshape(np.mean(couplings_ij, axis=0), (1, qm1))
        return couplings_ij - avx - avy + np.mean(couplings_ij)

    def compute_params(self, seqbackmapper=None, ranked_by=None, linear_dist=None, num_site_pairs=None):
        """meanfield_dca.py:661-752: fields of every site and the gauge-shifted couplings of the top
        site pairs of a ranking.  The blocks are cut and shifted on the device; only the selected
        (q-1)^2 blocks travel to the host."""
        if ranked_by is None:
            ranked_by = 'fn_apc'
        if linear_dist is None:
            linear_dist = 4
        RANKING_METHODS = ('FN', 'FN_APC', 'DI', 'DI_APC')
        ranked_by = ranked_by.strip().upper()
        if ranked_by not in RANKING_METHODS:
            logger.error('\n\tUnknown ranking {!r}; available: {}'.format(ranked_by, RANKING_METHODS))
            raise MeanFieldDCAException
        dca_scores = {'FN': self.compute_sorted_FN, 'FN_APC': self.compute_sorted_FN_APC, 'DI': self.compute_sorted_DI,
                      'DI_APC': self.compute_sorted_DI_APC}[ranked_by](seqbackmapper=seqbackmapper)
        f = self.__ctx.mf_fields()
        L = self.__sequences_len
        if seqbackmapper is not None:
            # refseq position -> MSA column (the scores above are in refseq positions)
            mapping_dict = {value: key for key, value in self.__refseq_mapping_dict.items()}
        else:
            mapping_dict = {i: i for i in range(L)}
        if num_site_pairs is None:
            num_site_pairs = len(seqbackmapper.ref_sequence) if seqbackmapper is not None else len(mapping_dict.keys())
        logger.info('\n\tExtracting fields')
        fields_mapped = [(i, f[mapping_dict[i]]) for i in mapping_dict.keys()]
        logger.info('\n\tCouplings of the best {} site pairs with |i - j| > {} in the {} ranking'.format(
            num_site_pairs, linear_dist, ranked_by))
        pairs, names = [], []
        count_pairs = 0
        for pair, _score in dca_scores:
            if abs(pair[0] - pair[1]) > linear_dist:
                count_pairs += 1
                if count_pairs > num_site_pairs:
                    break
                i, j = mapping_dict[pair[0]], mapping_dict[pair[1]]
                if i > j:
                    logger.error('\n\tSite pair out of order: i < j is required')
                    raise MeanFieldDCAException
                pairs.append((i, j))
                names.append(pair)
        if count_pairs < num_site_pairs:
            logger.warning('\n\tOnly {} ranked site pairs satisfy the distance filter; their couplings are returned.'.format(count_pairs))
        blocks = self.__ctx.mf_pair_couplings(pairs, shift=True)
        couplings_ranked = [(pair, blocks[k].reshape(-1)) for k, pair in enumerate(names)]
        return tuple(fields_mapped), tuple(couplings_ranked)

    def compute_two_site_model_fields(self, couplings, reg_fi):
        """meanfield_dca.py:556-585 -> float64[pairs, 2, q]: the fields h_i, h_j of every pair's
        two-site model, fitted to the regularised single-site frequencies (fixed point on the
        device, one workgroup per pair; pair order as in the pair-site frequencies)."""
        logger.info('\n\tFitting the two-site model fields of every site pair')
        return msa_numerics.compute_two_site_model_fields(
            couplings=couplings, reg_fi=reg_fi, seqs_len=self.__sequences_len,
            num_site_states=self.__num_site_states)

    def get_site_pair_di_score(self):
        """meanfield_dca.py:793-830 -> {(i, j): DI} for i < j, in pair order.  The chain frequencies ->
        correlation matrix -> couplings -> two-site fields -> DI stays on the device (the context keeps
        the couplings of the current pseudocount resident); only the pairs' DI values come back."""
        self._device_scores(False)
        logger.info('\n\tDirect information (DI) of every site pair, on the device')
        di = self.__ctx.mf_di_scores(False)
        iu, ju = np.triu_indices(self.__sequences_len, k=1)
        return {(int(i), int(j)): di[k] for k, (i, j) in enumerate(zip(iu, ju))}

    def compute_sorted_DI(self, seqbackmapper=None):
        """meanfield_dca.py:832-855 (two-site model fields + direct information, msa_numerics.py:378-533,
        one workgroup per site pair on the device)."""
        self._device_scores(False)
        logger.info('\n\tDirect information (DI) of every site pair, on the device')
        return self._maybe_mapped(_ranked(self.__ctx.mf_di_scores(False), self.__sequences_len, self.__ctx), seqbackmapper)

    def compute_sorted_DI_APC(self, seqbackmapper=None):
        """meanfield_dca.py:848-899."""
        self._device_scores(False)
        logger.info('\n\tAverage product correction of the DI scores')
        return self._maybe_mapped(_ranked(self.__ctx.mf_di_scores(True), self.__sequences_len, self.__ctx), seqbackmapper)
