"""SequenceBackmapper -- mirror of pydca/sequence_backmapper/sequence_backmapper.py:22-466: finds the
MSA row that matches a reference sequence best (local alignment score of the reference against
every gap-stripped row), aligns the two and maps MSA columns to reference positions.

The reference delegates the alignments to Bio.pairwise2.align.localds (biopython, absent from the
reference tree and from this image); here they run in libdca_hip.so (dca_sw_scores / dca_sw_align,
Smith-Waterman with affine gaps, same matrices and penalties).  pairwise2 returns a list of
co-optimal alignments in an undocumented order and the reference takes the first; where several
optimal alignments exist this module's choice (include/dca_hip.h) may differ -- parity for this
row of the scope table is unpinned, exactly as the reference's own test only checks that more
than one site is mapped (tests/sequence_backmapper_test.py:38-42).
"""
import logging
import os

from .. import _lib
from ..fasta_reader import fasta_reader
from . import scoring_matrix

logger = logging.getLogger(__name__)


class SequenceBackmapper:
    """sequence_backmapper.py:22-78."""

    def __init__(self, msa_file=None, alignment_data=None, ref_seq=None, refseq_file=None, biomolecule=None):
        self.__biomolecule = biomolecule.strip().upper()
        if msa_file:
            self.__alignment = fasta_reader.get_alignment_char_form(msa_file, biomolecule=self.__biomolecule)
        elif alignment_data is not None and len(alignment_data):
            unique_seqs = []
            seen = set()
            for seq in alignment_data:
                key = tuple(seq)
                if key not in seen:
                    seen.add(key)
                    unique_seqs.append(seq)
            self.__alignment = fasta_reader.sequences_to_char_form(unique_seqs, self.__biomolecule)
        else:
            logger.error('\n\tPlease provide alignment file or a list of alignments')
            raise ValueError
        if refseq_file:
            self.__ref_sequence = self._reference_sequence(refseq_file=refseq_file)
        elif ref_seq:
            self.__ref_sequence = ref_seq.strip().upper()
        else:
            logger.error('\n\tPlease provide a reference sequence or a FASTA file containing a reference sequence')
            raise ValueError
        self._validate_refseq()

    @property
    def alignment(self):
        return self.__alignment

    @property
    def ref_sequence(self):
        return self.__ref_sequence

    def __str__(self):
        return '<A sequence backmapper object of biomolecule type {}>'.format(self.__biomolecule)

    def _validate_refseq(self):
        """sequence_backmapper.py:128-152: standard residues only, no gaps."""
        standard = [s for s in fasta_reader.RES_TO_INT_ALL[self.__biomolecule].keys() if s not in ('-', '.', '~')]
        for res in self.__ref_sequence:
            if res not in standard:
                logger.error('\n\tReference sequence should only contain standard residues')
                raise ValueError
        return None

    def _reference_sequence(self, refseq_file):
        """sequence_backmapper.py:155-183: first record of the file."""
        logger.info('\n\tObtaining reference sequence from file:\n\t\t{}'.format(refseq_file))
        ref_seqs = fasta_reader.get_alignment_char_form(refseq_file, biomolecule=self.__biomolecule)
        ref_sequence = ref_seqs[0]
        if len(ref_seqs) > 1:
            logger.warning('\n\tFound multiple reference sequences in file {}.\n\tFirst sequence taken as reference'.format(
                os.path.basename(refseq_file)))
        if not ref_sequence:
            logger.error('\n\tNo reference sequence found')
            raise ValueError
        return ref_sequence.strip().upper()

    def _scoring(self):
        if self.__biomolecule not in scoring_matrix.MATRICES:
            logger.error('\n\tUnknown biomolecule type. Cannot figure out the scoring matrix.')
            raise ValueError
        return (scoring_matrix.MATRICES[self.__biomolecule],) + scoring_matrix.GAP_PENALTIES[self.__biomolecule]

    def align_pairs_local(self, ref_seq, other_seq, score_only=False):
        """sequence_backmapper.py:186-230.  Returns the score, or a one-element list holding
        (ref_aligned, other_aligned, score, begin, end) in pairwise2's layout: both sequences
        in full, unaligned ends sharing columns and padded with '-', [begin, end) the columns of
        the local alignment."""
        sub, gap_open, gap_extend = self._scoring()
        if score_only:
            return float(_lib.sw_scores(ref_seq, [other_seq], sub, gap_open, gap_extend)[0])
        mid_a, mid_b, score, sa, sb = _lib.sw_align(ref_seq, other_seq, sub, gap_open, gap_extend)
        pre = max(sa, sb)
        head_a = '-' * (pre - sa) + ref_seq[:sa]
        head_b = '-' * (pre - sb) + other_seq[:sb]
        ea = sa + len(mid_a.replace('-', ''))
        eb = sb + len(mid_b.replace('-', ''))
        tail_a, tail_b = ref_seq[ea:], other_seq[eb:]
        post = max(len(tail_a), len(tail_b))
        full_a = head_a + mid_a + tail_a + '-' * (post - len(tail_a))
        full_b = head_b + mid_b + tail_b + '-' * (post - len(tail_b))
        return [(full_a, full_b, float(score), pre, pre + len(mid_a))]

    def find_matching_seqs_from_alignment(self):
        """sequence_backmapper.py:233-283: rows with the highest local score (all of them, MSA order)."""
        logger.info('\n\tSearching for sequence(s) that match best with the reference sequence')
        first = self.__alignment[0]
        if first.replace('-', '') == self.__ref_sequence:
            logger.info('\n\tFirst sequence in alignment (gaps removed) matches reference,'
                        '\n\tSkipping regorous search for matching sequence')
            return [first]
        sub, gap_open, gap_extend = self._scoring()
        scores = _lib.sw_scores(self.__ref_sequence, [s.replace('-', '') for s in self.__alignment], sub, gap_open, gap_extend)
        max_score = scores.max()
        best = [self.__alignment[k] for k in range(len(self.__alignment)) if scores[k] == max_score]
        if len(best) > 1:
            logger.warning('\n\tFound {} sequences in MSA that match the reference'
                           '\n\tThe first sequence is taken as matching'.format(len(best)))
        return best

    @staticmethod
    def align_subsequences(ref_middle_subseq=None, template_subseq_in_msa=None, num_res_middle_template=None):
        """sequence_backmapper.py:286-336: copy the template's MSA gaps into the reference part."""
        mapped_ref_subseq = []
        res_count = 0
        pos = 0
        for site in template_subseq_in_msa:
            if res_count == num_res_middle_template:
                break
            if site != '-':
                mapped_ref_subseq.append(ref_middle_subseq[pos])
                pos += 1
                res_count += 1
                if pos == len(ref_middle_subseq):
                    break
            else:
                if ref_middle_subseq[pos] != '-':
                    mapped_ref_subseq.append('-')
                else:
                    mapped_ref_subseq.append(ref_middle_subseq[pos])
                    pos += 1
        mapped_ref_subseq.extend(list(ref_middle_subseq[pos:]))
        return ''.join(mapped_ref_subseq)

    def map_to_reference_sequence(self):
        """sequence_backmapper.py:339-466 -> {MSA column: reference position}."""
        logger.info('\n\tBackmapping reference sequence to MSA')
        template_seq_in_msa = self.find_matching_seqs_from_alignment()[0]
        template_gaps_removed = template_seq_in_msa.replace('-', '')
        ref_aligned, template_aligned, _score, start_indx, end_indx = self.align_pairs_local(
            self.__ref_sequence, template_gaps_removed)[0]
        ref_middle_subseq = ref_aligned[start_indx:end_indx]
        template_middle_subseq = template_aligned[start_indx:end_indx]
        num_leading_res_template = len(template_aligned[:start_indx].replace('-', ''))
        num_leading_res_ref = len(ref_aligned[:start_indx].replace('-', ''))
        num_res_middle_template = len(template_middle_subseq.replace('-', ''))
        res_count = 0
        start_indx_in_msa = 0
        for k, site in enumerate(template_seq_in_msa):
            if res_count == num_leading_res_template:
                start_indx_in_msa = k
                break
            if site != '-':
                res_count += 1
        template_subseq_in_msa = template_seq_in_msa[start_indx_in_msa:]
        backmapped_ref_subseq = self.align_subsequences(
            ref_middle_subseq=ref_middle_subseq, template_subseq_in_msa=template_subseq_in_msa,
            num_res_middle_template=num_res_middle_template)
        mapped_sites = dict()
        mapped_res_count = 0
        for k, site in enumerate(backmapped_ref_subseq):
            if k == len(template_seq_in_msa) - start_indx_in_msa:
                break
            if site != '-':
                mapped_sites[mapped_res_count + num_leading_res_ref] = start_indx_in_msa + k
                mapped_res_count += 1
        logger.info('\n\tNumber of residues mapped: {}\n\tNumber of residues in the (original) reference sequence: {}'.format(
            len(mapped_sites), len(self.__ref_sequence)))
        return {value: key for key, value in mapped_sites.items()}
