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
            logger.error('SequenceBackmapper needs msa_file or alignment_data')
            raise ValueError
        if refseq_file:
            self.__ref_sequence = self._reference_sequence(refseq_file=refseq_file)
        elif ref_seq:
            self.__ref_sequence = ref_seq.strip().upper()
        else:
            logger.error('SequenceBackmapper needs ref_seq or refseq_file')
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
                logger.error('the reference sequence may hold standard residues only (no gaps, no ambiguity codes)')
                raise ValueError
        return None

    def _reference_sequence(self, refseq_file):
        """sequence_backmapper.py:155-183: first record of the file."""
        logger.info('reference sequence from {}'.format(refseq_file))
        ref_seqs = fasta_reader.get_alignment_char_form(refseq_file, biomolecule=self.__biomolecule)
        ref_sequence = ref_seqs[0]
        if len(ref_seqs) > 1:
            logger.warning('{} holds several sequences; the first one is the reference'.format(os.path.basename(refseq_file)))
        if not ref_sequence:
            logger.error('the reference sequence file holds no sequence')
            raise ValueError
        return ref_sequence.strip().upper()

    def _scoring(self):
        if self.__biomolecule not in scoring_matrix.MATRICES:
            logger.error('no substitution matrix for biomolecule {}'.format(self.__biomolecule))
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
        logger.info('looking for the alignment row closest to the reference sequence')
        first = self.__alignment[0]
        if first.replace('-', '') == self.__ref_sequence:
            logger.info('the first row, without its gaps, is the reference sequence: no search needed')
            return [first]
        sub, gap_open, gap_extend = self._scoring()
        scores = _lib.sw_scores(self.__ref_sequence, [s.replace('-', '') for s in self.__alignment], sub, gap_open, gap_extend)
        max_score = scores.max()
        best = [self.__alignment[k] for k in range(len(self.__alignment)) if scores[k] == max_score]
        if len(best) > 1:
            logger.warning('{} rows match the reference equally well; the first one is used'.format(len(best)))
        return best

    @staticmethod
    def _thread_through_row(aligned_ref, row_tail, n_template):
        """Lays the reference side of a pairwise alignment (`aligned_ref`: residues and '-') along a stretch of the
        template's MSA row (`row_tail`) and yields one character per produced column: the reference residue that
        falls on that column, or '-'.

        The rule is the reference implementation's (sequence_backmapper.py:286-336), kept with its corner cases
        because the mapping is defined by it (pinned by tests/golden/backmap_cases.json, which the reference itself
        produced): a residue column of the row takes the next alignment column whatever it holds; a gap column of the
        row swallows one alignment column only if that column is a gap on the reference side; the walk stops once
        `n_template` residues of the row or the whole alignment have been used, and whatever is left of the
        alignment is emitted unchanged.  An alignment exhausted by a swallowed gap raises IndexError, as there."""
        cursor, used = 0, 0
        for site in row_tail:
            if used == n_template:
                break
            here = aligned_ref[cursor]                  # IndexError when a swallowed gap exhausted the alignment
            if site == '-':
                yield '-'
                cursor += here == '-'
                continue
            yield here
            cursor += 1
            used += 1
            if cursor == len(aligned_ref):
                break
        yield from aligned_ref[cursor:]

    @staticmethod
    def align_subsequences(ref_middle_subseq=None, template_subseq_in_msa=None, num_res_middle_template=None):
        """Public helper of the reference class (sequence_backmapper.py:286-336): the reference part of the local
        alignment re-spaced with the MSA gaps of the template row, as a string."""
        return ''.join(SequenceBackmapper._thread_through_row(ref_middle_subseq, template_subseq_in_msa,
                                                              num_res_middle_template))

    @staticmethod
    def _column_after_residues(row, count):
        """First column of `row` at which `count` residues lie to the left (0 when the row has no such column --
        the reference's loop leaves its start index at 0 then, sequence_backmapper.py:403-412)."""
        seen = 0
        for column, site in enumerate(row):
            if seen == count:
                return column
            seen += site != '-'
        return 0

    def map_to_reference_sequence(self):
        """-> {MSA column: position in the reference sequence} (sequence_backmapper.py:339-466).

        The row of the alignment that matches the reference best is aligned locally against the reference; the columns
        of that local alignment are then laid along the row (`_thread_through_row`) starting at the first column
        that has all of the row's unaligned leading residues to its left, and every column that received a reference
        residue is mapped to that residue's position, counted from the first aligned one."""
        logger.info('\n\tMapping the reference sequence onto the alignment columns')
        row = self.find_matching_seqs_from_alignment()[0]
        ref_full, template_full, _score, begin, end = self.align_pairs_local(self.__ref_sequence, row.replace('-', ''))[0]
        residues = lambda text: len(text) - text.count('-')            # noqa: E731
        first_column = self._column_after_residues(row, residues(template_full[:begin]))
        ref_position = residues(ref_full[:begin])
        placed = list(self._thread_through_row(ref_full[begin:end], row[first_column:], residues(template_full[begin:end])))
        column_to_position = {}
        for column, char in zip(range(first_column, len(row)), placed):
            if char != '-':
                column_to_position[column] = ref_position
                ref_position += 1
        logger.info('\n\t{} of the {} residues of the reference sequence have a column'.format(
            len(column_to_position), len(self.__ref_sequence)))
        return column_to_position
