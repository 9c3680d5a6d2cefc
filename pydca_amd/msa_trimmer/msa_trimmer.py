"""MSATrimmer -- mirror of pydca/msa_trimmer/msa_trimmer.py:15-207: columns to drop from an MSA by gap
fraction, or with respect to the row that matches a reference sequence.  Host-side O(N L) work."""
import logging
from collections import namedtuple

import numpy as np

from ..sequence_backmapper.sequence_backmapper import SequenceBackmapper

logger = logging.getLogger(__name__)

SeqRecord = namedtuple('SeqRecord', ['id', 'seq'])


class MSATrimmerException(Exception):
    """Raises exceptions related to MSA trimming"""


def read_fasta_records(file_name):
    """(id, sequence) records of a FASTA file as Bio.AlignIO.read(..., 'fasta') yields them:
    id = header up to the first blank, multi-line sequences joined, case kept."""
    records, name, cur = [], None, []
    with open(file_name) as fh:
        for line in fh:
            line = line.strip()
            if not line:
                continue
            if line.startswith('>'):
                if name is not None:
                    records.append(SeqRecord(name, ''.join(cur)))
                fields = line[1:].split()
                name, cur = (fields[0] if fields else ''), []
            elif name is not None:
                cur.append(line)
    if name is not None:
        records.append(SeqRecord(name, ''.join(cur)))
    if not records:
        raise ValueError('No records found in handle')
    if len(set(len(r.seq) for r in records)) != 1:
        raise ValueError('Sequences must all be the same length')
    return records


class MSATrimmer:
    def __init__(self, msa_file, biomolecule=None, max_gap=None, refseq_file=None):
        """msa_trimmer.py:17-50."""
        self.__msa_file = msa_file
        self.__refseq_file = refseq_file
        self.__max_gap = 0.5 if max_gap is None else max_gap
        if self.__max_gap > 1.0 or self.__max_gap < 0.0:
            logger.error('max_gap is a fraction: 0 <= max_gap <= 1')
            raise MSATrimmerException
        self.__biomolecule = biomolecule.strip().upper() if biomolecule is not None else biomolecule
        self.__alignment_data = read_fasta_records(self.__msa_file)

    @property
    def alignment_data(self):
        return self.__alignment_data

    def _gap_mask(self):
        """bool[N, L]: True where a record has '.' or '-'."""
        chars = np.array([np.frombuffer(r.seq.encode('latin-1'), dtype=np.uint8) for r in self.__alignment_data])
        return (chars == ord('.')) | (chars == ord('-'))

    def compute_msa_columns_gap_size(self):
        """Fraction of gap characters in every column (msa_trimmer.py:60-95), as a tuple of floats."""
        gaps = self._gap_mask().sum(axis=0)
        num_seqs = float(len(self.__alignment_data))
        return tuple(float(g) / num_seqs for g in gaps)

    def msa_columns_beyond_max_gap(self):
        """Columns whose gap fraction exceeds max_gap, ascending (msa_trimmer.py:98-119)."""
        fractions = np.array(self.compute_msa_columns_gap_size())
        return tuple(int(c) for c in np.flatnonzero(fractions > self.__max_gap))

    def trim_by_gap_size(self):
        """Columns to drop when trimming by gap fraction alone (msa_trimmer.py:122-137)."""
        return self.msa_columns_beyond_max_gap()

    def trim_by_refseq(self, remove_all_gaps=False):
        """Columns to drop with respect to the row that matches the reference sequence (msa_trimmer.py:140-194): the
        columns where that row has a gap -- all of them with remove_all_gaps, otherwise only those that are also beyond
        the gap-fraction threshold."""
        backmapper = SequenceBackmapper(msa_file=self.__msa_file, refseq_file=self.__refseq_file, biomolecule=self.__biomolecule)
        row = backmapper.find_matching_seqs_from_alignment()[0]
        logger.info('\n\tRow of the alignment taken as the reference sequence:\n\t{}'.format(row))
        gapped = np.frombuffer(row.encode('latin-1'), dtype=np.uint8)
        gapped = (gapped == ord('-')) | (gapped == ord('.'))
        if not remove_all_gaps:
            keep_candidates = np.zeros(gapped.size, dtype=bool)
            keep_candidates[list(self.msa_columns_beyond_max_gap())] = True
            gapped &= keep_candidates
        columns = tuple(int(c) for c in np.flatnonzero(gapped))
        logger.info('\n\t{} columns are removed'.format(len(columns)))
        return columns

    def get_msa_trimmed_by_refseq(self, remove_all_gaps=False):
        """[(id, sequence without the trim_by_refseq columns)] for every record (msa_trimmer.py:197-207)."""
        length = len(self.__alignment_data[0].seq)
        keep = np.ones(length, dtype=bool)
        keep[list(self.trim_by_refseq(remove_all_gaps=remove_all_gaps))] = False
        kept = np.flatnonzero(keep)
        out = []
        for record in self.__alignment_data:
            chars = np.frombuffer(record.seq.encode('latin-1'), dtype=np.uint8)
            out.append((record.id, chars[kept].tobytes().decode('latin-1')))
        return out
