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
            logger.error('\n\tThe value of max_gap should be between 0 and 1')
            raise MSATrimmerException
        self.__biomolecule = biomolecule.strip().upper() if biomolecule is not None else biomolecule
        self.__alignment_data = read_fasta_records(self.__msa_file)

    @property
    def alignment_data(self):
        return self.__alignment_data

    def compute_msa_columns_gap_size(self):
        """msa_trimmer.py:60-95: fraction of '.' / '-' per column."""
        chars = np.array([np.frombuffer(r.seq.encode('latin-1'), dtype=np.uint8) for r in self.__alignment_data])
        num_seqs = len(self.__alignment_data)
        gaps = ((chars == ord('.')) | (chars == ord('-'))).sum(axis=0)
        return tuple(float(g) / float(num_seqs) for g in gaps)

    def msa_columns_beyond_max_gap(self):
        """msa_trimmer.py:98-119."""
        columns_gap_size = self.compute_msa_columns_gap_size()
        return tuple(i for i in range(len(columns_gap_size)) if columns_gap_size[i] > self.__max_gap)

    def trim_by_gap_size(self):
        """msa_trimmer.py:122-137."""
        return tuple(self.msa_columns_beyond_max_gap())

    def trim_by_refseq(self, remove_all_gaps=False):
        """msa_trimmer.py:140-194."""
        seqbackmapper = SequenceBackmapper(msa_file=self.__msa_file, refseq_file=self.__refseq_file,
                                           biomolecule=self.__biomolecule)
        first_matching_seq = seqbackmapper.find_matching_seqs_from_alignment()[0]
        gap_symbols = ['-', '.']
        if not remove_all_gaps:
            candidates = self.msa_columns_beyond_max_gap()
            columns_to_remove = [i for i in candidates if first_matching_seq[i] in gap_symbols]
        else:
            seqs_len = len(self.__alignment_data[0].seq)
            columns_to_remove = [i for i in range(seqs_len) if first_matching_seq[i] in gap_symbols]
        return tuple(columns_to_remove)

    def get_msa_trimmed_by_refseq(self, remove_all_gaps=False):
        """msa_trimmer.py:197-207."""
        columns_to_remove = set(self.trim_by_refseq(remove_all_gaps=remove_all_gaps))
        trimmed_msa = list()
        for record in self.__alignment_data:
            trimmed_seq = [record.seq[i] for i in range(len(record.seq)) if i not in columns_to_remove]
            trimmed_msa.append((record.id, ''.join(trimmed_seq)))
        return trimmed_msa
