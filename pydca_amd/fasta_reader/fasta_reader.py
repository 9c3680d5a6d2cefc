"""FASTA input for the mfDCA path with the semantics of the reference's reader
(pydca/fasta_reader/fasta_reader.py): multi-line records, upper-casing, 1-based integer
states with gap = q, characters outside the table mapped to the gap state (:138-149),
exact duplicates dropped keeping the first occurrence (:153).  Host-side string work only;
no Biopython dependency."""
import logging

import numpy as np

logger = logging.getLogger(__name__)

RES_TO_INT_ALL = {
    'PROTEIN': {
        'A': 1, 'C': 2, 'D': 3, 'E': 4, 'F': 5, 'G': 6, 'H': 7, 'I': 8, 'K': 9, 'L': 10,
        'M': 11, 'N': 12, 'P': 13, 'Q': 14, 'R': 15, 'S': 16, 'T': 17, 'V': 18, 'W': 19, 'Y': 20,
        '-': 21, '.': 21, '~': 21,
    },
    'RNA': {'A': 1, 'C': 2, 'G': 3, 'U': 4, '-': 5, '.': 5, '~': 5},
}


class FastaReaderError(Exception):
    """Raised for problems while reading alignment data."""


def res_to_char(biomolecule):
    """fasta_reader.py:53-76: int -> letter ('.' and '~' excluded, so 21 / 5 maps to '-')."""
    RES_TO_INT = RES_TO_INT_ALL[biomolecule.strip().upper()]
    return {val: key for key, val in RES_TO_INT.items() if key not in ('.', '~')}


def get_alignment_from_fasta_file(file_name):
    """-> list of upper-cased sequence strings (fasta_reader.py:81-119)."""
    alignment, name, cur = [], None, []
    try:
        with open(file_name) as fh:
            for line in fh:
                line = line.strip()
                if not line:
                    continue
                if line.startswith('>'):
                    if name is not None:
                        alignment.append(''.join(cur))
                    name, cur = line[1:], []
                elif name is not None:
                    cur.append(line)
        if name is not None:
            alignment.append(''.join(cur))
    except Exception as expt:
        logger.error('\n\tError occured while reading from fasta file: {}.\n\tError type:{}\n\tArguments:{!r}'.format(
            file_name, type(expt).__name__, expt.args))
        raise
    alignment = [s.strip().upper() for s in alignment if s.strip()]
    if not alignment:
        logger.error('\n\tNo sequences found in {}'.format(file_name))
        raise ValueError
    lengths = {len(s) for s in alignment}
    if len(lengths) != 1:
        raise ValueError('Sequences in {} do not all have the same length'.format(file_name))
    return alignment


def alignment_letter2int(alignment, biomolecule='protein'):
    """-> list of lists of 1-based integer states, duplicates removed (fasta_reader.py:122-163)."""
    biomolecule = biomolecule.strip().upper()
    if biomolecule not in ('PROTEIN', 'RNA'):
        logger.error('\n\tBiomolecule {!r} is neither PROTEIN nor RNA'.format(biomolecule))
        raise ValueError
    q = 21 if biomolecule == 'PROTEIN' else 5
    table = np.full(256, q, dtype=np.int32)
    for ch, v in RES_TO_INT_ALL[biomolecule].items():
        table[ord(ch)] = v
    rows, seen = [], set()
    for seq in alignment:
        r = table[np.frombuffer(str(seq).upper().encode('latin-1'), dtype=np.uint8)]
        key = r.tobytes()
        if key not in seen:
            seen.add(key)
            rows.append(r.tolist())
    logger.info('\n\tRecords read from the file: {}'.format(len(alignment)))
    if not rows:
        logger.error('\n\tThe alignment is empty after encoding')
        raise ValueError
    return rows


def get_alignment_int_array(file_name, biomolecule='protein', zero_based=False):
    """The de-duplicated alignment of fasta_reader.py:166-188 as an int array [N', L] of the reference's 1-based states
    (uint8; gap = q).  Read by the native reader in libdca_hip.so (one mmap, threaded encoding, hashed first-occurrence
    de-duplication: milliseconds for 10^5 sequences where the per-record Python path needs a second); files the
    native reader declines (non-ASCII bytes) go through the text-mode path below -- same result by construction,
    tests/test_host_logic.py compares the two on every fixture."""
    from .. import _lib
    biomolecule = biomolecule.strip().upper()
    if biomolecule not in ('PROTEIN', 'RNA'):
        logger.error('\n\tBiomolecule {!r} is neither PROTEIN nor RNA'.format(biomolecule))
        raise ValueError
    try:
        X0, raw = _lib.read_fasta(file_name, _lib.PROTEIN if biomolecule == 'PROTEIN' else _lib.RNA)
    except _lib.DcaBackendError as exc:
        if exc.code == _lib.DCA_ERR_IO:
            logger.error('\n\tError occured while reading from fasta file: {}'.format(file_name))
            raise FileNotFoundError(file_name)
        if exc.code != _lib.DCA_ERR_RESIDUE:
            raise
        X1 = np.array(alignment_letter2int(get_alignment_from_fasta_file(file_name), biomolecule), dtype=np.uint8)
        return X1 - np.uint8(1) if zero_based else X1
    logger.info('\n\tRecords read from the file: {}'.format(raw))
    return X0 if zero_based else X0 + np.uint8(1)       # zero_based: the device's coding (gap = q - 1), no extra pass


def get_alignment_int_form(file_name, biomolecule='protein'):
    """fasta_reader.py:166-188 -> list of lists of 1-based states."""
    return get_alignment_int_array(file_name, biomolecule).tolist()


def sequences_to_char_form(seqs_lst, biomolecule):
    """fasta_reader.py:227-249: integer sequences -> strings (gap state -> '-')."""
    RES_TO_CHAR = res_to_char(biomolecule)
    return [''.join(RES_TO_CHAR[res] for res in seq_int) for seq_int in seqs_lst]


def get_alignment_char_form(file_name, biomolecule='PROTEIN'):
    """fasta_reader.py:191-224: the de-duplicated alignment with non-standard residues turned into
    '-' (int form and back)."""
    biomolecule = biomolecule.strip().upper()
    return sequences_to_char_form(get_alignment_int_form(file_name, biomolecule=biomolecule), biomolecule)
