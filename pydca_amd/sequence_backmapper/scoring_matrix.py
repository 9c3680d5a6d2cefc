"""Substitution matrices of the reference's local alignments as 26 x 26 integer tables indexed by
(letter - 'A') for the native aligner (dca_sw_scores / dca_sw_align in libdca_hip.so).

NUC44: the nucleotide part of pydca/sequence_backmapper/scoring_matrix.py:7-12 (match 5,
mismatch -4; the reference sequence and the gap-stripped MSA rows hold standard residues only, so
the ambiguity codes of the table are never looked up).
BLOSUM62: the standard matrix (Henikoff & Henikoff 1992) that the reference takes from
Bio.SubsMat.MatrixInfo (sequence_backmapper.py:4, biopython 1.74 -- a dependency outside the
reference tree).  Gap penalties: protein -10 / -1, RNA -8 / 0 (sequence_backmapper.py:206-217).
"""
import numpy as np

_BLOSUM62_ORDER = "ARNDCQEGHILKMFPSTWYV"
_BLOSUM62_ROWS = """
 4 -1 -2 -2  0 -1 -1  0 -2 -1 -1 -1 -1 -2 -1  1  0 -3 -2  0
-1  5  0 -2 -3  1  0 -2  0 -3 -2  2 -1 -3 -2 -1 -1 -3 -2 -3
-2  0  6  1 -3  0  0  0  1 -3 -3  0 -2 -3 -2  1  0 -4 -2 -3
-2 -2  1  6 -3  0  2 -1 -1 -3 -4 -1 -3 -3 -1  0 -1 -4 -3 -3
 0 -3 -3 -3  9 -3 -4 -3 -3 -1 -1 -3 -1 -2 -3 -1 -1 -2 -2 -1
-1  1  0  0 -3  5  2 -2  0 -3 -2  1  0 -3 -1  0 -1 -2 -1 -2
-1  0  0  2 -4  2  5 -2  0 -3 -3  1 -2 -3 -1  0 -1 -3 -2 -2
 0 -2  0 -1 -3 -2 -2  6 -2 -4 -4 -2 -3 -3 -2  0 -2 -2 -3 -3
-2  0  1 -1 -3  0  0 -2  8 -3 -3 -1 -2 -1 -2 -1 -2 -2  2 -3
-1 -3 -3 -3 -1 -3 -3 -4 -3  4  2 -3  1  0 -3 -2 -1 -3 -1  3
-1 -2 -3 -4 -1 -2 -3 -4 -3  2  4 -2  2  0 -3 -2 -1 -2 -1  1
-1  2  0 -1 -3  1  1 -2 -1 -3 -2  5 -1 -3 -1  0 -1 -3 -2 -2
-1 -1 -2 -3 -1  0 -2 -3 -2  1  2 -1  5  0 -2 -1 -1 -1 -1  1
-2 -3 -3 -3 -2 -3 -3 -3 -1  0  0 -3  0  6 -4 -2 -2  1  3 -1
-1 -2 -2 -1 -3 -1 -1 -2 -2 -3 -3 -1 -2 -4  7 -1 -1 -4 -3 -2
 1 -1  1  0 -1  0  0  0 -1 -2 -2  0 -1 -2 -1  4  1 -3 -2 -2
 0 -1  0 -1 -1 -1 -1 -2 -2 -1 -1 -1 -1 -2 -1  1  5 -2 -2  0
-3 -3 -4 -4 -2 -2 -3 -2 -2 -3 -2 -3 -1  1 -4 -3 -2 11  2 -3
-2 -2 -2 -3 -2 -1 -2 -3  2 -1 -1 -2 -1  3 -3 -2 -2  2  7 -1
 0 -3 -3 -3 -1 -2 -2 -3 -3  3  1 -2  1 -1 -2 -2  0 -3 -1  4
"""


def _table(order, rows, default):
    t = np.full((26, 26), default, dtype=np.int32)
    for a, row in zip(order, rows):
        for b, v in zip(order, row):
            t[ord(a) - 65, ord(b) - 65] = v
    return t


BLOSUM62 = _table(_BLOSUM62_ORDER, [[int(v) for v in ln.split()] for ln in _BLOSUM62_ROWS.strip().splitlines()], -4)
NUC44 = _table("ACGU", [[5 if a == b else -4 for b in "ACGU"] for a in "ACGU"], -4)

GAP_PENALTIES = {"PROTEIN": (-10, -1), "RNA": (-8, 0)}
MATRICES = {"PROTEIN": BLOSUM62, "RNA": NUC44}
