"""TEST INFRASTRUCTURE ONLY -- pure-Python Smith-Waterman with affine gaps (Gotoh), the textbook
recurrence that Bio.pairwise2.align.localds evaluates for SequenceBackmapper.align_pairs_local
(sequence_backmapper.py:186-230; gap of length n costs open + (n-1)*extend).  Small inputs only.
Used by tests/ to check dca_sw_scores / dca_sw_align of the host library."""


def local_score(a, b, score, gap_open, gap_extend):
    NEG = -10 ** 9
    la, lb = len(a), len(b)
    H = [[0] * (lb + 1) for _ in range(la + 1)]
    E = [[NEG] * (lb + 1) for _ in range(la + 1)]
    F = [[NEG] * (lb + 1) for _ in range(la + 1)]
    best = 0
    for i in range(1, la + 1):
        for j in range(1, lb + 1):
            E[i][j] = max(E[i - 1][j] + gap_extend, H[i - 1][j] + gap_open)
            F[i][j] = max(F[i][j - 1] + gap_extend, H[i][j - 1] + gap_open)
            H[i][j] = max(0, H[i - 1][j - 1] + score(a[i - 1], b[j - 1]), E[i][j], F[i][j])
            best = max(best, H[i][j])
    return best


def alignment_score(aligned_a, aligned_b, score, gap_open, gap_extend):
    """Score of a gapped alignment (both strings equal length, '-' = gap)."""
    total, in_gap_a, in_gap_b = 0, False, False
    for x, y in zip(aligned_a, aligned_b):
        if x == '-':
            total += gap_extend if in_gap_a else gap_open
            in_gap_a, in_gap_b = True, False
        elif y == '-':
            total += gap_extend if in_gap_b else gap_open
            in_gap_b, in_gap_a = True, False
        else:
            total += score(x, y)
            in_gap_a = in_gap_b = False
    return total
