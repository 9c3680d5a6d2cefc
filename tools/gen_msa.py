#!/usr/bin/env python3
"""Deterministic synthetic alignments.

Per-site profiles ~ Dirichlet(0.3); N/5 founders drawn from the profiles; every
sequence copies a random founder and resamples each site from the profile with
probability 0.15 (so sequence reweighting is non-trivial); L/2 planted pairs (j-i > 4)
with a fixed random state permutation pi: with probability 0.6, x_nj = pi(x_ni).
Codes are 0-based with gap = q-1 (the C++ backend's coding); alphabet order
ACDEFGHIKLMNPQRSTVWY- (protein) / ACGU- (RNA).
"""
import argparse

import numpy as np

ALPHABET = {21: "ACDEFGHIKLMNPQRSTVWY-", 5: "ACGU-"}
SEEDS = {"B": 12345, "C": 12345, "D": 12346, "E": 12347}


def _sample_rows(rng, cdf, shape):
    """Draw states from per-site categorical cdfs (L x q) for an (n x L) block."""
    u = rng.random(shape)
    return (u[..., None] > cdf[None, :, :-1]).sum(axis=2).astype(np.uint8)


def generate(L, N, q, seed):
    rng = np.random.default_rng(seed)
    prof = rng.dirichlet(0.3 * np.ones(q), size=L)
    cdf = np.cumsum(prof, axis=1)
    nf = max(1, N // 5)
    founders = _sample_rows(rng, cdf, (nf, L))
    X = founders[rng.integers(nf, size=N)]
    block = max(1, (1 << 22) // max(1, L * q))
    for s in range(0, N, block):
        e = min(N, s + block)
        mask = rng.random((e - s, L)) < 0.15
        fresh = _sample_rows(rng, cdf, (e - s, L))
        X[s:e] = np.where(mask, fresh, X[s:e])
    # planted pairs
    npairs = L // 2
    for _ in range(npairs):
        i = int(rng.integers(0, max(1, L - 5)))
        j = int(rng.integers(i + 5, L)) if i + 5 < L else None
        if j is None:
            continue
        pi = rng.permutation(q).astype(np.uint8)
        hit = rng.random(N) < 0.6
        X[hit, j] = pi[X[hit, i]]
    return np.ascontiguousarray(X)


def dedup(X):
    """Keep the first occurrence of every distinct row, in order."""
    _, first = np.unique(X, axis=0, return_index=True)
    return np.ascontiguousarray(X[np.sort(first)])


def write_fasta(path, X, q):
    table = np.frombuffer(ALPHABET[q].encode(), dtype=np.uint8)
    with open(path, "wb") as fh:
        for n, row in enumerate(X):
            fh.write(b">s%d\n" % n)
            fh.write(table[row].tobytes())
            fh.write(b"\n")


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--L", type=int, required=True)
    ap.add_argument("--N", type=int, required=True)
    ap.add_argument("--q", type=int, default=21)
    ap.add_argument("--seed", type=int, default=12345)
    ap.add_argument("--out", required=True)
    a = ap.parse_args()
    write_fasta(a.out, generate(a.L, a.N, a.q, a.seed), a.q)
