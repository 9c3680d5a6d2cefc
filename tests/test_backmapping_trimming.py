"""Host-side rows of the scope table: reference-sequence back-mapping and MSA trimming
(8 f4).  The alignments run in libdca_hip.so's host code (no GPU involved), checked against the
pure-Python recurrence in oracle/sw.py; the mapping is checked on the reference's own test inputs
(tests/tests_input of the reference, copied as fixtures) with the reference's own assertion
(more than one site mapped, tests/sequence_backmapper_test.py:38-42) and with stronger ones."""
import os

import numpy as np
import pytest

from conftest import data_file


@pytest.fixture(scope="module")
def sm():
    from pydca_amd.sequence_backmapper import scoring_matrix
    return scoring_matrix


def test_scoring_matrices(sm):
    B = sm.BLOSUM62
    idx = [ord(c) - 65 for c in "ARNDCQEGHILKMFPSTWYV"]
    sub = B[np.ix_(idx, idx)]
    assert np.array_equal(sub, sub.T)
    assert [int(B[ord(c) - 65, ord(c) - 65]) for c in "ARNDCQEGHILKMFPSTWYV"] == [4, 5, 6, 6, 9, 5, 5, 6, 8, 4, 4, 5, 5, 6, 7, 4, 5, 11, 7, 4]
    assert int(sub.sum()) == int(np.trace(sub)) + 2 * int(np.triu(sub, 1).sum())
    # published row/column checks of BLOSUM62 (W row, C row)
    assert [int(B[ord("W") - 65, ord(c) - 65]) for c in "FYW"] == [1, 2, 11] and int(B[ord("C") - 65, ord("W") - 65]) == -2
    N = sm.NUC44
    assert N[0, 0] == 5 and N[ord("A") - 65, ord("U") - 65] == -4 and N[ord("G") - 65, ord("C") - 65] == -4
    assert sm.GAP_PENALTIES == {"PROTEIN": (-10, -1), "RNA": (-8, 0)}


@pytest.mark.parametrize("bio,alphabet", [("PROTEIN", "ARNDCQEGHILKMFPSTWYV"), ("RNA", "ACGU")])
def test_native_smith_waterman_vs_python_recurrence(sm, bio, alphabet):
    from oracle import sw
    from pydca_amd import _lib
    sub = sm.MATRICES[bio]
    go, ge = sm.GAP_PENALTIES[bio]
    score = lambda x, y: int(sub[ord(x) - 65, ord(y) - 65])
    rng = np.random.default_rng(11)
    ref = "".join(rng.choice(list(alphabet), 40))
    seqs = []
    for k in range(12):
        s = list(ref[rng.integers(0, 10):rng.integers(25, 40)])
        for _ in range(int(rng.integers(0, 6))):                 # mutate, insert, delete
            p = int(rng.integers(0, len(s)))
            op = rng.integers(0, 3)
            if op == 0:
                s[p] = str(rng.choice(list(alphabet)))
            elif op == 1:
                s[p:p] = list(rng.choice(list(alphabet), int(rng.integers(1, 4))))
            else:
                del s[p:p + int(rng.integers(1, 4))]
        seqs.append("".join(s))
    seqs.append("")            # empty sequence
    got = _lib.sw_scores(ref, seqs, sub, go, ge)
    want = [sw.local_score(ref, s, score, go, ge) for s in seqs]
    assert list(got) == want
    for s in seqs[:6]:
        a, b, sc, sa, sb = _lib.sw_align(ref, s, sub, go, ge)
        assert sc == sw.local_score(ref, s, score, go, ge)
        assert len(a) == len(b) and sw.alignment_score(a, b, score, go, ge) == sc
        assert a.replace("-", "") == ref[sa:sa + len(a.replace("-", ""))]
        assert b.replace("-", "") == s[sb:sb + len(b.replace("-", ""))]


CASES = [("MSA_RF00059_trimmed_gap_treshold_50.fa", "ref_seq_RF00059.faa", "rna"),
         ("PF02826.faa", "ref_seq_PF02826.faa", "protein"),
         ("MSA_RF00059_trimmed_gap_treshold_50.fa", "ref_seq_RF00059_test1.faa", "rna"),
         ("MSA_RF00059_trimmed_gap_treshold_50.fa", "ref_seq_RF00059_test2.faa", "rna"),
         ("MSA_RF00059_trimmed_gap_treshold_50.fa", "ref_seq_RF00059_test3.faa", "rna"),
         ("MSA_RF00059_trimmed_gap_treshold_50.fa", "ref_seq_RF00059_test4.faa", "rna"),
         ("MSA_RF00167.fa", "ref_RF00167.fa", "rna")]


@pytest.mark.parametrize("msa,ref,bio", CASES)
def test_map_to_reference_sequence_on_the_reference_test_inputs(msa, ref, bio):
    from pydca_amd.fasta_reader import fasta_reader
    from pydca_amd.sequence_backmapper.sequence_backmapper import SequenceBackmapper
    aln = fasta_reader.get_alignment_int_form(data_file(msa), biomolecule=bio)
    bm = SequenceBackmapper(alignment_data=aln, refseq_file=data_file(ref), biomolecule=bio)
    mapping = bm.map_to_reference_sequence()
    assert len(mapping) > 1                                     # the reference's own assertion
    template = bm.find_matching_seqs_from_alignment()[0]
    cols = sorted(mapping)
    assert [mapping[c] for c in cols] == sorted(mapping.values())            # order preserving
    assert len(set(mapping.values())) == len(mapping)
    assert all(template[c] != "-" for c in cols)
    # every one of these reference sequences occurs verbatim in its alignment: total, exact mapping
    assert len(mapping) == len(bm.ref_sequence)
    assert all(template[c] == bm.ref_sequence[r] for c, r in mapping.items())
    same = SequenceBackmapper(msa_file=data_file(msa), refseq_file=data_file(ref), biomolecule=bio).map_to_reference_sequence()
    assert same == mapping


def _golden_json(name):
    import json
    from conftest import GOLDEN
    with open(os.path.join(GOLDEN, name)) as fh:
        return json.load(fh)


def test_backmapping_logic_equals_the_reference_on_its_own_outputs():
    """Everything downstream of the pairwise alignment -- SequenceBackmapper.align_subsequences and
    map_to_reference_sequence (sequence_backmapper.py:286-466) -- against 300 cases that the REFERENCE's code produced
    (tests/golden/make_golden.py:backmap_golden; the alignment is injected on both sides, so Bio.pairwise2's choice
    among co-optimal alignments is the only part of row f3 that stays unpinned), including the inputs on which the
    reference raises IndexError."""
    import types
    from pydca_amd.sequence_backmapper.sequence_backmapper import SequenceBackmapper as SB
    cases = _golden_json("backmap_cases.json")
    assert len(cases) == 300 and sum(1 for c in cases if c["raises"]) >= 1
    for k, c in enumerate(cases):
        mid_ref = c["aligned_ref"][c["begin"]:c["end"]]
        nmid = len(c["aligned_template"][c["begin"]:c["end"]].replace("-", ""))
        if c["align_subsequences"].startswith("!"):
            with pytest.raises(IndexError):
                SB.align_subsequences(ref_middle_subseq=mid_ref, template_subseq_in_msa=c["row"], num_res_middle_template=nmid)
        else:
            assert SB.align_subsequences(ref_middle_subseq=mid_ref, template_subseq_in_msa=c["row"],
                                         num_res_middle_template=nmid) == c["align_subsequences"], k
        obj = object.__new__(SB)
        obj._SequenceBackmapper__alignment = [c["row"]]
        obj._SequenceBackmapper__ref_sequence = c["ref"]
        obj._SequenceBackmapper__biomolecule = "RNA"
        aln = [(c["aligned_ref"], c["aligned_template"], 0.0, c["begin"], c["end"])]
        obj.align_pairs_local = types.MethodType(lambda self, a, b, score_only=False, _aln=aln: _aln, obj)
        obj.find_matching_seqs_from_alignment = types.MethodType(lambda self, _row=c["row"]: [_row], obj)
        if c["raises"]:
            with pytest.raises(IndexError):
                obj.map_to_reference_sequence()
        else:
            got = obj.map_to_reference_sequence()
            assert sorted([int(a), int(b)] for a, b in got.items()) == c["mapping"], k


def test_trimmer_selections_equal_the_reference(monkeypatch):
    """MSATrimmer's column selections on MSA_RF00167.fa for five max_gap values, both refseq modes and the trimmed
    records against what the reference's MSATrimmer returned (tests/golden/trimmer_cases.json); the matching row is
    found here by the product's own Smith-Waterman search and must be the row the fixture was made with."""
    from pydca_amd.msa_trimmer.msa_trimmer import MSATrimmer
    from pydca_amd.sequence_backmapper.sequence_backmapper import SequenceBackmapper
    G = _golden_json("trimmer_cases.json")
    bm = SequenceBackmapper(msa_file=data_file("MSA_RF00167.fa"), refseq_file=data_file("ref_RF00167.fa"), biomolecule="rna")
    assert bm.find_matching_seqs_from_alignment()[0].replace(".", "-") == G["matching_row"].replace(".", "-").upper()
    for c in G["cases"]:
        tr = MSATrimmer(data_file("MSA_RF00167.fa"), biomolecule="rna", max_gap=c["max_gap"], refseq_file=data_file("ref_RF00167.fa"))
        assert list(tr.compute_msa_columns_gap_size()[:12]) == c["gap_size_first"]
        assert list(tr.msa_columns_beyond_max_gap()) == c["beyond"] and list(tr.trim_by_gap_size()) == c["by_gap"]
        assert list(tr.trim_by_refseq()) == c["by_refseq"]
        assert list(tr.trim_by_refseq(remove_all_gaps=True)) == c["by_refseq_all"]
        trimmed = tr.get_msa_trimmed_by_refseq()
        assert [list(trimmed[0]), list(trimmed[-1])] == c["trimmed_first"] and len(trimmed[0][1]) == c["trimmed_len"]


def test_backmapper_with_insertions_and_input_validation():
    from pydca_amd.sequence_backmapper.sequence_backmapper import SequenceBackmapper
    msa = [[1, 2, 3, 5, 4, 1, 2, 5, 3, 4, 1, 1], [2, 2, 3, 5, 4, 1, 3, 5, 3, 4, 2, 1]]      # RNA ints, 5 = gap
    bm = SequenceBackmapper(alignment_data=msa, ref_seq="cguacgua", biomolecule="rna")     # row 0 without A.., partial
    assert bm.alignment == ["ACG-UAC-GUAA", "CCG-UAG-GUCA"] and bm.ref_sequence == "CGUACGUA"
    mapping = bm.map_to_reference_sequence()
    assert mapping == {1: 0, 2: 1, 4: 2, 5: 3, 6: 4, 8: 5, 9: 6, 10: 7}
    out = bm.align_pairs_local("CGUACGUA", "ACGUACGUAA")
    assert out[0][2] == 40.0 and out[0][3:] == (1, 9) and out[0][0] == "-CGUACGUA-" and out[0][1] == "ACGUACGUAA"
    assert bm.align_pairs_local("CGUACGUA", "ACGUACGUAA", score_only=True) == 40.0
    assert SequenceBackmapper.align_subsequences("AAAA", "B--BBB", 4) == "A--AAA"
    with pytest.raises(ValueError):
        SequenceBackmapper(alignment_data=msa, ref_seq="ACGN", biomolecule="rna")
    with pytest.raises(ValueError):
        SequenceBackmapper(alignment_data=msa, biomolecule="rna")
    with pytest.raises(ValueError):
        SequenceBackmapper(ref_seq="ACGU", biomolecule="rna")


def test_trim_by_refseq_reproduces_the_notebook_trimming(tmp_path):
    """pydca trim_by_refseq rna MSA_RF00167.fa ref_RF00167.fa --remove_all_gaps (examples/pydca_demo.ipynb)
    -> the 71-column alignment that tests/golden/make_golden.py built independently from the
    REFERENCE record's gap pattern."""
    from pydca_amd import main
    from pydca_amd.msa_trimmer.msa_trimmer import MSATrimmer, MSATrimmerException, read_fasta_records
    out = main.run_pydca(["trim_by_refseq", "rna", data_file("MSA_RF00167.fa"), data_file("ref_RF00167.fa"), "--remove_all_gaps",
                          "--output_dir", str(tmp_path / "t")])
    assert os.path.basename(out) == "Trimmed_MSA_RF00167.fa"
    got = read_fasta_records(out)
    want = read_fasta_records(data_file("MSA_RF00167_trimmed71.fa"))
    assert len(got) == len(want) == 2704 and all(g.seq == w.seq for g, w in zip(got, want))
    assert got[5].id == want[5].id.split()[0]
    # without --remove_all_gaps only the gappy columns (> max_gap) of the matching row go
    tr = MSATrimmer(data_file("MSA_RF00167.fa"), biomolecule="rna", refseq_file=data_file("ref_RF00167.fa"), max_gap=0.05)
    cols = tr.trim_by_refseq()
    assert set(cols) <= set(tr.msa_columns_beyond_max_gap()) and 0 < len(cols) <= 31
    assert len(tr.get_msa_trimmed_by_refseq()[0][1]) == 102 - len(cols)
    with pytest.raises(MSATrimmerException):
        MSATrimmer(data_file("MSA_RF00167.fa"), max_gap=1.5)


def test_trim_by_gap_size(tmp_path):
    from pydca_amd import main
    from pydca_amd.msa_trimmer.msa_trimmer import MSATrimmer, read_fasta_records
    f = tmp_path / "m.fa"
    f.write_text(">a x\nAC-G.\n>b\nA--G-\n>c\nAC-GU\n>d\n-C-GU\n")
    tr = MSATrimmer(str(f), max_gap=0.4)
    assert tr.compute_msa_columns_gap_size() == (0.25, 0.25, 1.0, 0.0, 0.5)
    assert tr.trim_by_gap_size() == (2, 4)
    assert MSATrimmer(str(f)).trim_by_gap_size() == (2,)          # default max_gap 0.5, strict inequality
    out = main.run_pydca(["trim_by_gap_size", str(f), "--max_gap", "0.4", "--output_dir", str(tmp_path / "o")])
    assert open(out).read() == ">a\nACG\n>b\nA-G\n>c\nACG\n>d\n-CG\n"
    recs = read_fasta_records(data_file("MSA_RF00167.fa"))
    gaps = MSATrimmer(data_file("MSA_RF00167.fa")).compute_msa_columns_gap_size()
    col = 7
    assert abs(gaps[col] - sum(r.seq[col] in ".-" for r in recs) / len(recs)) < 1e-15
