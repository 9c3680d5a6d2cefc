"""CPU-only tests: host logic of the product (readers, sharding, writers, CLI surface), the
C-ABI library's exports, and the rule that the product never touches oracle/."""
import ctypes as C
import os
import re
import subprocess
import sys

import numpy as np
import pytest

from conftest import ROOT, data_file, golden

sys.path.insert(0, ROOT)


def test_library_exports_every_declared_symbol():
    """Every function declared in include/dca_hip.h is exported by libdca_hip.so (no compute calls)."""
    from pydca_amd import _lib
    header = open(os.path.join(ROOT, "include", "dca_hip.h")).read()
    header = re.sub(r"/\*.*?\*/", "", header, flags=re.S)
    header = re.sub(r"typedef[^;{]*(\{[^}]*\})?[^;]*;", "", header, flags=re.S)
    header = re.sub(r"^\s*#.*$", "", header, flags=re.M)          # preprocessor lines
    declared = set(re.findall(r"\b([A-Za-z_][A-Za-z0-9_]*)\s*\([^;{]*\)\s*;", header))
    declared -= {"dca_reduce_hook"}
    assert {"plmdcaBackend", "freeFieldsAndCouplings", "dca_plm_gradient", "dca_mf_run"} <= declared
    lib = C.CDLL(_lib.LIB_PATH)
    missing = [name for name in sorted(declared) if not hasattr(lib, name)]
    assert not missing, missing
    assert set(_lib.EXPORTS) <= declared | {"dca_mf_corr_from_freqs"}


def test_no_gpu_fails_loudly_instead_of_falling_back():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is visible")
    from pydca_amd import _lib
    with pytest.raises(_lib.DcaBackendError) as ei:
        _lib.Context(0, _lib.DCA_F32)
    assert "no CPU fallback" in str(ei.value) or ei.value.code == -6
    # the drop-in symbol reports NULL + message, it does not crash or compute on the host
    lib = _lib.lib()
    assert not lib.plmdcaBackend(2, 5, os.fsencode(data_file("toy_rna.fa")), 10, 0.8, 1.0, 1.0, 3, 1, False)


def test_product_never_imports_the_oracle():
    bad = []
    for dirpath, _dirs, files in os.walk(os.path.join(ROOT, "pydca_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".cpp", ".h")):
                txt = open(os.path.join(dirpath, f), errors="replace").read()
                if re.search(r"^\s*(from|import)\s+oracle\b", txt, flags=re.M) or "oracle/" in txt and f.endswith((".hip", ".cpp", ".h")):
                    bad.append(os.path.join(dirpath, f))
    assert not bad, bad


@pytest.mark.parametrize("tag,fname,bio", [("toy_rna", "toy_rna.fa", 2), ("toy_protein", "toy_protein.fa", 1),
                                            ("rf71", "MSA_RF00167_trimmed71.fa", 2), ("rf00167", "MSA_RF00167.fa", 2)])
def test_cxx_reader_semantics_bit_exact(tag, fname, bio):
    """dca_read_msa == PlmDCA::readSequencesFromFile (plmdca_numerics.cpp:685-767) on the
    rows the real reference kept (golden)."""
    from pydca_amd import _lib
    G = golden("plm_" + tag)
    X, raw = _lib.read_msa(data_file(fname), bio, int(G["L"]))
    assert raw == int(G["raw_count"])
    assert np.array_equal(X, G["X"])


def test_cxx_reader_errors(tmp_path):
    from pydca_amd import _lib
    p = tmp_path / "dna.fa"
    p.write_text(">a\nACGT\n")            # 'T' is the gap state in the reference's RNA table (:729)
    X, raw = _lib.read_msa(str(p), 2, 4)
    assert raw == 1 and X.tolist() == [[0, 1, 2, 4]]
    p.write_text(">a\nAC*U\n")            # a character the table lacks: the reference throws (:752)
    with pytest.raises(_lib.DcaBackendError) as ei:
        _lib.read_msa(str(p), 2, 4)
    assert ei.value.code == -3
    with pytest.raises(_lib.DcaBackendError) as ei:
        _lib.read_msa(str(tmp_path / "missing.fa"), 2, 4)
    assert ei.value.code == -2
    p2 = tmp_path / "ok.fa"
    p2.write_text(">a\nacgu-\n\n>b\nACGU-\n>c\nNNNN.\n")   # lower case, duplicate, empty line, unknown letters -> gap
    X, raw = _lib.read_msa(str(p2), 2, 5)
    assert raw == 3 and X.tolist() == [[0, 1, 2, 3, 4], [4, 4, 4, 4, 4]]


def _reader_sweep_cases():
    return [str(c) for c in golden("reader_sweep")["cases"]]


@pytest.mark.parametrize("case", _reader_sweep_cases())
def test_cxx_reader_sweep(case, tmp_path):
    """dca_read_msa == PlmDCA::readSequencesFromFile on the alphabet sweep (all 26 letters in both
    cases, the gap characters, duplicates, CRLF, long lines): rows array_equal to what the compiled
    reference returned, an error code where it throws."""
    from pydca_amd import _lib
    G = golden("reader_sweep")
    p = tmp_path / (case + ".fa")
    p.write_bytes(G[case + "_text"].tobytes())
    bio, L = int(G[case + "_bio"]), int(G[case + "_L"])
    if bool(G[case + "_throws"]):
        with pytest.raises(_lib.DcaBackendError) as ei:
            _lib.read_msa(str(p), bio, L)
        assert ei.value.code == -3
    else:
        X, raw = _lib.read_msa(str(p), bio, L)
        assert np.array_equal(X, G[case + "_rows"])


@pytest.mark.parametrize("tag,fname,bio", [("toy_rna", "toy_rna.fa", "rna"), ("toy_protein", "toy_protein.fa", "protein"),
                                            ("rf00167", "MSA_RF00167.fa", "RNA")])
def test_python_reader_matches_reference_reader(tag, fname, bio):
    """fasta_reader.get_alignment_int_form == the reference's Biopython-based reader (golden X)."""
    from pydca_amd.fasta_reader import fasta_reader
    G = golden("mf_" + tag)
    X = np.array(fasta_reader.get_alignment_int_form(data_file(fname), bio), dtype=np.int32)
    assert np.array_equal(X, G["X"])


def _text_mode_reader(path, bio):
    from pydca_amd.fasta_reader import fasta_reader
    return np.array(fasta_reader.alignment_letter2int(fasta_reader.get_alignment_from_fasta_file(path), bio), dtype=np.uint8)


@pytest.mark.parametrize("fname,bio", [("toy_rna.fa", "rna"), ("toy_protein.fa", "protein"), ("MSA_RF00167.fa", "rna"),
                                       ("MSA_RF00167_trimmed71.fa", "rna"), ("PF02826.faa", "protein"),
                                       ("MSA_RF00059_trimmed_gap_treshold_50.fa", "rna")])
def test_native_fasta_reader_equals_text_mode_reader(fname, bio):
    """dca_read_fasta (mmap, threaded encoding, hashed first-occurrence de-duplication) == the per-record Python
    reader that test_python_reader_matches_reference_reader pins to the reference's Biopython-based reader."""
    from pydca_amd.fasta_reader import fasta_reader
    A = fasta_reader.get_alignment_int_array(data_file(fname), bio)
    assert A.dtype == np.uint8 and np.array_equal(A, _text_mode_reader(data_file(fname), bio))
    assert fasta_reader.get_alignment_int_form(data_file(fname), bio) == A.tolist()


def test_native_fasta_reader_edge_cases(tmp_path):
    """Multi-line records, CRLF and lone-CR line ends, surrounding blanks, lower case, characters outside the alphabet
    (gap state, fasta_reader.py:138-149), text before the first header, empty records, duplicates across different
    line wrapping, no trailing newline; unequal lengths and empty files raise ValueError, a missing file
    FileNotFoundError; non-ASCII bytes take the text-mode path."""
    from pydca_amd import _lib
    from pydca_amd.fasta_reader import fasta_reader
    p = tmp_path / "edge.fa"
    p.write_bytes(b"junk before any header\n>a desc\r\nACGU\r\n-acg\r\n>b\n  ACGU-ACG\t\n>empty\n\n>c\rNNNN\r..~*\r>d\nACGU\n-ACG\n\n>e\nUUUUUUUU")
    A = fasta_reader.get_alignment_int_array(str(p), "rna")
    assert np.array_equal(A, _text_mode_reader(str(p), "rna"))
    assert A.tolist() == [[1, 2, 3, 4, 5, 1, 2, 3], [5] * 8, [4] * 8]            # a == b == d; c is all gaps
    X0, raw = _lib.read_fasta(str(p), _lib.RNA)
    assert raw == 5 and np.array_equal(X0 + 1, A)
    q = tmp_path / "prot.fa"
    q.write_text(">x\nacdefghiklmnpqrstvwy-.~bjouxz*1\n")
    assert fasta_reader.get_alignment_int_array(str(q), "protein").tolist() == [list(range(1, 21)) + [21] * 11]
    bad = tmp_path / "ragged.fa"
    bad.write_text(">a\nACGU\n>b\nACG\n")
    with pytest.raises(ValueError):
        fasta_reader.get_alignment_int_array(str(bad), "rna")
    empty = tmp_path / "empty.fa"
    empty.write_text("\n\n")
    with pytest.raises(ValueError):
        fasta_reader.get_alignment_int_array(str(empty), "rna")
    with pytest.raises(FileNotFoundError):
        fasta_reader.get_alignment_int_array(str(tmp_path / "missing.fa"), "rna")
    with pytest.raises(ValueError):
        fasta_reader.get_alignment_int_array(str(p), "lipid")
    u = tmp_path / "utf8.fa"
    u.write_bytes(b">a\nACXU\n>b\nACG\xc3\xa9\n")              # 'e' with an acute accent, UTF-8: one character, two bytes
    with pytest.raises(_lib.DcaBackendError) as ei:
        _lib.read_fasta(str(u), _lib.RNA)
    assert ei.value.code == _lib.DCA_ERR_RESIDUE
    assert fasta_reader.get_alignment_int_array(str(u), "rna").tolist() == _text_mode_reader(str(u), "rna").tolist()


def test_readers_take_a_fifo(tmp_path):
    """Inputs that are not regular files (a FIFO here; /dev/stdin and process substitutions alike) cannot be mapped: the
    native readers read them into memory and run the same indexer -- the reference's std::ifstream and Biopython readers
    accept such inputs too."""
    import ctypes as C
    import threading
    from pydca_amd import _lib
    G = golden("plm_toy_rna")
    src = open(data_file("toy_rna.fa"), "rb").read()

    def through_fifo(call):
        fifo = str(tmp_path / "msa.fifo")
        if os.path.exists(fifo):
            os.unlink(fifo)
        os.mkfifo(fifo)
        t = threading.Thread(target=lambda: open(fifo, "wb").write(src))
        t.start()
        try:
            return call(fifo)
        finally:
            t.join()
    X, raw = through_fifo(lambda f: _lib.read_fasta(f, _lib.RNA))
    X_file, raw_file = _lib.read_fasta(data_file("toy_rna.fa"), _lib.RNA)
    assert raw == raw_file and np.array_equal(X, X_file)

    def cxx(fifo):
        out = np.empty((int(G["raw_count"]) + 1, int(G["L"])), dtype=np.uint8)
        rawc = C.c_int(0)
        n = _lib.lib().dca_read_msa(os.fsencode(fifo), 2, int(G["L"]), out.ctypes.data_as(C.c_void_p), out.shape[0], C.byref(rawc))
        return out[:n], rawc.value
    X2, raw2 = through_fifo(cxx)
    assert raw2 == int(G["raw_count"]) and np.array_equal(X2, G["X"])
    with pytest.raises(Exception):
        _lib.read_fasta(str(tmp_path), _lib.RNA)          # a directory is still an error


def test_readers_large_alignment_with_scattered_duplicates(tmp_path):
    """Both native readers on 20 000 x 300 with 15 % duplicates scattered through the file: first occurrences kept in
    file order (numpy reference), row hashes + memcmp, several host threads."""
    from pydca_amd import _lib
    from tools.gen_msa import ALPHABET, write_fasta
    rng = np.random.default_rng(5)
    N, L, q = 20000, 300, 21
    base = rng.integers(0, q, size=(N, L), dtype=np.uint8)
    dup = rng.random(N) < 0.15
    dup[0] = False
    src = rng.integers(0, np.maximum(np.arange(N), 1))          # an earlier row
    X = base.copy()
    for n in np.nonzero(dup)[0]:
        X[n] = X[src[n]]
    f = tmp_path / "big.fa"
    write_fasta(str(f), X, q)
    _, first = np.unique(X, axis=0, return_index=True)
    expect = X[np.sort(first)]
    got, raw = _lib.read_msa(str(f), _lib.PROTEIN, L)
    assert raw == N and np.array_equal(got, expect)
    got2, raw2 = _lib.read_fasta(str(f), _lib.PROTEIN)
    assert raw2 == N and np.array_equal(got2, expect)
    # first L' < L characters only (plmdca_numerics.cpp:750-753): more duplicates appear
    Lp = 3
    _, firstp = np.unique(X[:, :Lp], axis=0, return_index=True)
    gotp, _ = _lib.read_msa(str(f), _lib.PROTEIN, Lp)
    assert np.array_equal(gotp, X[np.sort(firstp), :Lp])


def test_shard_bounds_partition_and_halo():
    from pydca_amd import parallel
    for n in (1, 7, 64, 1000, 50000):
        for world in (1, 2, 3, 8):
            if world > n:
                continue
            cover = []
            for r in range(world):
                a, b = parallel.shard_bounds(n, world, r)
                assert 0 <= a <= b <= n
                cover += list(range(a, b))
                first, stop, halo = parallel.shard_with_halo(n, world, r, 40)
                assert stop == b and first == a - halo and 0 <= halo <= 40 and first >= 0
                if r == 0:
                    assert halo == 0
            assert cover == list(range(n))
            sizes = [parallel.shard_bounds(n, world, r)[1] - parallel.shard_bounds(n, world, r)[0] for r in range(world)]
            assert max(sizes) - min(sizes) <= 1


def test_initial_x_host_matches_oracle(oracle_plm):
    from pydca_amd import parallel
    G = golden("plm_toy_protein")
    x = parallel.initial_x(G["X"], G["w"], int(G["q"]), np.float32)
    ref = oracle_plm.init_x(G["X"], G["w"], int(G["q"]))
    np.testing.assert_allclose(x, ref, rtol=2e-6, atol=2e-6)
    np.testing.assert_allclose(x, G["x0"], rtol=2e-6, atol=2e-6)


def test_score_file_format(tmp_path):
    """dca_utilities.write_sorted_dca_scores: header rule, metadata, 1-based sites, column widths
    (pydca/dca_utilities/dca_utilities.py:236-266)."""
    from pydca_amd.dca_utilities import dca_utilities

    class Inst:
        biomolecule, num_sequences, sequences_len = "RNA", 10, 7
        sequence_identity, lambda_h, lambda_J, max_iterations = 0.8, 1.0, 20.0, 100

    path = dca_utilities.get_dca_output_file_path(str(tmp_path), "/x/y/MSA_abc.fa", prefix="PLMDCA_apc_fn_scores_", postfix=".txt")
    assert os.path.basename(path) == "PLMDCA_apc_fn_scores_MSA_abc.txt"
    dca_utilities.write_sorted_dca_scores(path, [((0, 5), 1.25), ((2, 3), 0.5)], metadata=dca_utilities.plmdca_param_metadata(Inst()),
                                          score_type="PLMDCA Frobenius norm, average product corrected (APC)")
    lines = open(path).read().splitlines()
    assert lines[0] == "#" + "=" * 70
    assert lines[1] == "# PARAMETERS USED FOR THIS COMPUTATION: "
    assert "#\tlambda_J: 20.0" in lines
    assert lines[-2] == "{0:<7} {1:<14} {2:<35}".format(1, 6, 1.25)
    assert lines[-1] == "{0:<7} {1:<14} {2:<35}".format(3, 4, 0.5)


def test_cli_surface():
    """Sub-commands and flags of the reference CLIs (plmdca_main.py:262-328, mfdca_main.py:310-394)."""
    env = dict(os.environ, PYTHONPATH=ROOT)
    out = subprocess.run([sys.executable, "-m", "pydca_amd.plmdca_main", "compute_fn", "--help"], env=env, capture_output=True, text=True)
    assert out.returncode == 0
    for flag in ("--seqid", "--lambda_h", "--lambda_J", "--max_iterations", "--num_threads", "--refseq_file", "--verbose", "--apc", "--output_dir"):
        assert flag in out.stdout
    out = subprocess.run([sys.executable, "-m", "pydca_amd.mfdca_main", "compute_fn", "--help"], env=env, capture_output=True, text=True)
    assert out.returncode == 0
    for flag in ("--seqid", "--pseudocount", "--refseq_file", "--output_dir", "--verbose", "--apc"):
        assert flag in out.stdout
    out = subprocess.run([sys.executable, "-m", "pydca_amd.plmdca_main"], env=env, capture_output=True, text=True)
    assert "compute_fn" in out.stdout and "compute_di" in out.stdout and "compute_params" in out.stdout


def test_class_validation_matches_reference():
    """Argument validation happens before any GPU work (plmdca.py:53-71, meanfield_dca.py:76-95)."""
    from pydca_amd.meanfield_dca.meanfield_dca import MeanFieldDCA
    from pydca_amd.plmdca.plmdca import PlmDCA, PlmDCAException
    f = data_file("toy_rna.fa")
    with pytest.raises(PlmDCAException):
        PlmDCA(f, "dna")
    with pytest.raises(PlmDCAException):
        PlmDCA(f, "rna", seqid=1.5)
    with pytest.raises(PlmDCAException):
        PlmDCA(f, "rna", lambda_J=-1.0)
    p = PlmDCA(f, "rna")
    assert (p.sequences_len, p.num_sequences, p.max_iterations) == (10, 48, 100)
    assert abs(p.lambda_h - 0.2 * 9) < 1e-12 and p.sequence_identity == 0.8
    assert p.map_index_couplings(0, 1, 0, 0) == 10 * 5
    with pytest.raises(NotImplementedError):
        p.effective_num_sequences
    with pytest.raises(ValueError):
        MeanFieldDCA(f, "rna", pseudocount=1.0)
    with pytest.raises(ValueError):
        MeanFieldDCA(f, "rna", seqid=0.0)
    with pytest.raises(ValueError):
        MeanFieldDCA(f, "lipid")


def test_ranked_list_equals_pythons_stable_sort():
    """pydca_amd/_ranking.py builds the classes' return value without a Python-level loop; it must equal what the reference
    builds (meanfield_dca.py:940: sorted(dict.items(), key=score, reverse=True) -- stable, so ties stay in (i, j) order),
    element for element and type for type (pairs are tuples of ints, scores numpy scalars), also with ties, L = 2 and L = 1."""
    from pydca_amd import _ranking
    rng = np.random.default_rng(5)
    for L in (1, 2, 3, 17, 60):
        npairs = L * (L - 1) // 2
        scores = np.round(rng.standard_normal(npairs), 1)          # one decimal: plenty of ties
        ref = sorted((((i, j), scores[k]) for k, (i, j) in enumerate((i, j) for i in range(L) for j in range(i + 1, L))),
                     key=lambda kv: kv[1], reverse=True)
        for order in (None, np.argsort(-scores, kind='stable').astype(np.int32)):
            got = _ranking.ranked(scores, L, order)
            assert got == ref
            assert all(type(p) is tuple and type(p[0]) is int and isinstance(s, np.float64) for p, s in got)
    assert _ranking.pair_tuples(60) is _ranking.pair_tuples(60)    # cached per L


def test_devices_argument_is_validated_like_the_other_arguments():
    """devices= / --devices (the counterpart of the reference's num_threads): parsing and validation need no GPU."""
    from pydca_amd import multi_gpu
    from pydca_amd.plmdca import plmdca
    assert multi_gpu.parse_devices("0, 1,2") == [0, 1, 2] and multi_gpu.parse_devices([5]) == [5]
    assert multi_gpu.parse_devices(None) is None and multi_gpu.parse_devices("") is None
    for bad in ("a,b", "-1", "0,0", [1, 1]):
        os.environ.pop("DCA_RCCL_PATH", None)
        with pytest.raises(ValueError):
            multi_gpu.parse_devices(bad)
    with pytest.raises(plmdca.PlmDCAException):
        plmdca.PlmDCA(data_file("toy_rna.fa"), "rna", devices="0;1")
    p = plmdca.PlmDCA(data_file("toy_rna.fa"), "rna", devices=[3])             # one entry: the same as device=3; nothing runs yet
    assert p.sequences_len == 10


def test_ranked_list_c_module_equals_python_construction():
    """pydca_amd/csrc/fastrank.c builds the ranked list [((i, j), numpy.float64), ...] in one C loop; the pure-Python construction of
    pydca_amd/_ranking.py is the same list (ties in pair order, as sorted(..., reverse=True) on the pair-ordered list gives the
    reference, meanfield_dca.py:940)."""
    from pydca_amd import _ranking as R
    assert R._fast is not None, "build it: make -C pydca_amd/csrc"
    rng = np.random.default_rng(4)
    for L in (2, 3, 17, 120):
        n = L * (L - 1) // 2
        s = np.round(rng.random(n), 2)                   # many ties
        R._seen_once.discard(L)
        R._pair_tuples.pop(L, None)
        first = R.ranked(s, L)                           # the first list of a process for this L: tuples made on the fly
        fast = R.ranked(s, L)                            # the second: from the cache
        assert first == fast and L in R._pair_tuples
        keep, R._fast = R._fast, None
        try:
            R._pair_tuples.clear()
            slow = R.ranked(s, L)
        finally:
            R._fast = keep
            R._pair_tuples.clear()
        ref = sorted(zip(zip(*[a.tolist() for a in np.triu_indices(L, 1)]), s), key=lambda t: t[1], reverse=True)
        assert fast == slow == ref
        assert all(type(sc) is np.float64 and type(p) is tuple and type(p[0]) is int for p, sc in fast)
    assert R.ranked(np.array([]), 1) == []
