#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ from the REAL reference.

Run once in the build container (needs /root/reference; never on the GPU box):
    python tests/golden/make_golden.py [--skip-slow]

What it does
  * copies the reference's own test/example DATA files (alignments only) into
    tests/golden/data/ and writes the toy / trimmed alignments derived from them;
  * plmDCA: calls the reference's compiled C++ (oracle/_ref/libpydca_ref.so, built by
    `make -C oracle ref` from the sources under /root/reference) for the de-duplicated
    sequences, weights, initial x and (x, fx, g) triples of PlmDCA::gradient, plus a few
    full `plmdcaBackend` runs (1 thread => deterministic);
  * mfDCA: imports the reference's Python package from /root/reference.  numba and
    Biopython are not installed in the image, so two probe stubs are put on sys.path in
    a temp dir: `numba.jit` = identity decorator, `prange` = range, and a minimal
    `Bio.AlignIO.read` / `Bio.Align.MultipleSeqAlignment`.  The stubbed import is
    validated against the values published in examples/pydca_demo.ipynb cell 10
    (KAT_MF below) before anything is written.
No reference source text is stored: fixtures are inputs and numeric outputs only.
"""
import argparse
import os
import shutil
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
DATA = os.path.join(HERE, "data")
REF = "/root/reference"
sys.path.insert(0, ROOT)
sys.dont_write_bytecode = True

from oracle import plm as oplm  # noqa: E402

# examples/pydca_demo.ipynb, cell 10 (mfDCA) and cell 5 (plmDCA) outputs -- published values
KAT_MF = [((44, 56), 4.290005464965937), ((16, 28), 4.210860173400806),
          ((17, 27), 4.207758141824402), ((45, 55), 4.190480172499375),
          ((43, 57), 4.039065434604404)]
KAT_PLM = [((45, 55), 2.571105010148373), ((44, 56), 2.5499034354788233),
           ((16, 28), 2.530926710394089), ((17, 27), 2.502918758504699),
           ((43, 57), 2.4373260100372316)]


def read_records(path):
    recs, name, cur = [], None, []
    with open(path) as fh:
        for line in fh:
            line = line.rstrip("\n")
            if line.startswith(">"):
                if name is not None:
                    recs.append((name, "".join(cur)))
                name, cur = line[1:], []
            elif line.strip():
                cur.append(line.strip())
    if name is not None:
        recs.append((name, "".join(cur)))
    return recs


def write_fasta(path, recs):
    with open(path, "w") as fh:
        for name, seq in recs:
            fh.write(">%s\n%s\n" % (name, seq))


def make_toy(path, alphabet, n, L, seed):
    """Small random alignment with clusters (so reweighting matters), a few exact
    duplicates (so dedup matters) and lower-case letters (so toupper matters)."""
    rng = np.random.default_rng(seed)
    q = len(alphabet)
    prof = rng.dirichlet(0.5 * np.ones(q), size=L)
    founders = np.stack([[rng.choice(q, p=prof[i]) for i in range(L)] for _ in range(max(2, n // 6))])
    rows = []
    for k in range(n):
        r = founders[rng.integers(len(founders))].copy()
        m = rng.random(L) < 0.25
        r[m] = [rng.choice(q, p=prof[i]) for i in np.nonzero(m)[0]]
        rows.append(r)
    for k in range(3):
        rows[n - 1 - k] = rows[k].copy()
    recs = []
    for k, r in enumerate(rows):
        s = "".join(alphabet[v] for v in r)
        if k % 7 == 3:
            s = s.lower()
        recs.append(("t%03d" % k, s))
    write_fasta(path, recs)


def install_stubs(tmp):
    with open(os.path.join(tmp, "numba.py"), "w") as fh:
        fh.write("def jit(*a, **k):\n"
                 "    if len(a) == 1 and callable(a[0]) and not k:\n        return a[0]\n"
                 "    return lambda f: f\nprange = range\n")
    os.makedirs(os.path.join(tmp, "Bio"))
    open(os.path.join(tmp, "Bio", "__init__.py"), "w").close()
    with open(os.path.join(tmp, "Bio", "Align.py"), "w") as fh:
        fh.write("class MultipleSeqAlignment(list):\n    pass\n")
    with open(os.path.join(tmp, "Bio", "AlignIO.py"), "w") as fh:
        fh.write(
            "from .Align import MultipleSeqAlignment\n"
            "class _Rec:\n"
            "    def __init__(self, i, s):\n        self.id = i; self.seq = s\n"
            "def read(fn, fmt):\n"
            "    recs, name, cur = MultipleSeqAlignment(), None, []\n"
            "    for line in open(fn):\n"
            "        line = line.strip()\n"
            "        if line.startswith('>'):\n"
            "            if name is not None: recs.append(_Rec(name, ''.join(cur)))\n"
            "            name, cur = line[1:], []\n"
            "        elif line: cur.append(line)\n"
            "    if name is not None: recs.append(_Rec(name, ''.join(cur)))\n"
            "    return recs\n")
    sys.path[:0] = [tmp, REF]


def perturbed(x0, L, q):
    x = x0.copy()
    k = np.arange(x.size - L * q, dtype=np.float64)
    x[L * q:] = (0.05 * np.sin(0.37 * k)).astype(x.dtype)
    return x


def plm_golden(tag, path, bio, L, q, seqid, lh, lJ, full_run=None, subsample=None):
    ref = oplm.Reference(path, bio, L, q, seqid, lh, lJ, threads=1)
    out = dict(L=L, q=q, seqid=np.float32(seqid), lambda_h=np.float32(lh), lambda_J=np.float32(lJ),
               X=ref.seqs(), w=ref.weights(), raw_count=len(read_records(path)))
    x0 = ref.init_x()
    f0, g0 = ref.gradient(x0)
    xp = perturbed(x0, L, q)
    f1, g1 = ref.gradient(xp)
    out.update(fx0=np.float32(f0), fx1=np.float32(f1))
    if subsample:
        idx = np.arange(0, x0.size, subsample)
        out.update(idx=idx, h0=x0[:L * q], g0_sub=g0[idx], g1_sub=g1[idx],
                   g0_norm=np.linalg.norm(g0.astype(np.float64)), g1_norm=np.linalg.norm(g1.astype(np.float64)))
    else:
        out.update(x0=x0, g0=g0, g1=g1)
    if full_run:
        for name, (mit, thr) in full_run.items():
            out["run_%s" % name] = ref.backend(path, bio, seqid, lh, lJ, mit, threads=thr)
            out["run_%s_max_iterations" % name] = mit
    ref.close()
    np.savez_compressed(os.path.join(HERE, "plm_%s.npz" % tag), **out)
    print("plm_%s: N'=%d  fx0=%.6f fx1=%.6f" % (tag, out["X"].shape[0], f0, f1))


def mf_golden(tag, path, bio, pseudocount, seqid, stages):
    from pydca.meanfield_dca import meanfield_dca
    inst = meanfield_dca.MeanFieldDCA(path, bio, pseudocount=pseudocount, seqid=seqid)
    out = dict(X=np.array(inst.alignment, dtype=np.int32), w=np.array(inst.sequences_weight),
               pseudocount=pseudocount, seqid=seqid, q=inst.num_site_states)
    if stages:
        fi = inst.get_single_site_freqs()
        out["fi"] = fi.copy()
        out["reg_fi"] = inst.get_reg_single_site_freqs()
        out["fij"] = inst.get_pair_site_freqs()
        reg_fij = inst.get_reg_pair_site_freqs()
        out["reg_fij"] = reg_fij
        corr = inst.construct_corr_mat(out["reg_fi"], reg_fij)
        out["corr_mat"] = corr
        out["couplings"] = inst.compute_couplings(corr)
    fn = inst.compute_sorted_FN()
    apc = inst.compute_sorted_FN_APC()
    out["fn_pairs"] = np.array([p for p, _ in fn], dtype=np.int32)
    out["fn_scores"] = np.array([s for _, s in fn])
    out["apc_pairs"] = np.array([p for p, _ in apc], dtype=np.int32)
    out["apc_scores"] = np.array([s for _, s in apc])
    np.savez_compressed(os.path.join(HERE, "mf_%s.npz" % tag), **out)
    print("mf_%s: N'=%d  top=%s" % (tag, out["X"].shape[0], apc[:2]))
    return apc


def pair_order(sorted_list, L):
    """(pair, score) list -> scores in (0,1),(0,2),... order."""
    d = {tuple(p): s for p, s in sorted_list}
    return np.array([d[(i, j)] for i in range(L - 1) for j in range(i + 1, L)])


def di_golden(tag, path, bio, pseudocount, seqid, plm_seqid=None):
    """Direct-information goldens.  mfDCA: MeanFieldDCA.compute_sorted_DI[_APC] through the stubbed
    import.  plmDCA: PlmDCA.compute_direct_info_unsorted_DI's body (plmdca.py:683-720) replayed with
    the reference's own plmdca/msa_numerics functions on the couplings of the stored reference run
    (plm_<tag>.npz run_a); the class itself cannot be imported without its compiled backend."""
    from pydca.meanfield_dca import meanfield_dca
    from pydca.plmdca import msa_numerics as plm_num
    from pydca.fasta_reader.fasta_reader import get_alignment_int_form
    inst = meanfield_dca.MeanFieldDCA(path, bio, pseudocount=pseudocount, seqid=seqid)
    L, q = inst.sequences_len, inst.num_site_states
    out = dict(L=L, q=q, pseudocount=pseudocount, seqid=seqid,
               mf_di=pair_order(inst.compute_sorted_DI(), L), mf_di_apc=pair_order(inst.compute_sorted_DI_APC(), L))
    # the two remaining public DI methods of the class (meanfield_dca.py:556, :793), called as a user would
    reg_fi = inst.get_reg_single_site_freqs()
    couplings = inst.compute_couplings(inst.construct_corr_mat(reg_fi, inst.get_reg_pair_site_freqs()))
    out["mf_reg_fi"] = np.array(reg_fi)
    out["mf_fields"] = np.array(inst.compute_two_site_model_fields(couplings, reg_fi))
    d = inst.get_site_pair_di_score()
    out["mf_di_dict_keys"] = np.array(list(d.keys()), dtype=np.int32)
    out["mf_di_dict_values"] = np.array(list(d.values()))
    g = np.load(os.path.join(HERE, "plm_%s.npz" % tag))
    x = g["run_a"]
    sid = float(g["seqid"]) if plm_seqid is None else plm_seqid
    blocks = x[L * q:].reshape(L * (L - 1) // 2, q, q)[:, :q - 1, :q - 1]
    couplings = np.array(blocks.reshape(-1))           # get_couplings_no_gap_state, plmdca.py:246-268
    aln = np.array(get_alignment_int_form(path, biomolecule=bio))
    w = plm_num.compute_sequences_weight(alignment_data=aln, sequence_identity=sid)
    fi = plm_num.compute_single_site_freqs(alignment_data=aln, num_site_states=q, seqs_weight=w)
    reg_fi = plm_num.get_reg_single_site_freqs(single_site_freqs=fi, seqs_len=L, num_site_states=q, pseudocount=0.5)
    fields = plm_num.compute_two_site_model_fields(couplings=couplings, reg_fi=reg_fi, seqs_len=L, num_site_states=q)
    di = plm_num.compute_direct_info(couplings=couplings, fields_ij=fields, reg_fi=reg_fi, seqs_len=L,
                                     num_site_states=q)
    out.update(plm_reg_fi=np.array(reg_fi), plm_fields=np.array(fields), plm_di=np.array(di), plm_seqid=sid)
    np.savez_compressed(os.path.join(HERE, "di_%s.npz" % tag), **out)
    print("di_%s: mf top %.6g  plm top %.6g" % (tag, out["mf_di"].max(), out["plm_di"].max()))


def params_golden(tag, path, bio, pseudocount, seqid):
    """MeanFieldDCA.compute_fields / compute_params goldens through the stubbed import
    (meanfield_dca.py:588-752), three rankings with different filters."""
    from pydca.meanfield_dca import meanfield_dca
    inst = meanfield_dca.MeanFieldDCA(path, bio, pseudocount=pseudocount, seqid=seqid)
    L = inst.sequences_len
    fd = inst.compute_fields()
    out = dict(L=L, q=inst.num_site_states, pseudocount=pseudocount, seqid=seqid,
               fields=np.array([fd[i] for i in range(L)]))
    for name, kw in (("default", {}), ("fn_ld2_n5", dict(ranked_by="fn", linear_dist=2, num_site_pairs=5)),
                     ("diapc_ld1_n40", dict(ranked_by="DI_APC", linear_dist=1, num_site_pairs=40))):
        fields, couplings = inst.compute_params(**kw)
        out["%s_field_sites" % name] = np.array([s for s, _ in fields], dtype=np.int32)
        out["%s_fields" % name] = np.array([f for _, f in fields])
        out["%s_pairs" % name] = np.array([p for p, _ in couplings], dtype=np.int32).reshape(-1, 2)
        out["%s_couplings" % name] = np.array([c for _, c in couplings]).reshape(len(couplings), -1)
    np.savez_compressed(os.path.join(HERE, "params_%s.npz" % tag), **out)
    print("params_%s: %d default pairs, first %s" % (tag, len(out["default_pairs"]), out["default_pairs"][:1]))


def runs_golden():
    """plm_runs.npz: (a) the reference's own lbfgs() on its own gradient, 1 thread (deterministic), on the
    three small alignments -- status, iterations, evaluations and the full-precision per-iteration
    (fx, xnorm, gnorm, step, ls), i.e. what plmdcaBackend.cpp:137-146 prints in verbose mode, plus the
    final x, checked here to be bit-equal to what `plmdcaBackend` itself returns; (b) the reference's
    run-to-run spread: three runs (8, 8 and 1 threads) on RF00167 with the default
    lambda = 0.2 (L-1) and 100 iterations, and on the trimmed 71-column alignment with the notebook's
    parameters; stored as FN / FN_APC (float64 from the float32 parameters) with status and counts."""
    from oracle import mf as omf
    out = {}
    small = (("toy_rna", "toy_rna.fa", 2, 10, 5, 1.8, 1.8, 100), ("toy_protein", "toy_protein.fa", 1, 8, 21, 1.0, 5.0, 30),
             ("rf71", "MSA_RF00167_trimmed71.fa", 2, 71, 5, 1.0, 20.0, 500))
    for tag, fname, bio, L, q, lh, lJ, mit in small:
        path = os.path.join(DATA, fname)
        ref = oplm.Reference(path, bio, L, q, 0.8, lh, lJ, threads=1)
        r = ref.lbfgs_run(mit)
        xb = ref.backend(path, bio, 0.8, lh, lJ, mit, threads=1)
        assert np.array_equal(xb, r["x"]), tag           # recorder == the reference's own entry point
        ref.close()
        out.update({tag + "_x": r["x"], tag + "_fx": np.float32(r["fx"]), tag + "_status": r["status"],
                    tag + "_iterations": r["iterations"], tag + "_evaluations": r["evaluations"],
                    tag + "_trace": r["trace"], tag + "_max_iterations": mit,
                    tag + "_lambda_h": np.float32(lh), tag + "_lambda_J": np.float32(lJ)})
        print("runs %-12s status %d iterations %d evaluations %d" % (tag, r["status"], r["iterations"], r["evaluations"]))
    spread = (("rf00167", "MSA_RF00167.fa", 102, 20.2, 20.2, 100), ("rf71", "MSA_RF00167_trimmed71.fa", 71, 1.0, 20.0, 500))
    for tag, fname, L, lh, lJ, mit in spread:
        path = os.path.join(DATA, fname)
        fn, apc, st = [], [], []
        for thr in (8, 8, 1):
            ref = oplm.Reference(path, 2, L, 5, 0.8, lh, lJ, threads=thr)
            r = ref.lbfgs_run(mit)
            ref.close()
            fn.append(omf.plm_fn(r["x"], L, 5, apc_correct=False))
            apc.append(omf.plm_fn(r["x"], L, 5, apc_correct=True))
            st.append((r["status"], r["iterations"], r["evaluations"], thr))
            print("spread %-8s threads %d: status %d iterations %d evaluations %d" % (tag, thr, *st[-1][:3]))
        out.update({"spread_%s_fn" % tag: np.array(fn), "spread_%s_apc" % tag: np.array(apc),
                    "spread_%s_stats" % tag: np.array(st, dtype=np.int32), "spread_%s_max_iterations" % tag: mit,
                    "spread_%s_lambda_h" % tag: np.float32(lh), "spread_%s_lambda_J" % tag: np.float32(lJ)})
    np.savez_compressed(os.path.join(HERE, "plm_runs.npz"), **out)


def _random_backmap_case(rng, alphabet):
    """A reference sequence, an MSA row (with '-') of a related template and SOME gapped local alignment of the
    reference against the gap-free template in Bio.pairwise2's layout (both sequences in full, unaligned ends padded
    with '-', [begin, end) the aligned columns).  Optimality is irrelevant: this feeds the post-alignment logic."""
    n = int(rng.integers(6, 40))
    ref = "".join(rng.choice(list(alphabet), n))
    t = list(ref[int(rng.integers(0, 4)):n - int(rng.integers(0, 4))])
    for _ in range(int(rng.integers(0, 5))):
        if not t:
            break
        pos = int(rng.integers(0, len(t)))
        op = int(rng.integers(0, 3))
        if op == 0:
            t[pos] = str(rng.choice(list(alphabet)))
        elif op == 1:
            t[pos:pos] = list(rng.choice(list(alphabet), int(rng.integers(1, 3))))
        else:
            del t[pos:pos + int(rng.integers(1, 3))]
    t = "".join(t) or str(rng.choice(list(alphabet)))
    row = []
    for ch in t:                                    # the template as an MSA row: gaps sprinkled in
        row.extend("-" * int(rng.integers(0, 3) if rng.random() < 0.3 else 0))
        row.append(ch)
    row.extend("-" * int(rng.integers(0, 3)))
    row = "".join(row)
    sa, sb = int(rng.integers(0, min(4, len(ref)))), int(rng.integers(0, min(4, len(t))))
    ia, ib, ma, mb = sa, sb, [], []
    while ia < len(ref) and ib < len(t) and rng.random() < 0.97:
        r = rng.random()
        if r < 0.8 or not ma:
            ma.append(ref[ia]); mb.append(t[ib]); ia += 1; ib += 1
        elif r < 0.9:
            ma.append("-"); mb.append(t[ib]); ib += 1
        else:
            ma.append(ref[ia]); mb.append("-"); ia += 1
    pre = max(sa, sb)
    ta, tb = ref[ia:], t[ib:]
    post = max(len(ta), len(tb))
    full_a = "-" * (pre - sa) + ref[:sa] + "".join(ma) + ta + "-" * (post - len(ta))
    full_b = "-" * (pre - sb) + t[:sb] + "".join(mb) + tb + "-" * (post - len(tb))
    return dict(ref=ref, row=row, aligned_ref=full_a, aligned_template=full_b, begin=pre, end=pre + len(ma))


def backmap_golden():
    """backmap_cases.json: the reference's own post-alignment logic (SequenceBackmapper.align_subsequences and
    map_to_reference_sequence, sequence_backmapper.py:286-466) on random inputs, and MSATrimmer's column selections
    (msa_trimmer.py:98-207) on a small alignment.  Bio.pairwise2 / Bio.SubsMat are not installed: they are stubbed
    only far enough for the imports to succeed, and the pairwise alignment itself is INJECTED (align_pairs_local and
    find_matching_seqs_from_alignment are replaced per case), so what is pinned is everything downstream of it."""
    import json
    import types
    from pydca.sequence_backmapper import sequence_backmapper as ref_bm
    rng = np.random.default_rng(2024)
    cases = []
    for k in range(300):
        c = _random_backmap_case(rng, "ACGU" if k % 2 else "ARNDCQEGHILKMFPSTWYV")
        obj = object.__new__(ref_bm.SequenceBackmapper)
        obj._SequenceBackmapper__alignment = [c["row"]]
        obj._SequenceBackmapper__ref_sequence = c["ref"]
        obj._SequenceBackmapper__biomolecule = "RNA" if k % 2 else "PROTEIN"
        aln = [(c["aligned_ref"], c["aligned_template"], 0.0, c["begin"], c["end"])]
        obj.align_pairs_local = types.MethodType(lambda self, a, b, score_only=False, _aln=aln: _aln, obj)
        obj.find_matching_seqs_from_alignment = types.MethodType(lambda self, _row=c["row"]: [_row], obj)
        try:
            mapping = obj.map_to_reference_sequence()
            c["mapping"] = sorted([int(a), int(b)] for a, b in mapping.items())
            c["raises"] = None
        except Exception as exc:                      # the reference's walk can run off its input; pinned as such
            c["mapping"] = None
            c["raises"] = type(exc).__name__
        mid_ref = c["aligned_ref"][c["begin"]:c["end"]]
        nmid = len(c["aligned_template"][c["begin"]:c["end"]].replace("-", ""))
        try:
            c["align_subsequences"] = ref_bm.SequenceBackmapper.align_subsequences(
                ref_middle_subseq=mid_ref, template_subseq_in_msa=c["row"], num_res_middle_template=nmid)
        except Exception as exc:
            c["align_subsequences"] = "!" + type(exc).__name__
        cases.append(c)
    with open(os.path.join(HERE, "backmap_cases.json"), "w") as fh:
        json.dump(cases, fh, indent=0)
    print("backmap_cases: %d cases, %d raise" % (len(cases), sum(1 for c in cases if c["raises"])))


def trimmer_golden():
    """trimmer_cases.json: MSATrimmer (msa_trimmer.py:15-207) of the reference on the bundled RF00167 alignment and its
    reference sequence (the matching row is found without pairwise2 only when the first row matches, so the matching
    row is injected as in backmap_golden) -- column selections for several max_gap values and both refseq modes."""
    import json
    import types
    from pydca.msa_trimmer import msa_trimmer as ref_tr
    from pydca.sequence_backmapper import sequence_backmapper as ref_bm
    msa = os.path.join(DATA, "MSA_RF00167.fa")
    refseq = os.path.join(DATA, "ref_RF00167.fa")
    recs = read_records(msa)
    target = read_records(refseq)[0][1].upper()
    match = [s for _n, s in recs if s.replace("-", "").replace(".", "").upper() == target][0]
    orig = ref_bm.SequenceBackmapper.find_matching_seqs_from_alignment
    ref_bm.SequenceBackmapper.find_matching_seqs_from_alignment = lambda self: [match]
    out = {"matching_row": match, "cases": []}
    try:
        for max_gap in (0.0, 0.1, 0.5, 0.9, 1.0):
            tr = ref_tr.MSATrimmer(msa, biomolecule="rna", max_gap=max_gap, refseq_file=refseq)
            gaps = tr.compute_msa_columns_gap_size()
            trimmed = tr.get_msa_trimmed_by_refseq(remove_all_gaps=False)
            out["cases"].append(dict(max_gap=max_gap, gap_size_first=list(gaps[:12]), beyond=list(tr.msa_columns_beyond_max_gap()),
                                     by_gap=list(tr.trim_by_gap_size()), by_refseq=list(tr.trim_by_refseq()),
                                     by_refseq_all=list(tr.trim_by_refseq(remove_all_gaps=True)),
                                     trimmed_first=[list(trimmed[0]), list(trimmed[-1])], trimmed_len=len(trimmed[0][1])))
    finally:
        ref_bm.SequenceBackmapper.find_matching_seqs_from_alignment = orig
    with open(os.path.join(HERE, "trimmer_cases.json"), "w") as fh:
        json.dump(out, fh, indent=0)
    print("trimmer_cases: %d cases" % len(out["cases"]))


def install_bio_alignment_stubs(tmp):
    """Bio.pairwise2 and Bio.SubsMat.MatrixInfo far enough for `import` to succeed (their functions are never reached:
    the callers are replaced per case); Bio.AlignIO records get the attributes MSATrimmer reads (.id, .seq)."""
    with open(os.path.join(tmp, "Bio", "pairwise2.py"), "w") as fh:
        fh.write("class _A:\n    def localds(self, *a, **k):\n        raise NotImplementedError('pairwise2 is not installed')\nalign = _A()\n")
    os.makedirs(os.path.join(tmp, "Bio", "SubsMat"))
    open(os.path.join(tmp, "Bio", "SubsMat", "__init__.py"), "w").close()
    with open(os.path.join(tmp, "Bio", "SubsMat", "MatrixInfo.py"), "w") as fh:
        fh.write("blosum62 = {}\n")


def reader_sweep_cases():
    """Inputs of the reader sweep: every capital and small letter, the three gap characters, a
    duplicate row, CRLF line ends, a line longer than L -- and the inputs on which the reference throws
    (a character its table lacks, a line shorter than L, a line holding only a carriage return)."""
    up = "ABCDEFGHIJKLMNOPQRSTUVWXYZ"
    body = (">upper\n%s\n>lower\n%s\n>gaps\n%s\n\n>dup of upper\n%s\n>long line\n%sACGU\n>mixed\n%s\n"
            % (up, up.lower(), ("-.~" * 9)[:26], up, up[::-1], "aCgU-tT.~nNxX" * 2))
    cases = [("alphabet", 26, body.encode()),
             ("crlf", 4, b">a\r\nACGU\r\n>b\r\nACGT\r\n>c\r\nacgt\r\n"),
             ("dna_letters", 8, b">a\nACGTACGT\n>b\nACGUACGU\n>c\nAC-TAC.T\n"),
             ("err_star", 4, b">a\nACGU\n>b\nAC*U\n"),
             ("err_digit", 4, b">a\nAC1U\n"),
             ("err_space", 4, b">a\nAC U\n"),
             ("err_short", 4, b">a\nACGU\n>b\nACG\n"),
             ("err_blank_cr", 4, b">a\r\nACGU\r\n\r\n>b\r\nACGA\r\n")]
    return cases


def reader_golden():
    """Expected rows (or "the reference throws") of PlmDCA::readSequencesFromFile
    (plmdca_numerics.cpp:685-767) for reader_sweep_cases(), from the compiled reference.  Every case
    runs in a child process: the reference throws C++ exceptions and an out-of-range read of a short
    line is undefined behaviour there."""
    import multiprocessing as mp
    out = {}
    names = []
    for bio, q in ((1, 21), (2, 5)):
        for name, L, text in reader_sweep_cases():
            tag = "%s_%s" % ("protein" if bio == 1 else "rna", name)
            path = os.path.join(tempfile.gettempdir(), "reader_sweep_%s.fa" % tag)
            with open(path, "wb") as fh:
                fh.write(text)
            with mp.get_context("fork").Pool(1) as pool:
                rows = pool.apply(_reader_child, (path, bio, L, q))
            os.unlink(path)
            names.append(tag)
            out[tag + "_text"] = np.frombuffer(text, dtype=np.uint8)
            out[tag + "_L"] = L
            out[tag + "_bio"] = bio
            out[tag + "_throws"] = rows is None
            out[tag + "_rows"] = np.zeros((0, L), np.uint8) if rows is None else rows
            print("reader %-22s %s" % (tag, "throws" if rows is None else "%d rows" % len(rows)))
    out["cases"] = np.array(names)
    np.savez_compressed(os.path.join(HERE, "reader_sweep.npz"), **out)


def _reader_child(path, bio, L, q):
    try:
        ref = oplm.Reference(path, bio, L, q, 0.8, 1.0, 1.0, threads=1)
    except RuntimeError:
        return None
    rows = ref.seqs()
    ref.close()
    return rows


def config_c_golden():
    """config_C_reference.npz: the compiled reference itself at BASELINE.json's config C (tools/gen_msa.py seed 12345,
    L=200 N=10k q=21, lambda_h=1 lambda_J=50) at the initial point and at the perturbed point of the goldens' recipe:
    fx, every 997th gradient element, ||g||, and the reference's OWN float32 error against the float64 oracle (1 and 8
    threads) -- the yardstick for the float32 device path, whose accumulation chains are as long as the reference's."""
    from tools.gen_msa import SEEDS, dedup, generate
    L, N, q, lh, lJ = 200, 10000, 21, 1.0, 50.0
    X = dedup(generate(L, N, q, SEEDS["C"]))
    letters = "ACDEFGHIKLMNPQRSTVWY-"
    tmp = tempfile.mkdtemp(prefix="pydca_c_")
    try:
        path = os.path.join(tmp, "c.fa")
        with open(path, "w") as fh:
            for i, row in enumerate(X):
                fh.write(">s%d\n%s\n" % (i, "".join(letters[v] for v in row)))
        w = oplm.weights(X, 0.8, np.float32)
        x0 = oplm.init_x(X, w, q)
        out = {"stride": 997, "n_unique": X.shape[0]}
        for name, x in (("x0", x0), ("x1", perturbed(x0, L, q))):
            fx_o, g_o = oplm.gradient(X, w.astype(np.float64), q, lh, lJ, x.astype(np.float64), carry=True)
            errs = []
            for thr in (1, 8):
                ref = oplm.Reference(path, 1, L, q, 0.8, lh, lJ, threads=thr)
                assert np.array_equal(ref.seqs(), X) and np.array_equal(ref.weights(), w)
                if name == "x0":
                    np.testing.assert_allclose(ref.init_x(), x0, rtol=2e-6, atol=2e-6)
                fx_r, g_r = ref.gradient(x.astype(np.float32))
                ref.close()
                errs.append(float(np.linalg.norm(g_r.astype(np.float64) - g_o) / np.linalg.norm(g_o)))
                if thr == 1:
                    out.update({name + "_fx": np.float32(fx_r), name + "_g_sub": g_r[::997].copy(), name + "_gnorm": np.float64(np.linalg.norm(g_r.astype(np.float64)))})
            out[name + "_ref_err_vs_f64"] = np.array(errs)
            print("config C %s: reference float32 vs float64 oracle rel. error %.3e (1 thread) %.3e (8 threads), fx %.9g" % (name, errs[0], errs[1], out[name + "_fx"]))
        np.savez_compressed(os.path.join(HERE, "config_C_reference.npz"), **out)
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--skip-slow", action="store_true")
    ap.add_argument("--only-reader", action="store_true", help="regenerate only reader_sweep.npz")
    ap.add_argument("--only-runs", action="store_true", help="regenerate only plm_runs.npz (needs data/ present)")
    ap.add_argument("--only-backmap", action="store_true", help="regenerate only backmap_cases.json / trimmer_cases.json")
    ap.add_argument("--only-params", action="store_true", help="regenerate only params_*.npz")
    ap.add_argument("--only-di", action="store_true", help="regenerate only di_*.npz (needs plm_*.npz present)")
    ap.add_argument("--only-config-c", action="store_true", help="regenerate only config_C_reference.npz")
    args = ap.parse_args()
    if args.only_config_c:
        oplm.build(ref=True)
        config_c_golden()
        return
    if args.only_reader:
        oplm.build(ref=True)
        reader_golden()
        return
    if args.only_runs:
        oplm.build(ref=True)
        runs_golden()
        return
    if args.only_backmap:
        tmp = tempfile.mkdtemp(prefix="pydca_stubs_")
        try:
            install_stubs(tmp)
            install_bio_alignment_stubs(tmp)
            backmap_golden()
            trimmer_golden()
        finally:
            shutil.rmtree(tmp, ignore_errors=True)
        return
    if args.only_params:
        tmp = tempfile.mkdtemp(prefix="pydca_stubs_")
        try:
            install_stubs(tmp)
            params_golden("toy_rna", os.path.join(DATA, "toy_rna.fa"), "rna", 0.5, 0.8)
            params_golden("toy_protein", os.path.join(DATA, "toy_protein.fa"), "protein", 0.5, 0.8)
        finally:
            shutil.rmtree(tmp, ignore_errors=True)
        return
    if args.only_di:
        tmp = tempfile.mkdtemp(prefix="pydca_stubs_")
        try:
            install_stubs(tmp)
            di_golden("toy_rna", os.path.join(DATA, "toy_rna.fa"), "rna", 0.5, 0.8)
            di_golden("toy_protein", os.path.join(DATA, "toy_protein.fa"), "protein", 0.5, 0.8)
            di_golden("rf71", os.path.join(DATA, "MSA_RF00167_trimmed71.fa"), "rna", 0.5, 0.8)
        finally:
            shutil.rmtree(tmp, ignore_errors=True)
        return
    os.makedirs(DATA, exist_ok=True)
    oplm.build(ref=True)

    # ---- data files ---------------------------------------------------------
    rf = os.path.join(DATA, "MSA_RF00167.fa")
    pf = os.path.join(DATA, "PF02826.faa")
    shutil.copyfile(os.path.join(REF, "examples", "MSA_RF00167.fa"), rf)
    shutil.copyfile(os.path.join(REF, "examples", "ref_RF00167.fa"), os.path.join(DATA, "ref_RF00167.fa"))
    shutil.copyfile(os.path.join(REF, "tests", "tests_input", "PF02826.faa"), pf)
    # inputs of the reference's sequence_backmapper / trimming tests (tests/input_files_path.py)
    extra = ["MSA_RF00059_trimmed_gap_treshold_50.fa", "ref_seq_RF00059.faa", "ref_seq_RF00059_test1.faa",
             "ref_seq_RF00059_test2.faa", "ref_seq_RF00059_test3.faa", "ref_seq_RF00059_test4.faa", "ref_seq_PF02826.faa"]
    for name in extra:
        shutil.copyfile(os.path.join(REF, "tests", "tests_input", name), os.path.join(DATA, name))
    for p in [rf, pf, os.path.join(DATA, "ref_RF00167.fa")] + [os.path.join(DATA, n) for n in extra]:
        os.chmod(p, 0o644)
    recs = read_records(rf)
    refrec = [s for n, s in recs if "REFERENCE" in n][0]
    keep = [k for k, ch in enumerate(refrec) if ch not in "-.~"]
    rf71 = os.path.join(DATA, "MSA_RF00167_trimmed71.fa")
    write_fasta(rf71, [(n, "".join(s[k] for k in keep)) for n, s in recs])
    assert len(keep) == 71
    toy_rna = os.path.join(DATA, "toy_rna.fa")
    toy_prot = os.path.join(DATA, "toy_protein.fa")
    make_toy(toy_rna, "ACGU-", 48, 10, 7)
    make_toy(toy_prot, "ACDEFGHIKLMNPQRSTVWY-", 40, 8, 11)

    # ---- plmDCA via the compiled reference -------------------------------------
    reader_golden()
    plm_golden("toy_rna", toy_rna, 2, 10, 5, 0.8, 1.8, 1.8, full_run={"a": (100, 1)})
    plm_golden("toy_protein", toy_prot, 1, 8, 21, 0.8, 1.0, 5.0, full_run={"a": (30, 1)})
    plm_golden("rf71", rf71, 2, 71, 5, 0.8, 1.0, 20.0, full_run={"a": (500, 1), "b": (500, 8)})
    plm_golden("rf00167", rf, 2, 102, 5, 0.8, 20.2, 20.2, subsample=97)
    if not args.skip_slow:
        plm_golden("pf02826", pf, 1, 195, 21, 0.8, 1.0, 50.0, subsample=9973)
    runs_golden()
    if not args.skip_slow:
        config_c_golden()

    # ---- mfDCA via the stubbed import of the reference ---------------------------
    tmp = tempfile.mkdtemp(prefix="pydca_stubs_")
    try:
        install_stubs(tmp)
        apc = mf_golden("rf71", rf71, "rna", 0.5, 0.8, stages=False)
        for (pair, val), (rp, rv) in zip(KAT_MF, apc):
            assert tuple(rp) == pair and abs(rv - val) <= 1e-12 * abs(val), (pair, val, rp, rv)
        print("stubbed import reproduces the notebook mfDCA KAT")
        mf_golden("toy_rna", toy_rna, "rna", 0.5, 0.8, stages=True)
        mf_golden("toy_protein", toy_prot, "protein", 0.5, 0.8, stages=True)
        mf_golden("toy_rna_theta02_seqid1", toy_rna, "rna", 0.2, 1.0, stages=True)
        mf_golden("rf00167", rf, "rna", 0.5, 0.8, stages=False)
        di_golden("toy_rna", toy_rna, "rna", 0.5, 0.8)
        di_golden("toy_protein", toy_prot, "protein", 0.5, 0.8)
        di_golden("rf71", rf71, "rna", 0.5, 0.8)
        params_golden("toy_rna", toy_rna, "rna", 0.5, 0.8)
        params_golden("toy_protein", toy_prot, "protein", 0.5, 0.8)
        if not args.skip_slow:
            mf_golden("pf02826", pf, "protein", 0.5, 0.8, stages=False)
        install_bio_alignment_stubs(tmp)
        backmap_golden()
        trimmer_golden()
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    np.savez(os.path.join(HERE, "kat_notebook.npz"),
             mf_pairs=np.array([p for p, _ in KAT_MF]), mf_scores=np.array([s for _, s in KAT_MF]),
             plm_pairs=np.array([p for p, _ in KAT_PLM]), plm_scores=np.array([s for _, s in KAT_PLM]))


if __name__ == "__main__":
    main()
