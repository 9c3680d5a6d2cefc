// Host-side MSA reader with the semantics of PlmDCA::readSequencesFromFile
// (pydca/plmdca/plmdca_numerics.cpp:685-767): every non-empty line not starting with '>'
// is one sequence, only its first L characters are used (upper-cased), residues map to
// 0-based codes with gap = q-1, rows already seen are dropped (first occurrence kept).
// Where the reference throws (unopenable file :743-746, character missing from its table
// :752) this returns an error code instead.
#include <cctype>
#include <fstream>
#include <string>
#include <unordered_set>

#include "dca_internal.h"

namespace {

// residue tables of plmdca_numerics.cpp:708-717 (protein) and :722-731 (RNA: ACGU, the three gap
// characters and every other capital letter -- 'T' included, :729 -- are the gap state)
struct CodeTable {
    int8_t protein[256];
    int8_t rna[256];
    CodeTable()
    {
        for (int c = 0; c < 256; ++c) protein[c] = rna[c] = -1;
        const char* aa = "ACDEFGHIKLMNPQRSTVWY";
        for (int k = 0; aa[k]; ++k) protein[(unsigned char)aa[k]] = (int8_t)k;
        for (const char* p = "-.~BJOUXZ"; *p; ++p) protein[(unsigned char)*p] = 20;
        rna[(unsigned char)'A'] = 0; rna[(unsigned char)'C'] = 1; rna[(unsigned char)'G'] = 2; rna[(unsigned char)'U'] = 3;
        for (const char* p = "-~.BDEFHIJKLMNOPQRSTVWXYZ"; *p; ++p) rna[(unsigned char)*p] = 4;
    }
};
const CodeTable kCodes;

}  // namespace

int dca_read_msa_impl(const char* path, int biomolecule, int L, uint8_t* out, int capacity, int* raw_count)
{
    if (!path || L <= 0 || (biomolecule != DCA_BIOMOLECULE_PROTEIN && biomolecule != DCA_BIOMOLECULE_RNA)) {
        dca_set_error("dca_read_msa: bad arguments");
        return DCA_ERR_ARG;
    }
    std::ifstream in(path);
    if (in.fail()) {
        dca_set_error("Unable to open file %s", path);
        return DCA_ERR_IO;
    }
    const int8_t* table = biomolecule == DCA_BIOMOLECULE_PROTEIN ? kCodes.protein : kCodes.rna;
    std::unordered_set<std::string> seen;
    std::string line, row((size_t)L, '\0');
    int nuniq = 0, nraw = 0;
    while (std::getline(in, line)) {
        // like the reference, only '\n' ends a line (std::getline, :748): a '\r' beyond column L is never
        // looked at, one inside the first L columns is a character its table lacks
        if (line.empty() || line[0] == '>') continue;
        if ((int)line.size() < L) {
            dca_set_error("sequence line %d of %s is shorter than %d", nraw + 1, path, L);
            return DCA_ERR_RESIDUE;
        }
        for (int s = 0; s < L; ++s) {
            const int code = table[(unsigned char)std::toupper((unsigned char)line[s])];
            if (code < 0) {
                dca_set_error("character '%c' of sequence line %d is not in the residue table", line[s], nraw + 1);
                return DCA_ERR_RESIDUE;
            }
            row[s] = (char)code;
        }
        ++nraw;
        if (!seen.insert(row).second) continue;
        if (out) {
            if (nuniq >= capacity) {
                dca_set_error("dca_read_msa: capacity %d too small", capacity);
                return DCA_ERR_ARG;
            }
            memcpy(out + (size_t)nuniq * L, row.data(), (size_t)L);
        }
        ++nuniq;
    }
    if (raw_count) *raw_count = nraw;
    return nuniq;
}

// ---------------------------------------------------------------------------------------------
// Local pairwise alignment (Smith-Waterman with Gotoh's affine gaps) for the reference-sequence
// back-mapping (SequenceBackmapper.align_pairs_local, sequence_backmapper.py:186-230, which calls
// Bio.pairwise2.align.localds -- biopython 1.74, not part of the reference tree).  Gap of length n
// costs open + (n-1)*extend as in pairwise2 (penalize_extend_when_opening = False).  sub: 26x26
// substitution scores indexed by letter - 'A'.
//
// dca_sw_scores: best local score of `ref` against each of `nseq` sequences packed back to back in
// `seqs` (offsets[k] .. offsets[k+1]); this is the search loop of find_matching_seqs_from_alignment
// (:233-283).  dca_sw_align: one alignment with traceback.  Tie rules (pairwise2 returns a list and
// the reference takes element 0; its ordering is not documented): the end cell is the first maximum
// in row-major order and the traceback prefers diagonal, then a gap in the second sequence, then a
// gap in the reference.
namespace {
struct SwRows {
    std::vector<int> H, E;
};

inline int sub_score(const int* sub, unsigned char a, unsigned char b)
{
    const int ia = a - 'A', ib = b - 'A';
    if (ia < 0 || ia >= 26 || ib < 0 || ib >= 26) return -4;
    return sub[ia * 26 + ib];
}

int sw_score_one(const char* a, int la, const char* b, int lb, const int* sub, int open, int ext, SwRows& w)
{
    // rows over a (reference), columns over b
    w.H.assign(lb + 1, 0);
    w.E.assign(lb + 1, -(1 << 28));      // best score ending in a gap in a (vertical move), per column
    int best = 0;
    for (int i = 1; i <= la; ++i) {
        int diag = 0, F = -(1 << 28);    // F: gap in b (horizontal move) along the row
        w.H[0] = 0;
        for (int j = 1; j <= lb; ++j) {
            const int up = w.H[j];
            w.E[j] = std::max(w.E[j] + ext, up + open);
            F = std::max(F + ext, w.H[j - 1] + open);
            int h = diag + sub_score(sub, (unsigned char)a[i - 1], (unsigned char)b[j - 1]);
            h = std::max(h, std::max(w.E[j], F));
            h = std::max(h, 0);
            diag = up;
            w.H[j] = h;
            best = std::max(best, h);
        }
    }
    return best;
}
}  // namespace

extern "C" int dca_sw_scores(const char* ref, int lref, const char* seqs, const int* offsets, int nseq, const int* sub,
                             int gap_open, int gap_extend, int* scores_out)
{
    if (!ref || !seqs || !offsets || !sub || !scores_out || lref < 0 || nseq < 0) return DCA_ERR_ARG;
    SwRows w;
    for (int k = 0; k < nseq; ++k)
        scores_out[k] = sw_score_one(ref, lref, seqs + offsets[k], offsets[k + 1] - offsets[k], sub, gap_open, gap_extend, w);
    return DCA_OK;
}

// aligned_a / aligned_b: capacity la + lb + 1 each; receive the aligned REGION only (with '-').
// start_a / start_b: 0-based index of the first residue of the region in each sequence.
extern "C" int dca_sw_align(const char* a, int la, const char* b, int lb, const int* sub, int gap_open, int gap_extend,
                            int* score_out, int* start_a, int* start_b, char* aligned_a, char* aligned_b, int* aligned_len)
{
    if (!a || !b || !sub || !score_out || !start_a || !start_b || !aligned_a || !aligned_b || !aligned_len) return DCA_ERR_ARG;
    const int W = lb + 1;
    const int NEG = -(1 << 28);
    std::vector<int> H((size_t)(la + 1) * W, 0), E((size_t)(la + 1) * W, NEG), F((size_t)(la + 1) * W, NEG);
    int best = 0, bi = 0, bj = 0;
    for (int i = 1; i <= la; ++i)
        for (int j = 1; j <= lb; ++j) {
            const size_t c = (size_t)i * W + j;
            E[c] = std::max(E[c - W] + gap_extend, H[c - W] + gap_open);     // gap in b's row direction: consumes a[i-1]
            F[c] = std::max(F[c - 1] + gap_extend, H[c - 1] + gap_open);     // consumes b[j-1]
            int h = H[c - W - 1] + sub_score(sub, (unsigned char)a[i - 1], (unsigned char)b[j - 1]);
            h = std::max(h, std::max(E[c], F[c]));
            h = std::max(h, 0);
            H[c] = h;
            if (h > best) { best = h; bi = i; bj = j; }
        }
    *score_out = best;
    std::string ra, rb;
    int i = bi, j = bj, state = 0;    // 0: in H, 1: in E (gap in b), 2: in F (gap in a)
    while (i > 0 && j > 0) {
        const size_t c = (size_t)i * W + j;
        if (state == 0) {
            if (H[c] == 0) break;
            if (H[c] == H[c - W - 1] + sub_score(sub, (unsigned char)a[i - 1], (unsigned char)b[j - 1])) {
                ra.push_back(a[i - 1]); rb.push_back(b[j - 1]); --i; --j;
            } else if (H[c] == E[c]) state = 1;
            else state = 2;
        } else if (state == 1) {
            ra.push_back(a[i - 1]); rb.push_back('-');
            if (E[c] == H[c - W] + gap_open) state = 0;
            --i;
        } else {
            ra.push_back('-'); rb.push_back(b[j - 1]);
            if (F[c] == H[c - 1] + gap_open) state = 0;
            --j;
        }
    }
    *start_a = i;
    *start_b = j;
    const int n = (int)ra.size();
    for (int k = 0; k < n; ++k) { aligned_a[k] = ra[n - 1 - k]; aligned_b[k] = rb[n - 1 - k]; }
    aligned_a[n] = aligned_b[n] = 0;
    *aligned_len = n;
    return DCA_OK;
}
