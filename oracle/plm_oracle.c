/* TEST INFRASTRUCTURE ONLY -- CPU oracle, not part of the shipped product.
 *
 * plmDCA oracle: a plain-C restatement of the reference's plmDCA hot path
 * (/root/reference/pydca/plmdca/: plmdca_numerics.cpp, plmdcaBackend.cpp,
 * lbfgs/lib/lbfgs.cpp).  Only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may load this library; the product (pydca_amd) never does.
 *
 * Pinned against the compiled reference (oracle/_ref, see oracle/Makefile and
 * tests/golden/make_golden.py): reader/dedup, weights, initial x and the
 * (x, fx, g) triples of PlmDCA::gradient -- see tests/test_oracle_golden.py.
 *
 * Build: make -C oracle        -> oracle/_build/liboracle_plm.so
 */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <ctype.h>

/* libLBFGS status codes the reference can return (lbfgs/include/lbfgs.h:76-149) */
#define PLM_LBFGS_SUCCESS 0
#define PLM_LBFGS_ALREADY_MINIMIZED 2
#define PLM_LBFGSERR_OUTOFINTERVAL (-1003)
#define PLM_LBFGSERR_INCORRECT_TMINMAX (-1002)
#define PLM_LBFGSERR_ROUNDING_ERROR (-1001)
#define PLM_LBFGSERR_MINIMUMSTEP (-1000)
#define PLM_LBFGSERR_MAXIMUMSTEP (-999)
#define PLM_LBFGSERR_MAXIMUMLINESEARCH (-998)
#define PLM_LBFGSERR_MAXIMUMITERATION (-997)
#define PLM_LBFGSERR_WIDTHTOOSMALL (-996)
#define PLM_LBFGSERR_INVALIDPARAMETERS (-995)
#define PLM_LBFGSERR_INCREASEGRADIENT (-994)

/* plmdcaBackend.cpp:68-75 over the defaults of lbfgs.cpp:116-121 */
#define PLM_LBFGS_M 5
#define PLM_EPSILON 1e-3
#define PLM_MAX_LINESEARCH 5
#define PLM_FTOL 1e-4
#define PLM_GTOL 0.9
#define PLM_XTOL 1.0e-16
#define PLM_MIN_STEP 1e-20
#define PLM_MAX_STEP 1e20

/* P = L*q + L(L-1)/2*q^2, plmdca_numerics.cpp:40-42 */
static size_t plm_num_params(int L, int q)
{
    return (size_t)L * q + (size_t)L * (L - 1) / 2 * (size_t)q * q;
}

/* index of pair (i<j) in (0,1),(0,2)...(L-2,L-1) order, plmdca_numerics.cpp:340 */
static size_t plm_pair_index(int L, int i, int j)
{
    return (size_t)L * (L - 1) / 2 - (size_t)(L - i) * (L - i - 1) / 2 + (size_t)(j - i - 1);
}

size_t oracle_num_params(int L, int q) { return plm_num_params(L, q); }

/* residue table of readSequencesFromFile, plmdca_numerics.cpp:699-732.
 * biomolecule: 1 = protein (q=21), 2 = RNA (q=5).  Returns -1 for characters
 * the reference's map lacks (it throws std::out_of_range at :752).  The RNA map
 * holds all 26 letters: everything but A, C, G, U is the gap state, 'T' too (:729). */
int oracle_residue_code(int biomolecule, int ch)
{
    ch = toupper(ch);
    if (biomolecule == 1) {
        static const char aa[] = "ACDEFGHIKLMNPQRSTVWY";
        const char* p = ch ? strchr(aa, ch) : NULL;
        if (p) return (int)(p - aa);
        if (ch == '-' || ch == '.' || ch == '~' || ch == 'B' || ch == 'J' || ch == 'O' ||
            ch == 'U' || ch == 'X' || ch == 'Z') return 20;
        return -1;
    }
    switch (ch) {
        case 'A': return 0; case 'C': return 1; case 'G': return 2; case 'U': return 3;
        case '-': case '~': case '.': return 4;
        default: return (ch >= 'A' && ch <= 'Z') ? 4 : -1;
    }
}

/* reader + first-occurrence dedup, plmdca_numerics.cpp:685-767: every non-empty
 * line that does not start with '>' is one sequence; only its first L characters
 * are used.  Returns number of unique rows written (row-major uint8, 0-based,
 * gap = q-1), or <0: -1 cannot open, -2 unknown character / short line,
 * -3 capacity too small.  raw_count receives the number of sequence lines. */
int oracle_read_msa(const char* path, int biomolecule, int L, uint8_t* out, int capacity, int* raw_count)
{
    FILE* fp = fopen(path, "r");
    if (!fp) return -1;
    size_t cap = 1 << 16, len;
    char* line = (char*)malloc(cap);
    uint8_t* row = (uint8_t*)malloc(L);
    /* open hash of row indices for O(N) dedup (reference: std::find, O(N^2)) */
    size_t hsize = 1; while (hsize < (size_t)capacity * 2 + 16) hsize <<= 1;
    int* table = (int*)malloc(hsize * sizeof(int));
    for (size_t i = 0; i < hsize; ++i) table[i] = -1;
    int nuniq = 0, nraw = 0, rc = 0;
    while (fgets(line, (int)cap, fp)) {
        len = strlen(line);
        while (len == cap - 1 && line[len - 1] != '\n') {   /* grow for long lines */
            cap *= 2; line = (char*)realloc(line, cap);
            if (!fgets(line + len, (int)(cap - len), fp)) break;
            len = strlen(line);
        }
        while (len && line[len - 1] == '\n') line[--len] = 0;   /* std::getline: only '\n' ends a line */
        if (!len || line[0] == '>') continue;
        if ((int)len < L) { rc = -2; break; }
        uint64_t h = 1469598103934665603ull;
        for (int s = 0; s < L; ++s) {
            int c = oracle_residue_code(biomolecule, (unsigned char)line[s]);
            if (c < 0) { rc = -2; break; }
            row[s] = (uint8_t)c;
            h = (h ^ (uint64_t)c) * 1099511628211ull;
        }
        if (rc) break;
        ++nraw;
        size_t slot = (size_t)h & (hsize - 1);
        int dup = 0;
        while (table[slot] >= 0) {
            if (!memcmp(out + (size_t)table[slot] * L, row, L)) { dup = 1; break; }
            slot = (slot + 1) & (hsize - 1);
        }
        if (dup) continue;
        if (nuniq >= capacity) { rc = -3; break; }
        memcpy(out + (size_t)nuniq * L, row, L);
        table[slot] = nuniq++;
    }
    fclose(fp); free(line); free(row); free(table);
    if (raw_count) *raw_count = nraw;
    return rc ? rc : nuniq;
}

#define CAT_(a, b) a##b
#define CAT(a, b) CAT_(a, b)

#define REAL float
#define FN(name) CAT(name, _f32)
#define REAL_EXP expf
#define REAL_LOG logf
#define REAL_SQRT sqrtf
#define REAL_ABS fabsf
#include "plm_oracle_impl.h"
#undef REAL
#undef FN
#undef REAL_EXP
#undef REAL_LOG
#undef REAL_SQRT
#undef REAL_ABS

/* The float64 instantiation sums the OBJECTIVE with Neumaier compensation (ORACLE_COMPENSATED_FX): the reference has no
 * float64 build -- its float32 summation order is kept bit for bit by the instantiation above -- and the float64 oracle
 * is this repository's deterministic parity target.  With plain sums the objective of two float64
 * implementations differs by ~1e-13 (N*L terms in two orders); the line search interpolates on DIFFERENCES of objective
 * values, and over 100 iterations of an optimisation that does not converge that rounding noise grows to 6e-4 in the
 * scores at config E (profiles/r03_e_sensitivity_cap100_plain_sums.json).  A compensated sum does not depend on the order.
 * Round 4: the same holds for the optimiser's dot products (g.d, y.s, y.y, x.x, g.g and the two-loop recursion's s.d, y.d;
 * P = 5.5e7 products at config D, ~7e-13 of order-dependent rounding in a plain sequential sum), so vdot is compensated
 * too in this instantiation; the device sums the same rounded products in double-double. */
#define ORACLE_COMPENSATED_FX 1
/* Round 4: the float64 instantiation fixes the ORDER of its floating-point additions where the reference's float32 code
 * leaves a choice that no float64 implementation is bound to (ORACLE_CANONICAL_F64; the float32 instantiation above keeps
 * the reference's order bit for bit):
 *   logits     z = ((sum_j J) + h) + carry: coupling rows first, ascending j from zero (the reference starts from the
 *              carry: plmdca_numerics.cpp:499-515) -- a chunk-parallel scan cannot know the carry before the sum;
 *   residual   one rounded value r = w p(a) - w delta(a, x) is added to every sum (the reference adds -w and +w p(a) one
 *              after the other, :541-566);
 *   fields     the gradient sums of the fields are compensated like the objective (order-independent).
 *   chains     (round 5) the per-slot sums of the coupling gradient run over blocks of ORACLE_CANONICAL_BLOCK = 16384
 *              consecutive sequences, each block summed from zero in ascending n, the block sums added in ascending
 *              block order ((B0 + B1) + B2) + ... -- a fixed order that a device can follow with as many independent
 *              chains as there are blocks (round 4's single chain over all N left 60 workgroups on 256 CUs at config E:
 *              13 blocks there, 4 at config D); alignments of at most one block (config C, the test alignments) are one
 *              plain chain as before.  Why 16384 and not less: a device workgroup that walks several blocks has to add
 *              each finished block into the running sum in memory (a read-modify-write of its part of the gradient
 *              table), which at config D cost 3 ms of a 34 ms evaluation with blocks of 4096 (12 such passes, measured).
 * The two site views are merged as (2 lambda x + view_i) + view_j, as before.  Why: at config E two float64 runs that differ in nothing
 * but the order of these sums (1e-13 relative) end 7.5e-5 apart in FN after the reference's 100 iterations
 * (profiles/r04_sensitivity_E_cap100.json) -- the optimisation does not converge and amplifies rounding by ~1.2 x per
 * iteration -- so a device-vs-oracle comparison at 1e-4 needs BOTH to add in the same order; with it the device's
 * float64 gradient equals this oracle's bit for bit up to the last-place differences of the two exp() implementations. */
#ifndef ORACLE_PLAIN_F64            /* -DORACLE_PLAIN_F64: the reference's order in float64 too (tests: the two differ by rounding only) */
#define ORACLE_CANONICAL_F64 1
#ifndef ORACLE_CANONICAL_BLOCK
#define ORACLE_CANONICAL_BLOCK 16384  /* tests build a single-chain variant with a larger value */
#endif
#endif
#define REAL double
#define FN(name) CAT(name, _f64)
#define REAL_EXP exp
#define REAL_LOG log
#define REAL_SQRT sqrt
#define REAL_ABS fabs
#include "plm_oracle_impl.h"
