// TEST INFRASTRUCTURE ONLY -- not part of the shipped product.
//
// Thin C-ABI driver around the *reference's own* C++ class so that tests/,
// tests/golden/make_golden.py and bench.py's cpu_baseline leg can call the real
// reference numerics (PlmDCA::computeSeqsWeight / initFieldsAndCouplings /
// gradient, /root/reference/pydca/plmdca/plmdca_numerics.cpp:611,207,436).
//
// This file contains no reference code: it only #includes the reference header
// from where it lies under /root/reference at build time (see oracle/Makefile,
// target `ref`) and is linked against the reference's own translation units.
// Output goes to oracle/_ref/ (git-ignored, travels with gpurun).
#include "include/plmdca.h"   // resolved with -I/root/reference/pydca/plmdca
#include <cstdint>
#include <cstring>

extern "C" {

void* ref_open(const char* msa_file, unsigned biomolecule, unsigned L, unsigned q,
               float seqid, float lambda_h, float lambda_J, unsigned threads) {
    try {
        return new PlmDCA(msa_file, biomolecule, L, q, seqid, lambda_h, lambda_J, threads);
    } catch (...) {
        return nullptr;
    }
}

void ref_close(void* h) { delete static_cast<PlmDCA*>(h); }

// number of unique sequences the reference reader keeps (plmdca_numerics.cpp:685-767)
int ref_read_seqs(void* h, uint8_t* out, int capacity_rows, int L) {
    auto seqs = static_cast<PlmDCA*>(h)->readSequencesFromFile();
    int n = (int)seqs.size();
    if (out) {
        for (int r = 0; r < n && r < capacity_rows; ++r)
            for (int c = 0; c < L; ++c) out[(size_t)r * L + c] = (uint8_t)seqs[r][c];
    }
    return n;
}

int ref_weights(void* h, float* w, int capacity) {
    auto ws = static_cast<PlmDCA*>(h)->computeSeqsWeight();
    int n = (int)ws.size();
    for (int i = 0; i < n && i < capacity; ++i) w[i] = ws[i];
    return n;
}

void ref_init_x(void* h, float* x) { static_cast<PlmDCA*>(h)->initFieldsAndCouplings(x); }

float ref_gradient(void* h, const float* x, float* g) {
    return static_cast<PlmDCA*>(h)->gradient(x, g);
}

}  // extern "C"
