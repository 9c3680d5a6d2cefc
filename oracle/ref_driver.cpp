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

// ---- full optimisation through the reference's own lbfgs() -------------------------------------
// Same call as ObjectiveFunction::run (plmdcaBackend.cpp:46-96: m = 5, epsilon = 1e-3,
// max_linesearch = 5, ftol = 1e-4, everything else lbfgs_parameter_init), but with callbacks written
// here that record what the backend drops: the exit status, the number of objective evaluations and
// the full-precision per-iteration (fx, xnorm, gnorm, step, ls) that its verbose mode prints with %f.
#include "lbfgs.h"      // resolved with -I/root/reference/pydca/plmdca/lbfgs/include

namespace {
struct RunRecorder {
    PlmDCA* inst;
    int evaluations;
    int iterations;
    float* trace;       // 5 floats per iteration
    int trace_cap;
};
lbfgsfloatval_t rec_evaluate(void* p, const lbfgsfloatval_t* x, lbfgsfloatval_t* g, const int, const lbfgsfloatval_t) {
    RunRecorder* r = static_cast<RunRecorder*>(p);
    ++r->evaluations;
    return r->inst->gradient(x, g);
}
int rec_progress(void* p, const lbfgsfloatval_t*, const lbfgsfloatval_t*, const lbfgsfloatval_t fx,
                 const lbfgsfloatval_t xnorm, const lbfgsfloatval_t gnorm, const lbfgsfloatval_t step, int, int k, int ls) {
    RunRecorder* r = static_cast<RunRecorder*>(p);
    r->iterations = k;
    if (r->trace && k >= 1 && k <= r->trace_cap) {
        float* t = r->trace + 5 * (size_t)(k - 1);
        t[0] = fx; t[1] = xnorm; t[2] = gnorm; t[3] = step; t[4] = (float)ls;
    }
    return 0;
}
}  // namespace

extern "C" int ref_lbfgs_run(void* h, int n, unsigned max_iterations, float* x_out, float* fx_out,
                             int* iterations, int* evaluations, float* trace, int trace_cap) {
    PlmDCA* inst = static_cast<PlmDCA*>(h);
    lbfgsfloatval_t* x = lbfgs_malloc(n);
    if (!x) return 1;
    lbfgs_parameter_t param;
    lbfgs_parameter_init(&param);
    param.epsilon = 1E-3;
    param.max_iterations = (int)max_iterations;
    param.max_linesearch = 5;
    param.ftol = 1E-4;
    param.m = 5;
    inst->initFieldsAndCouplings(x);
    RunRecorder rec{inst, 0, 0, trace, trace_cap};
    lbfgsfloatval_t fx = 0;
    int ret = lbfgs(n, x, &fx, rec_evaluate, rec_progress, &rec, &param);
    memcpy(x_out, x, sizeof(float) * (size_t)n);
    lbfgs_free(x);
    if (fx_out) *fx_out = fx;
    if (iterations) *iterations = rec.iterations;
    if (evaluations) *evaluations = rec.evaluations;
    return ret;
}
