// Internal declarations shared by the translation units of libdca_hip.so.
// gfx950 (MI355X / CDNA4) only: wave = 64 lanes, 160 KiB LDS per CU, 8 XCDs.
#pragma once
#include <hip/hip_runtime.h>

#include <atomic>
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <map>
#include <string>
#include <vector>

#include "../../include/dca_hip.h"

void dca_set_error(const char* fmt, ...);

#define HIP_TRY(expr)                                                                   \
    do {                                                                                \
        hipError_t _e = (expr);                                                         \
        if (_e != hipSuccess) {                                                         \
            dca_set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
            return DCA_ERR_HIP;                                                         \
        }                                                                               \
    } while (0)

#define DCA_TRY(expr)                  \
    do {                               \
        int _rc = (expr);              \
        if (_rc != DCA_OK) return _rc; \
    } while (0)

// ---- capi.cpp : device allocations.  Blocks of at least 1 MiB go back to a process-wide, per-device cache
// when they are freed and later requests of a similar size are served from it (zero-filled), because
// hipMalloc / hipFree of GB-sized blocks costs tens of milliseconds per call on some hosts -- more than
// the whole mfDCA chain.  dca_dev_free waits for the device like hipFree does.
hipError_t dca_dev_malloc(void** p, size_t bytes, bool zero_recycled = true);   // false: buffers their first kernel overwrites completely
hipError_t dca_dev_free(void* p);

static inline size_t round_up(size_t v, size_t m) { return (v + m - 1) / m * m; }
static inline int ceil_div(int a, int b) { return (a + b - 1) / b; }

// number of XCDs on MI355X; workgroup b is observed to run on XCD b % 8
// (used for L2 locality only, never for correctness)
constexpr int kNumXcd = 8;

struct KernelClock {
    double ms = 0.0;
    int launches = 0;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> pending;
};

struct PlmEngineBase;
struct MfEngine;

#define DCA_SIDE_DEPTHS 3

struct dca_ctx {
    int device = 0;
    int precision = DCA_F32;
    hipStream_t stream = nullptr;

    // alignment (device): row-major bytes, row stride Ls (multiple of 128), zero padded
    int N = 0, L = 0, q = 0, Ls = 0;
    uint8_t* dX = nullptr;
    std::vector<uint8_t> hX;  // host copy, N x L, filled on demand by dca_host_msa()
    double* dLastScores = nullptr;   // most recent score vector (pair order), for dca_scores_order
    int nLastScores = 0;

    // weights
    bool have_weights = false;
    bool have_counts = false;
    uint32_t* dCounts = nullptr;
    double* dWd = nullptr;     // N doubles
    double meff = 0.0;
    unsigned long long weightsWork[2] = {0, 0};     // last weights launch: wave x 32-site groups compared / without the early exit
    int weightsPlanes = 0;                          // ... bit planes per group (5: q <= 32, 3: q <= 8)

    // scratch scalars: device slots + pinned host mirror
    double* dScal = nullptr;
    double* hScal = nullptr;

    PlmEngineBase* plm = nullptr;
    MfEngine* mf = nullptr;

    // native communicator (comm_rccl.cpp): an RCCL communicator whose collectives run on `stream`
    void* comm = nullptr;
    std::atomic<bool> comm_aborted{false};      // dca_comm_abort ran (from a watchdog thread): the communicator is already released
    int comm_rank = 0, comm_world = 0;
    void* commStage = nullptr;        // pieces received by the direct-exchange reduce-scatter ((world - 1) slices)
    size_t commStageBytes = 0;

    bool profiling = false;
    std::string profile_only;      // non-empty: only the stage of this name is clocked (bench.py's timed region: the roofline kernel alone)
    std::map<std::string, KernelClock> clocks;
};

// RAII-less helpers for bracketing a kernel with events on ctx->stream
struct ScopedKernelClock {
    dca_ctx* ctx;
    KernelClock* kc = nullptr;
    hipEvent_t a = nullptr, b = nullptr;
    ScopedKernelClock(dca_ctx* c, const char* tag) : ctx(c) {
        if (!c->profiling) return;
        if (!c->profile_only.empty() && c->profile_only != tag) return;
        kc = &c->clocks[tag];
        if (hipEventCreate(&a) != hipSuccess || hipEventCreate(&b) != hipSuccess) { kc = nullptr; return; }
        hipEventRecord(a, c->stream);
    }
    ~ScopedKernelClock() {
        if (!kc) return;
        hipEventRecord(b, ctx->stream);
        kc->pending.emplace_back(a, b);
        kc->launches += 1;
    }
};
void dca_flush_clocks(dca_ctx* ctx);

// host copy of the alignment (N x L), fetched from the device the first time it is needed
const uint8_t* dca_host_msa(dca_ctx* ctx);
// keeps a device copy of a score vector for dca_scores_order
int dca_remember_scores(dca_ctx* ctx, const double* dScores, int n);
// rank.hip: indices of a device score vector in descending order, ties by ascending index (stable)
int dca_scores_order_device(dca_ctx* ctx, const double* dScores, int n, int32_t* order_out /* host */);

// ---- weights.hip
// part / parts: this context counts the tile pairs  t % parts == part  of the upper triangle (1 / parts of the N^2 L / 2
// comparisons, evenly spread); finish = false leaves the partial counts in ctx->dCounts for the caller to sum
int dca_weights_compute(dca_ctx* ctx, double seqid, int compare_precision, int part = 0, int parts = 1, bool finish = true);
int dca_weights_finish(dca_ctx* ctx);     // w = 1 / count, Meff from ctx->dCounts

// ---- comm_rccl.cpp
int dca_comm_unique_id_impl(const char* rccl_path, void* id128);
int dca_comm_init_impl(dca_ctx* ctx, const char* rccl_path, const void* id128, int world, int rank);
void dca_comm_destroy_impl(dca_ctx* ctx);
int dca_comm_abort_impl(dca_ctx* ctx);
int dca_comm_info_impl(dca_ctx* ctx, int* world, int* rank);
int dca_comm_p2p_begin(dca_ctx* ctx);
int dca_comm_p2p_send(dca_ctx* ctx, const void* buf, size_t count, int dtype, int peer);
int dca_comm_p2p_recv(dca_ctx* ctx, void* buf, size_t count, int dtype, int peer);
int dca_comm_p2p_end(dca_ctx* ctx);
int dca_comm_native(dca_ctx* ctx, int op, void* buf, size_t count, int dtype, bool direct = false);   // direct: grouped send / recv + local sum
int dca_comm_native_reduce(dca_ctx* ctx, void* vec, size_t count, int dtype, double* scalar_dev);
int dca_comm_native_sum_u32(dca_ctx* ctx, uint32_t* buf, size_t count);

// ---- reductions.hip : deterministic device reductions
// out[slot] = sum(partials[0..n)) ; single block, fixed tree
int dca_reduce_partials(dca_ctx* ctx, const double* dPartials, int n, double* dOut);
int dca_sum_doubles(dca_ctx* ctx, const double* dVals, int n, double* dOut);  // two-stage

// ---- plm engine
struct PlmEngineBase {
    virtual ~PlmEngineBase() {}
    virtual int configure(double lambda_h, double lambda_J, int carry_mode, int chunk, int warmup,
                          int halo, int add_reg) = 0;
    virtual int configure_strips(double lambda_h, double lambda_J, int carry_mode, int chunk, int warmup) = 0;   // column-strip decomposition over ctx->comm
    virtual int init_x() = 0;
    virtual int set_x(const void* x, int dtype) = 0;
    virtual int get_x(void* x, int dtype) = 0;
    virtual int get_g(void* g, int dtype) = 0;
    virtual int gradient(double* fx_out) = 0;
    virtual int lbfgs_begin(int max_iterations, int verbose) = 0;
    virtual int lbfgs_iterate(int iterations, dca_plm_stats* st) = 0;
    virtual void lbfgs_end() = 0;       // abandons the optimisation in progress (x, g stay as they are)
    virtual int scores(int apc, double* out) = 0;
    virtual int di_scores(const double* reg_fi, int apc, double* out) = 0;
    virtual int pair_couplings(const int* pairs, int npairs, int shift, double* out) = 0;
    virtual int set_vector_sharding(int rank, int world, dca_comm_hook hook, void* user) = 0;
    dca_reduce_hook hook = nullptr;
    void* hook_user = nullptr;
    virtual void weights_changed() = 0;            // dca_compute_weights* / dca_set_weights ran: configure again, exchange scheme kept
    virtual int set_native_comm(int mode) = 0;     // 0 off, 1 all-reduce of g and fx, 2 sharded optimiser vectors, 3 the same by direct exchange
    virtual bool configured_for_comm() const = 0;  // configured: its slices follow a communicator (an unconfigured engine re-cuts them in configure)
    int native_mode = 0;                           // ... through ctx->comm (RCCL) on the context's stream
};
PlmEngineBase* dca_make_plm_engine(dca_ctx* ctx);

// ---- scoring.hip
// FN of (q-1)x(q-1) blocks.  src_kind 0: packed plm vector of element type `dtype`
// (DCA_F32/DCA_F64); 1: dense n x n double couplings with leading dimension ld.
int dca_fn_scores(dca_ctx* ctx, const void* src, int src_kind, int dtype, int L, int q, int ld,
                  int apc, double* dScoresOut /* device, pairs */);

int dca_di_scores(dca_ctx* ctx, const void* src, int src_kind, int dtype, const double* dRegFi, int L, int q, int ld,
                  int apc, double* dScoresOut /* device, pairs */);

int dca_pair_blocks(dca_ctx* ctx, const void* src, int src_kind, int dtype, int L, int q, int ld, const int* pairs, int npairs,
                    int shift, double* out /* host */);

int dca_di_from_arrays_impl(dca_ctx* ctx, const double* couplings, int layout, const double* reg_fi, int L, int q,
                            double* fields_out, double* di_out, const double* fields_in = nullptr);

// ---- mf engine
struct MfEngine;
MfEngine* dca_make_mf_engine(dca_ctx* ctx);
void dca_free_mf_engine(MfEngine*);
int dca_mf_engine_site_freqs(MfEngine*, double* fi_out);
int dca_mf_engine_pair_freqs(MfEngine*, double* fij_out);
int dca_mf_engine_corr(MfEngine*, double theta, double* corr_out);
int dca_mf_engine_couplings(MfEngine*, double* out);
int dca_mf_engine_scores(MfEngine*, int apc, double* out);
int dca_mf_engine_di(MfEngine*, int apc, double* out);
int dca_mf_engine_fields(MfEngine*, double* out);
void dca_mf_engine_set_hook(MfEngine*, dca_reduce_hook hook, void* user);
void dca_mf_engine_set_native(MfEngine*, bool on);
int dca_mf_engine_set_row_window(MfEngine*, int first, int count);   // count < 0: all rows
void dca_mf_engine_invalidate(MfEngine*);      // weights changed: counts, frequencies, C and J are recomputed on demand
int dca_mf_engine_pair_couplings(MfEngine*, const int* pairs, int npairs, int shift, double* out);

// ---- cholinv.hip : scale * inverse of an SPD matrix on the device (f64 MFMA)
// dA: n x n row-major (ld = n), n multiple of 64; destroyed (holds the triangular factor's inverse afterwards).
// dWork: >= 2*n*n doubles; *result points into it (its second half): scale * inv(A), full and bit-symmetric.
// info_out: 0 ok, >0 first non-positive pivot (1-based).
int dca_spd_inverse_device(dca_ctx* ctx, double* dA, int n, double* dWork, int* info_out, double scale, double** result);

// ---- host_io.cpp
int dca_read_msa_impl(const char* path, int biomolecule, int L, uint8_t* out, int capacity, int* raw_count);
int dca_read_msa_owned(const char* path, int biomolecule, int L, uint8_t** rows /* malloc'd, caller frees */, int* raw_count);   // one pass, no capacity
int dca_count_msa_lines_impl(const char* path);
