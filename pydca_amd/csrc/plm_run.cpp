// dca_plm_run: the one-call plmDCA entry of SURVEY section 8 b1 -- the reference's single call plmdcaBackend(...)
// (plmdcaBackend.cpp:151-201) with what it drops (status / iterations / evaluations / timers) and a device list.
//
// One rank per device as one HOST THREAD each of the calling process: a C / C++ / Fortran host gets several GPUs without a helper
// executable and without fork()ing a process that has HIP state (the Python classes use one process per GPU, pydca_amd/multi_gpu.py;
// RCCL communicators are per device and thread-safe, one per rank here).  What every rank runs is what multi_gpu.plm_rank runs:
// sequence weights with the comparisons divided over the ranks (ONE all-reduce of the N counts), the initial point from the
// whole alignment, then the optimisation under the column-strip decomposition (dca_plm_configure_strips: 1 / world of the bytes of the
// sequence-sharded schemes on the wires, float64 gradient bit-identical to one GPU's) -- deterministic, so the same call
// gives the same bytes from run to run.  A rank that fails aborts every communicator (dca_comm_abort), so the others come back
// instead of waiting for it.
#include <atomic>
#include <condition_variable>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "dca_internal.h"

int dca_read_msa_owned(const char* path, int biomolecule, int L, uint8_t** rows, int* raw_count);

namespace {

struct Shared {
    const dca_plm_args* a;
    const uint8_t* X; int N, L, q;
    int world;
    unsigned char ids[2][128];
    std::mutex mu;
    std::condition_variable cv;
    int arrived = 0, generation = 0;
    std::atomic<int> failed{0};
    int rc = DCA_OK;
    std::string message;
    std::vector<dca_ctx*> comms;           // contexts with a live communicator (what a failing rank aborts)
    void* x_out; int x_dtype;
    dca_plm_stats stats;
};

// a failure before the first collective must be seen by everybody BEFORE anyone enters one: host barrier with a verdict
bool meet(Shared& s)
{
    std::unique_lock<std::mutex> lk(s.mu);
    const int gen = s.generation;
    if (++s.arrived == s.world) { s.arrived = 0; ++s.generation; s.cv.notify_all(); }
    else s.cv.wait(lk, [&] { return s.generation != gen; });
    return s.failed.load() == 0;
}

void fail(Shared& s, int rank, int rc)
{
    std::vector<dca_ctx*> live;
    {
        std::lock_guard<std::mutex> lk(s.mu);
        if (!s.failed.exchange(1)) {
            s.rc = rc;
            s.message = "rank " + std::to_string(rank) + ": " + dca_last_error();
        }
        live = s.comms;
    }
    for (dca_ctx* c : live) dca_comm_abort(c);       // the peers' pending collectives fail; they end in their own fail()
}

void track(Shared& s, dca_ctx* c) { std::lock_guard<std::mutex> lk(s.mu); s.comms.push_back(c); }
void untrack(Shared& s, dca_ctx* c)
{
    std::lock_guard<std::mutex> lk(s.mu);
    for (size_t i = 0; i < s.comms.size(); ++i) if (s.comms[i] == c) { s.comms.erase(s.comms.begin() + i); break; }
}

void rank_main(Shared& s, int rank)
{
    const dca_plm_args& a = *s.a;
    const int dev = a.devices[rank], prec = a.precision;
    dca_ctx *full = nullptr, *run = nullptr;
    std::vector<uint32_t> counts((size_t)s.N);
    const size_t P = dca_plm_num_params(s.L, s.q);
    std::vector<unsigned char> x0;
    int rc = DCA_OK;
#define STEP(expr) if (rc == DCA_OK && s.failed.load() == 0) rc = (expr)
    // ---- nothing here needs a peer
    STEP(dca_create(&full, dev, prec));
    STEP(dca_set_msa(full, s.X, s.N, s.L, s.q));
    STEP(dca_create(&run, dev, prec));
    STEP(dca_set_msa(run, s.X, s.N, s.L, s.q));
    if (rc != DCA_OK) fail(s, rank, rc);
    if (meet(s)) {
        // ---- collectives from here on
        STEP(dca_comm_init(full, a.rccl_path, s.ids[0], s.world, rank));
        if (rc == DCA_OK && full->comm) track(s, full);
        STEP(dca_compute_weights_sharded(full, (double)a.seqid, prec));
        STEP(dca_get_weight_counts(full, counts.data()));
        STEP(dca_plm_configure(full, (double)a.lambda_h, (double)a.lambda_J, DCA_CARRY_CHUNKED, 0, 0, 0, 1));
        STEP(dca_plm_init_x(full));
        if (rc == DCA_OK) x0.resize(P * (prec == DCA_F64 ? 8 : 4));
        STEP(dca_plm_get_x(full, x0.data(), prec));
        STEP(dca_plm_release(full));            // only the initial point was wanted: its N x L q tables do not stay beside the strip's
        STEP(dca_set_weight_counts(run, counts.data()));
        STEP(dca_comm_init(run, a.rccl_path, s.ids[1], s.world, rank));
        if (rc == DCA_OK && run->comm) track(s, run);
        STEP(dca_plm_configure_strips(run, (double)a.lambda_h, (double)a.lambda_J, DCA_CARRY_CHUNKED, 0, 0));
        STEP(dca_plm_set_x(run, x0.data(), prec));
        STEP(dca_plm_lbfgs_begin(run, a.max_iterations, (a.verbose && rank == 0) ? 1 : 0));
        dca_plm_stats st;
        memset(&st, 0, sizeof(st));
        STEP(dca_plm_lbfgs_iterate(run, a.max_iterations ? a.max_iterations : 1 << 30, &st));
        // x: collective under the column strips (every rank gathers the owned ranges); rank 0 hands it out
        std::vector<unsigned char> xr;
        if (rc == DCA_OK && rank != 0) xr.resize(P * (s.x_dtype == DCA_F64 ? 8 : 4));
        STEP(dca_plm_get_x(run, rank == 0 ? s.x_out : xr.data(), s.x_dtype));
        if (rc == DCA_OK && rank == 0) s.stats = st;
        if (rc != DCA_OK && s.failed.load() == 0) fail(s, rank, rc);
        else if (rc != DCA_OK) fail(s, rank, rc);       // records nothing new, aborts what is still up
    }
#undef STEP
    // an aborted communicator is only forgotten by the destroy; a healthy one meets its peers there
    if (run) { untrack(s, run); dca_destroy(run); }
    if (full) { untrack(s, full); dca_destroy(full); }
}

int run_single(const dca_plm_args& a, const uint8_t* X, int N, int L, int q, void* x_out, int x_dtype, dca_plm_stats* stats)
{
    dca_ctx* ctx = nullptr;
    DCA_TRY(dca_create(&ctx, a.num_devices > 0 ? a.devices[0] : 0, a.precision));
    dca_plm_stats st;
    memset(&st, 0, sizeof(st));
    int rc = dca_set_msa(ctx, X, N, L, q);
    if (rc == DCA_OK) rc = dca_compute_weights(ctx, (double)a.seqid, a.precision);
    if (rc == DCA_OK) rc = dca_plm_configure(ctx, (double)a.lambda_h, (double)a.lambda_J, DCA_CARRY_CHUNKED, 0, 0, 0, 1);
    if (rc == DCA_OK) rc = dca_plm_init_x(ctx);
    if (rc == DCA_OK) rc = dca_plm_lbfgs_begin(ctx, a.max_iterations, a.verbose ? 1 : 0);
    if (rc == DCA_OK) rc = dca_plm_lbfgs_iterate(ctx, a.max_iterations ? a.max_iterations : 1 << 30, &st);
    if (rc == DCA_OK) rc = dca_plm_get_x(ctx, x_out, x_dtype);
    dca_destroy(ctx);
    if (rc == DCA_OK && stats) *stats = st;
    return rc;
}

}  // namespace

extern "C" int dca_plm_run(const dca_plm_args* a, void* x_out, int x_dtype, dca_plm_stats* stats)
{
    if (!a || !x_out || (x_dtype != DCA_F32 && x_dtype != DCA_F64) || (a->precision != DCA_F32 && a->precision != DCA_F64) ||
        (a->biomolecule != DCA_BIOMOLECULE_PROTEIN && a->biomolecule != DCA_BIOMOLECULE_RNA) || (!a->msa_file && !a->msa) ||
        a->seqs_len <= 1 || a->num_devices < 0 || (a->num_devices > 0 && !a->devices) || a->max_iterations < 0 ||
        !(a->seqid > 0.f && a->seqid <= 1.f) || a->lambda_h < 0.f || a->lambda_J < 0.f) {
        dca_set_error("dca_plm_run: bad arguments");
        return DCA_ERR_ARG;
    }
    if (a->exchange_scheme != 0 && a->exchange_scheme != 4) {
        dca_set_error("dca_plm_run runs the column-strip decomposition (exchange_scheme 0 or 4); the sequence-sharded schemes are reached through the stage API");
        return DCA_ERR_ARG;
    }
    const int q = a->biomolecule == DCA_BIOMOLECULE_PROTEIN ? 21 : 5, L = a->seqs_len;
    uint8_t* owned = nullptr;
    const uint8_t* X = a->msa;
    int N = a->num_seqs;
    if (a->msa_file) {
        int raw = 0;
        N = dca_read_msa_owned(a->msa_file, a->biomolecule, L, &owned, &raw);       // first-occurrence dedup as PlmDCA::readSequencesFromFile
        if (N < 0) { free(owned); return N; }
        if (N == 0) { free(owned); dca_set_error("no sequences in %s", a->msa_file); return DCA_ERR_IO; }
        X = owned;
    } else if (N <= 0) { dca_set_error("dca_plm_run: num_seqs"); return DCA_ERR_ARG; }
    int rc;
    if (a->num_devices <= 1) rc = run_single(*a, X, N, L, q, x_out, x_dtype, stats);
    else if (a->num_devices > L) { dca_set_error("dca_plm_run: more devices than sites"); rc = DCA_ERR_ARG; }
    else {
        Shared s;
        s.a = a; s.X = X; s.N = N; s.L = L; s.q = q; s.world = a->num_devices; s.x_out = x_out; s.x_dtype = x_dtype;
        memset(&s.stats, 0, sizeof(s.stats));
        rc = dca_comm_unique_id(a->rccl_path, s.ids[0]);
        if (rc == DCA_OK) rc = dca_comm_unique_id(a->rccl_path, s.ids[1]);
        if (rc == DCA_OK) {
            std::vector<std::thread> th;
            for (int r = 1; r < s.world; ++r) th.emplace_back(rank_main, std::ref(s), r);
            rank_main(s, 0);
            for (auto& t : th) t.join();
            if (s.failed.load()) { dca_set_error("%s", s.message.c_str()); rc = s.rc != DCA_OK ? s.rc : DCA_ERR_HIP; }
            else if (stats) *stats = s.stats;
        }
    }
    free(owned);
    return rc;
}
