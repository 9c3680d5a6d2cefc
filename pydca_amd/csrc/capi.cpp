// extern "C" surface of libdca_hip.so (see include/dca_hip.h).
#include <cstdlib>

#include "dca_internal.h"

static thread_local char g_err[1024] = "";

void dca_set_error(const char* fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

void dca_flush_clocks(dca_ctx* ctx)
{
    for (auto& kv : ctx->clocks) {
        for (auto& pr : kv.second.pending) {
            float ms = 0.f;
            hipEventSynchronize(pr.second);
            if (hipEventElapsedTime(&ms, pr.first, pr.second) == hipSuccess) kv.second.ms += ms;
            hipEventDestroy(pr.first);
            hipEventDestroy(pr.second);
        }
        kv.second.pending.clear();
    }
}

// ---- device block cache (declared in dca_internal.h)
#include <mutex>
#include <unordered_map>
namespace {
struct DevBlock { void* p; size_t bytes; int device; };
struct DevPool {
    std::mutex mu;
    std::unordered_map<void*, DevBlock> live;     // blocks handed out that are eligible for caching
    std::vector<DevBlock> cached;
    size_t cachedBytes = 0;
    size_t maxBytes = getenv("DCA_POOL_MAX_BYTES") ? strtoull(getenv("DCA_POOL_MAX_BYTES"), nullptr, 10) : ((size_t)64 << 30);
    void release_all() {          // caller holds mu
        for (auto& b : cached) { hipSetDevice(b.device); hipFree(b.p); }
        cached.clear(); cachedBytes = 0;
    }
};
DevPool& pool() { static DevPool* p = new DevPool(); return *p; }   // never destroyed: the HIP runtime may be gone at exit
constexpr size_t kPoolMinBytes = (size_t)1 << 20;
}  // namespace

hipError_t dca_dev_malloc(void** out, size_t bytes, bool zero_recycled)
{
    *out = nullptr;
    if (bytes < kPoolMinBytes) return hipMalloc(out, bytes);
    int dev = 0;
    hipGetDevice(&dev);
    DevPool& P = pool();
    void* hit = nullptr;
    size_t hitBytes = 0;
    {
        std::lock_guard<std::mutex> lk(P.mu);
        int best = -1;
        for (int i = 0; i < (int)P.cached.size(); ++i) {
            const DevBlock& b = P.cached[i];
            if (b.device != dev || b.bytes < bytes || b.bytes > bytes + bytes / 4) continue;
            if (best < 0 || b.bytes < P.cached[best].bytes) best = i;
        }
        if (best >= 0) {
            hit = P.cached[best].p; hitBytes = P.cached[best].bytes;
            P.cachedBytes -= hitBytes;
            P.cached.erase(P.cached.begin() + best);
            P.live[hit] = DevBlock{hit, hitBytes, dev};
        }
    }
    if (hit) {
        // a fresh hipMalloc block reads as zeros; keep that for recycled ones (the copy engine fills > 3 TB/s)
        if (zero_recycled) {
            hipError_t e = hipMemsetAsync(hit, 0, bytes, nullptr);
            if (e == hipSuccess) e = hipStreamSynchronize(nullptr);
            if (e != hipSuccess) return e;
        }
        *out = hit;
        return hipSuccess;
    }
    hipError_t e = hipMalloc(out, bytes);
    if (e != hipSuccess) {        // out of memory: give the cache back and try once more
        (void)hipGetLastError();
        { std::lock_guard<std::mutex> lk(P.mu); P.release_all(); }
        e = hipMalloc(out, bytes);
        if (e != hipSuccess) return e;
    }
    std::lock_guard<std::mutex> lk(P.mu);
    P.live[*out] = DevBlock{*out, bytes, dev};
    return hipSuccess;
}

hipError_t dca_dev_free(void* p)
{
    if (!p) return hipSuccess;
    DevPool& P = pool();
    DevBlock b{nullptr, 0, 0};
    {
        std::lock_guard<std::mutex> lk(P.mu);
        auto it = P.live.find(p);
        if (it != P.live.end()) { b = it->second; P.live.erase(it); }
    }
    if (!b.p) return hipFree(p);
    // same guarantee as hipFree: nothing on the device still uses the block when this returns
    int cur = 0;
    hipGetDevice(&cur);
    if (cur != b.device) hipSetDevice(b.device);
    hipError_t e = hipDeviceSynchronize();
    if (cur != b.device) hipSetDevice(cur);
    std::lock_guard<std::mutex> lk(P.mu);
    if (e != hipSuccess || P.cachedBytes + b.bytes > P.maxBytes) {
        hipSetDevice(b.device); hipError_t f = hipFree(b.p); hipSetDevice(cur);
        return f;
    }
    P.cached.push_back(b);
    P.cachedBytes += b.bytes;
    return hipSuccess;
}

#define CHECK_CTX(ctx)                                             \
    do {                                                           \
        if (!(ctx)) { dca_set_error("null context"); return DCA_ERR_ARG; } \
        hipError_t _e = hipSetDevice((ctx)->device);               \
        if (_e != hipSuccess) { dca_set_error("hipSetDevice: %s", hipGetErrorString(_e)); return DCA_ERR_HIP; } \
    } while (0)

namespace {
// dst[n][0..Ls) = src[n][0..L), zero padded; *maxCode = the largest code met (the range check of dca_set_msa, on the device: a
// pass over the 25 MB of config D costs the host 2.5 ms)
__global__ void pad_rows_kernel(const uint8_t* __restrict__ src, uint8_t* __restrict__ dst, int N, int L, int Ls, unsigned* __restrict__ maxCode)
{
    const size_t t = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    unsigned v = 0;
    if (t < (size_t)N * Ls) {
        const size_t n = t / Ls;
        const int c = (int)(t % Ls);
        v = c < L ? src[n * L + c] : 0u;
        dst[t] = (uint8_t)v;
    }
    // wave maximum, one atomic per wave that holds a code above the smallest alphabet
    for (int off = 32; off > 0; off >>= 1) v = max(v, (unsigned)__shfl_xor((int)v, off));
    // (read first: after the first few waves nobody has anything larger to report -- 800 000 atomics on one word cost config E 8 ms)
    if ((threadIdx.x & 63) == 0 && v > __hip_atomic_load(maxCode, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(maxCode, v);
}
__global__ void unpad_rows_kernel(const uint8_t* __restrict__ src, uint8_t* __restrict__ dst, int N, int L, int Ls)
{
    const size_t t = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (t >= (size_t)N * L) return;
    dst[t] = src[(t / L) * Ls + t % L];
}
}  // namespace

extern "C" {

const char* dca_last_error(void) { return g_err; }
const char* dca_version(void) { return "pydca_amd libdca_hip 0.1 (gfx950)"; }

size_t dca_release_cached_memory(void)
{
    DevPool& P = pool();
    std::lock_guard<std::mutex> lk(P.mu);
    const size_t n = P.cachedBytes;
    int cur = 0;
    hipGetDevice(&cur);
    P.release_all();
    hipSetDevice(cur);
    return n;
}

int dca_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

size_t dca_plm_num_params(int L, int q)
{
    return (size_t)L * q + (size_t)L * (L - 1) / 2 * (size_t)q * q;
}

int dca_read_msa(const char* path, int biomolecule, int L, uint8_t* out, int capacity, int* raw_count)
{
    return dca_read_msa_impl(path, biomolecule, L, out, capacity, raw_count);
}

int dca_count_msa_lines(const char* path) { return dca_count_msa_lines_impl(path); }
int dca_read_msa_alloc(const char* path, int biomolecule, int L, uint8_t** rows, int* raw_count)
{
    if (!rows) { dca_set_error("dca_read_msa_alloc: bad arguments"); return DCA_ERR_ARG; }
    return dca_read_msa_owned(path, biomolecule, L, rows, raw_count);
}

int dca_create(dca_ctx** out, int device, int precision)
{
    if (!out || (precision != DCA_F32 && precision != DCA_F64)) { dca_set_error("dca_create: bad arguments"); return DCA_ERR_ARG; }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) {
        dca_set_error("no HIP device visible: libdca_hip has no CPU fallback");
        return DCA_ERR_NO_DEVICE;
    }
    if (device < 0 || device >= ndev) { dca_set_error("device %d out of range (%d visible)", device, ndev); return DCA_ERR_ARG; }
    HIP_TRY(hipSetDevice(device));
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, device));
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
        dca_set_error("device %d is %s; this library is built for gfx950 (MI355X) only", device, prop.gcnArchName);
        return DCA_ERR_NO_DEVICE;
    }
    dca_ctx* ctx = new dca_ctx();
    ctx->device = device;
    ctx->precision = precision;
    hipError_t e = hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking);
    if (e == hipSuccess) e = dca_dev_malloc(reinterpret_cast<void**>(&ctx->dScal), 64 * sizeof(double));
    if (e == hipSuccess) e = hipHostMalloc(reinterpret_cast<void**>(&ctx->hScal), 64 * sizeof(double), hipHostMallocDefault);
    if (e != hipSuccess) {                       // nothing half-built is handed out or left behind
        dca_set_error("dca_create: %s", hipGetErrorString(e));
        dca_destroy(ctx);
        return DCA_ERR_HIP;
    }
    *out = ctx;
    return DCA_OK;
}

static void free_msa(dca_ctx* ctx)
{
    delete ctx->plm; ctx->plm = nullptr;
    if (ctx->mf) { dca_free_mf_engine(ctx->mf); ctx->mf = nullptr; }
    dca_dev_free(ctx->dX); ctx->dX = nullptr;
    dca_dev_free(ctx->dCounts); ctx->dCounts = nullptr;
    dca_dev_free(ctx->dWd); ctx->dWd = nullptr;
    dca_dev_free(ctx->dLastScores); ctx->dLastScores = nullptr; ctx->nLastScores = 0;
    ctx->hX.clear();
    ctx->have_weights = ctx->have_counts = false;
}

}  // extern "C" (internal C++ helpers follow)

const uint8_t* dca_host_msa(dca_ctx* ctx)
{
    if (ctx->hX.empty() && ctx->dX) {
        ctx->hX.resize((size_t)ctx->N * ctx->L);
        uint8_t* dTmp = nullptr;
        hipError_t e = dca_dev_malloc(reinterpret_cast<void**>(&dTmp), ctx->hX.size());
        if (e == hipSuccess) {
            hipLaunchKernelGGL(unpad_rows_kernel, dim3((unsigned)((ctx->hX.size() + 255) / 256)), dim3(256), 0, ctx->stream,
                               ctx->dX, dTmp, ctx->N, ctx->L, ctx->Ls);
            e = hipStreamSynchronize(ctx->stream);
            if (e == hipSuccess) e = hipMemcpy(ctx->hX.data(), dTmp, ctx->hX.size(), hipMemcpyDeviceToHost);
            dca_dev_free(dTmp);
        }
        if (e != hipSuccess) {
            ctx->hX.clear();
            dca_set_error("copying the alignment back to the host failed: %s", hipGetErrorString(e));
            return nullptr;
        }
    }
    return ctx->hX.empty() ? nullptr : ctx->hX.data();
}

int dca_remember_scores(dca_ctx* ctx, const double* dScores, int n)
{
    if (ctx->nLastScores != n) {
        dca_dev_free(ctx->dLastScores); ctx->dLastScores = nullptr; ctx->nLastScores = 0;
        HIP_TRY(dca_dev_malloc(reinterpret_cast<void**>(&ctx->dLastScores), (size_t)n * sizeof(double)));
        ctx->nLastScores = n;
    }
    HIP_TRY(hipMemcpyAsync(ctx->dLastScores, dScores, (size_t)n * sizeof(double), hipMemcpyDeviceToDevice, ctx->stream));
    return DCA_OK;
}

extern "C" {

void dca_destroy(dca_ctx* ctx)
{
    if (!ctx) return;
    hipSetDevice(ctx->device);
    if (ctx->stream) hipStreamSynchronize(ctx->stream);
    dca_comm_destroy_impl(ctx);
    dca_flush_clocks(ctx);
    free_msa(ctx);
    dca_dev_free(ctx->dScal);
    if (ctx->hScal) hipHostFree(ctx->hScal);
    if (ctx->stream) hipStreamDestroy(ctx->stream);
    delete ctx;
}

int dca_set_msa(dca_ctx* ctx, const uint8_t* X, int N, int L, int q)
{
    CHECK_CTX(ctx);
    if (!X || N <= 0 || L <= 1 || q < 2 || q > 32) { dca_set_error("dca_set_msa: bad arguments"); return DCA_ERR_ARG; }
    free_msa(ctx);
    ctx->N = ctx->L = ctx->q = ctx->Ls = 0;                // the context holds no alignment until everything below succeeded
    const int Ls = (int)round_up((size_t)L, 128);
    ctx->hX.clear();                                       // host copy is made on demand (dca_host_msa)
    // one contiguous copy + a repack kernel (a pitched hipMemcpy2D of narrow rows takes seconds)
    uint8_t* dTmp = nullptr;
    unsigned* dMax = nullptr;
    unsigned maxCode = 0;
    hipError_t e = dca_dev_malloc(reinterpret_cast<void**>(&ctx->dX), (size_t)N * Ls);
    if (e == hipSuccess) e = dca_dev_malloc(reinterpret_cast<void**>(&dTmp), (size_t)N * L);
    if (e == hipSuccess) e = dca_dev_malloc(reinterpret_cast<void**>(&dMax), sizeof(unsigned));
    if (e == hipSuccess) e = hipMemsetAsync(dMax, 0, sizeof(unsigned), ctx->stream);
    if (e == hipSuccess) e = hipMemcpy(dTmp, X, (size_t)N * L, hipMemcpyHostToDevice);
    if (e == hipSuccess) {
        const size_t total = (size_t)N * Ls;
        hipLaunchKernelGGL(pad_rows_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, ctx->stream, dTmp, ctx->dX, N, L, Ls, dMax);
        e = hipMemcpyAsync(&maxCode, dMax, sizeof(unsigned), hipMemcpyDeviceToHost, ctx->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    }
    dca_dev_free(dTmp);
    dca_dev_free(dMax);
    if (e == hipSuccess && maxCode >= (unsigned)q) {      // the slow search only runs on failure
        size_t k = 0;
        while (X[k] < q) ++k;
        free_msa(ctx);
        dca_set_error("dca_set_msa: code %d >= q at element %zu", (int)X[k], k);
        return DCA_ERR_ARG;
    }
    if (e == hipSuccess) e = dca_dev_malloc(reinterpret_cast<void**>(&ctx->dCounts), (size_t)N * sizeof(uint32_t));
    if (e == hipSuccess) e = dca_dev_malloc(reinterpret_cast<void**>(&ctx->dWd), (size_t)N * sizeof(double));
    if (e != hipSuccess) {
        free_msa(ctx);
        dca_set_error("uploading the alignment: %s", hipGetErrorString(e));
        return DCA_ERR_HIP;
    }
    ctx->N = N; ctx->L = L; ctx->q = q; ctx->Ls = Ls;
    return DCA_OK;
}

// Everything the engines derived from the weights (the plmDCA engine's copy, frequencies, counts, correlation matrix,
// couplings) is dropped when the weights change: the next call recomputes it instead of answering for the old weights.
// The engines themselves stay (with their reduce / comm hooks, native-comm mode and vector sharding): a sharded context that
// is re-weighted must not fall back to unreduced local sums.  The plmDCA engine has to be configured again.
static void weights_changed(dca_ctx* ctx)
{
    if (ctx->plm) ctx->plm->weights_changed();
    if (ctx->mf) dca_mf_engine_invalidate(ctx->mf);
}

int dca_compute_weights(dca_ctx* ctx, double seqid, int compare_precision)
{
    CHECK_CTX(ctx);
    if (!ctx->dX) { dca_set_error("dca_set_msa first"); return DCA_ERR_STATE; }
    if (compare_precision != DCA_F32 && compare_precision != DCA_F64) return DCA_ERR_ARG;
    weights_changed(ctx);
    return dca_weights_compute(ctx, seqid, compare_precision);
}

int dca_weights_partial_counts(dca_ctx* ctx, double seqid, int compare_precision, int part, int parts, uint32_t* counts_out)
{
    CHECK_CTX(ctx);
    if (!ctx->dX) { dca_set_error("dca_set_msa first"); return DCA_ERR_STATE; }
    if (compare_precision != DCA_F32 && compare_precision != DCA_F64) return DCA_ERR_ARG;
    weights_changed(ctx);
    DCA_TRY(dca_weights_compute(ctx, seqid, compare_precision, part, parts, false));
    if (counts_out) HIP_TRY(hipMemcpy(counts_out, ctx->dCounts, (size_t)ctx->N * sizeof(uint32_t), hipMemcpyDeviceToHost));
    return DCA_OK;
}

int dca_set_weight_counts(dca_ctx* ctx, const uint32_t* counts)
{
    CHECK_CTX(ctx);
    if (!ctx->dX || !counts) { dca_set_error("dca_set_msa first"); return DCA_ERR_STATE; }
    for (int n = 0; n < ctx->N; ++n)
        if (counts[n] == 0) { dca_set_error("dca_set_weight_counts: count of sequence %d is zero (every sequence counts itself)", n); return DCA_ERR_ARG; }
    weights_changed(ctx);
    HIP_TRY(hipMemcpy(ctx->dCounts, counts, (size_t)ctx->N * sizeof(uint32_t), hipMemcpyHostToDevice));
    return dca_weights_finish(ctx);
}

int dca_compute_weights_sharded(dca_ctx* ctx, double seqid, int compare_precision)
{
    CHECK_CTX(ctx);
    if (!ctx->dX) { dca_set_error("dca_set_msa first"); return DCA_ERR_STATE; }
    if (compare_precision != DCA_F32 && compare_precision != DCA_F64) return DCA_ERR_ARG;
    if (!ctx->comm) { dca_set_error("no communicator: dca_comm_init first"); return DCA_ERR_STATE; }
    weights_changed(ctx);
    DCA_TRY(dca_weights_compute(ctx, seqid, compare_precision, ctx->comm_rank, ctx->comm_world, false));
    DCA_TRY(dca_comm_native_sum_u32(ctx, ctx->dCounts, (size_t)ctx->N));       // integer sums: exact, order-free
    return dca_weights_finish(ctx);
}

int dca_set_weights(dca_ctx* ctx, const double* w)
{
    CHECK_CTX(ctx);
    if (!ctx->dX || !w) { dca_set_error("dca_set_msa first"); return DCA_ERR_STATE; }
    weights_changed(ctx);
    HIP_TRY(hipMemcpy(ctx->dWd, w, (size_t)ctx->N * sizeof(double), hipMemcpyHostToDevice));
    double s = 0;
    for (int n = 0; n < ctx->N; ++n) s += w[n];
    ctx->meff = s;
    ctx->have_weights = true;
    ctx->have_counts = false;
    return DCA_OK;
}

int dca_get_weights(dca_ctx* ctx, double* w_out)
{
    CHECK_CTX(ctx);
    if (!ctx->have_weights) { dca_set_error("weights not available"); return DCA_ERR_STATE; }
    HIP_TRY(hipMemcpy(w_out, ctx->dWd, (size_t)ctx->N * sizeof(double), hipMemcpyDeviceToHost));
    return DCA_OK;
}

int dca_weights_work(dca_ctx* ctx, uint64_t* out3)
{
    if (!ctx || !out3) return DCA_ERR_ARG;
    out3[0] = ctx->weightsWork[0]; out3[1] = ctx->weightsWork[1]; out3[2] = (uint64_t)ctx->weightsPlanes;
    return DCA_OK;
}

int dca_get_weight_counts(dca_ctx* ctx, uint32_t* counts_out)
{
    CHECK_CTX(ctx);
    if (!ctx->have_counts) { dca_set_error("counts not available"); return DCA_ERR_STATE; }
    HIP_TRY(hipMemcpy(counts_out, ctx->dCounts, (size_t)ctx->N * sizeof(uint32_t), hipMemcpyDeviceToHost));
    return DCA_OK;
}

int dca_get_meff(dca_ctx* ctx, double* meff_out)
{
    if (!ctx || !ctx->have_weights) { dca_set_error("weights not available"); return DCA_ERR_STATE; }
    *meff_out = ctx->meff;
    return DCA_OK;
}

// ------------------------------------------------------------------ plmDCA
static int need_plm(dca_ctx* ctx)
{
    if (!ctx->dX) { dca_set_error("dca_set_msa first"); return DCA_ERR_STATE; }
    if (!ctx->plm) ctx->plm = dca_make_plm_engine(ctx);
    return DCA_OK;
}

int dca_plm_configure(dca_ctx* ctx, double lambda_h, double lambda_J, int carry_mode, int chunk, int warmup, int halo, int add_regulariser)
{
    CHECK_CTX(ctx);
    DCA_TRY(need_plm(ctx));
    if (carry_mode < 0 || carry_mode > 2) return DCA_ERR_ARG;
    return ctx->plm->configure(lambda_h, lambda_J, carry_mode, chunk, warmup, halo, add_regulariser);
}
int dca_plm_configure_strips(dca_ctx* ctx, double lambda_h, double lambda_J, int carry_mode, int chunk, int warmup)
{
    CHECK_CTX(ctx);
    DCA_TRY(need_plm(ctx));
    if (carry_mode < 0 || carry_mode > 2) return DCA_ERR_ARG;
    return ctx->plm->configure_strips(lambda_h, lambda_J, carry_mode, chunk, warmup);
}
int dca_plm_release(dca_ctx* ctx)
{
    CHECK_CTX(ctx);
    if (!ctx->plm) return DCA_OK;
    // the engine's kernels may still be running on the context's stream
    if (hipStreamSynchronize(ctx->stream) != hipSuccess) { dca_set_error("dca_plm_release: stream synchronisation failed"); return DCA_ERR_HIP; }
    delete ctx->plm; ctx->plm = nullptr;
    return DCA_OK;
}
int dca_plm_init_x(dca_ctx* ctx) { CHECK_CTX(ctx); DCA_TRY(need_plm(ctx)); return ctx->plm->init_x(); }
int dca_plm_set_x(dca_ctx* ctx, const void* x, int dtype) { CHECK_CTX(ctx); DCA_TRY(need_plm(ctx)); return ctx->plm->set_x(x, dtype); }
int dca_plm_get_x(dca_ctx* ctx, void* x, int dtype) { CHECK_CTX(ctx); DCA_TRY(need_plm(ctx)); return ctx->plm->get_x(x, dtype); }
int dca_plm_get_g(dca_ctx* ctx, void* g, int dtype) { CHECK_CTX(ctx); DCA_TRY(need_plm(ctx)); return ctx->plm->get_g(g, dtype); }
int dca_plm_gradient(dca_ctx* ctx, double* fx_out) { CHECK_CTX(ctx); DCA_TRY(need_plm(ctx)); return ctx->plm->gradient(fx_out); }
int dca_plm_set_vector_sharding(dca_ctx* ctx, int rank, int world, dca_comm_hook hook, void* user)
{
    CHECK_CTX(ctx);
    DCA_TRY(need_plm(ctx));
    return ctx->plm->set_vector_sharding(rank, world, hook, user);
}
int dca_plm_set_reduce_hook(dca_ctx* ctx, dca_reduce_hook hook, void* user)
{
    CHECK_CTX(ctx);
    DCA_TRY(need_plm(ctx));
    if (hook && ctx->plm->native_mode == 4) { dca_set_error("configured for column strips: no reduce hook"); return DCA_ERR_STATE; }
    ctx->plm->hook = hook;
    ctx->plm->hook_user = user;
    if (hook && ctx->plm->native_mode == 1) ctx->plm->native_mode = 0;      // the hook replaces the native all-reduce
    return DCA_OK;
}
int dca_plm_lbfgs_begin(dca_ctx* ctx, int max_iterations, int verbose) { CHECK_CTX(ctx); DCA_TRY(need_plm(ctx)); return ctx->plm->lbfgs_begin(max_iterations, verbose); }
int dca_plm_lbfgs_iterate(dca_ctx* ctx, int iterations, dca_plm_stats* st) { CHECK_CTX(ctx); DCA_TRY(need_plm(ctx)); return ctx->plm->lbfgs_iterate(iterations, st); }
int dca_plm_lbfgs_end(dca_ctx* ctx) { CHECK_CTX(ctx); if (ctx->plm) ctx->plm->lbfgs_end(); return DCA_OK; }
int dca_plm_scores(dca_ctx* ctx, int apc, double* out) { CHECK_CTX(ctx); DCA_TRY(need_plm(ctx)); return ctx->plm->scores(apc, out); }
int dca_plm_di_scores(dca_ctx* ctx, const double* reg_fi, int apc, double* out)
{
    CHECK_CTX(ctx);
    DCA_TRY(need_plm(ctx));
    if (!reg_fi || !out) return DCA_ERR_ARG;
    return ctx->plm->di_scores(reg_fi, apc, out);
}

// ------------------------------------------------------------------ mfDCA
static int need_mf(dca_ctx* ctx)
{
    if (!ctx->dX) { dca_set_error("dca_set_msa first"); return DCA_ERR_STATE; }
    if (!ctx->have_weights) { dca_set_error("weights must be computed or set first"); return DCA_ERR_STATE; }
    if (!ctx->mf) ctx->mf = dca_make_mf_engine(ctx);
    return ctx->mf ? DCA_OK : DCA_ERR_NOMEM;
}
int dca_mf_single_site_freqs(dca_ctx* ctx, double* fi_out) { CHECK_CTX(ctx); DCA_TRY(need_mf(ctx)); return dca_mf_engine_site_freqs(ctx->mf, fi_out); }
int dca_mf_pair_site_freqs(dca_ctx* ctx, double* fij_out) { CHECK_CTX(ctx); DCA_TRY(need_mf(ctx)); return dca_mf_engine_pair_freqs(ctx->mf, fij_out); }
int dca_mf_corr_mat(dca_ctx* ctx, double pseudocount, double* corr_out) { CHECK_CTX(ctx); DCA_TRY(need_mf(ctx)); return dca_mf_engine_corr(ctx->mf, pseudocount, corr_out); }
int dca_mf_couplings(dca_ctx* ctx, double* out) { CHECK_CTX(ctx); DCA_TRY(need_mf(ctx)); return dca_mf_engine_couplings(ctx->mf, out); }
int dca_mf_scores(dca_ctx* ctx, int apc, double* out) { CHECK_CTX(ctx); DCA_TRY(need_mf(ctx)); return dca_mf_engine_scores(ctx->mf, apc, out); }
int dca_plm_pair_couplings(dca_ctx* ctx, const int* pairs, int npairs, int shift, double* out)
{
    CHECK_CTX(ctx);
    DCA_TRY(need_plm(ctx));
    if (npairs < 0 || (npairs > 0 && (!pairs || !out))) return DCA_ERR_ARG;
    return ctx->plm->pair_couplings(pairs, npairs, shift, out);
}
int dca_di_from_arrays(dca_ctx* ctx, const double* couplings, int layout, const double* reg_fi, int L, int q,
                       double* fields_out, double* di_out)
{
    CHECK_CTX(ctx);
    if (!couplings || !reg_fi || (layout != 1 && layout != 2) || (!fields_out && !di_out)) return DCA_ERR_ARG;
    return dca_di_from_arrays_impl(ctx, couplings, layout, reg_fi, L, q, fields_out, di_out);
}
int dca_di_from_fields(dca_ctx* ctx, const double* couplings, int layout, const double* reg_fi, const double* fields_ij,
                       int L, int q, double* di_out)
{
    CHECK_CTX(ctx);
    if (!couplings || !reg_fi || !fields_ij || !di_out || (layout != 1 && layout != 2)) return DCA_ERR_ARG;
    return dca_di_from_arrays_impl(ctx, couplings, layout, reg_fi, L, q, nullptr, di_out, fields_ij);
}
int dca_comm_unique_id(const char* rccl_path, void* id128)
{
    if (!id128) return DCA_ERR_ARG;
    return dca_comm_unique_id_impl(rccl_path, id128);
}
int dca_comm_init(dca_ctx* ctx, const char* rccl_path, const void* id128, int world, int rank)
{
    CHECK_CTX(ctx);
    if (!id128) return DCA_ERR_ARG;
    // slices and reductions configured for the previous communicator's world / rank do not carry over
    // (an engine in the middle of an optimisation refuses: its vector slices are cut for the old world, so the
    // communicator must not change under it -- DCA_ERR_STATE, nothing touched)
    if (ctx->comm) {
        if (ctx->plm && ctx->plm->configured_for_comm()) DCA_TRY(ctx->plm->set_native_comm(0));
        if (ctx->mf) dca_mf_engine_set_native(ctx->mf, false);
    }
    return dca_comm_init_impl(ctx, rccl_path, id128, world, rank);
}
int dca_comm_destroy(dca_ctx* ctx)
{
    CHECK_CTX(ctx);
    // giving the communicator back ends whatever optimisation was being driven over it (a caller that stops calling
    // dca_plm_lbfgs_iterate below its cap never reaches `finished`; without this the communicator could not be released)
    if (ctx->comm && ctx->plm) ctx->plm->lbfgs_end();
    if (ctx->comm && ctx->plm && ctx->plm->configured_for_comm()) DCA_TRY(ctx->plm->set_native_comm(0));
    if (ctx->mf) dca_mf_engine_set_native(ctx->mf, false);
    dca_comm_destroy_impl(ctx);
    return DCA_OK;
}
int dca_comm_abort(dca_ctx* ctx)
{
    if (!ctx) return DCA_ERR_ARG;          // no device switch, no stream work: this runs beside the thread that drives the context
    return dca_comm_abort_impl(ctx);
}
int dca_comm_info(dca_ctx* ctx, int* world, int* rank)
{
    CHECK_CTX(ctx);
    return dca_comm_info_impl(ctx, world, rank);
}
int dca_plm_set_native_comm(dca_ctx* ctx, int mode)
{
    CHECK_CTX(ctx);
    if (!ctx->plm) { dca_set_error("dca_plm_configure first"); return DCA_ERR_STATE; }
    return ctx->plm->set_native_comm(mode);        // drops the caller's hook only once the mode has been validated
}
int dca_mf_set_native_comm(dca_ctx* ctx, int on)
{
    CHECK_CTX(ctx);
    DCA_TRY(need_mf(ctx));
    if (on && !ctx->comm) { dca_set_error("no communicator: dca_comm_init first"); return DCA_ERR_STATE; }
    dca_mf_engine_set_native(ctx->mf, on != 0);
    return DCA_OK;
}
int dca_mf_set_row_window(dca_ctx* ctx, int first, int count)
{
    CHECK_CTX(ctx);
    DCA_TRY(need_mf(ctx));
    return dca_mf_engine_set_row_window(ctx->mf, first, count);
}
int dca_comm_allgather_host(dca_ctx* ctx, const double* mine, int n, double* all)
{
    CHECK_CTX(ctx);
    if (!mine || !all || n <= 0) return DCA_ERR_ARG;
    if (!ctx->comm) { dca_set_error("no communicator: dca_comm_init first"); return DCA_ERR_STATE; }
    const size_t total = (size_t)n * ctx->comm_world;
    double* d = nullptr;
    HIP_TRY(dca_dev_malloc(reinterpret_cast<void**>(&d), total * sizeof(double)));
    int rc = DCA_OK;
    // every rank contributes its values at its own offset of a zero vector: the sum IS the gathered vector (only sums are bound)
    if (hipMemsetAsync(d, 0, total * sizeof(double), ctx->stream) != hipSuccess ||
        hipMemcpyAsync(d + (size_t)n * ctx->comm_rank, mine, (size_t)n * sizeof(double), hipMemcpyHostToDevice, ctx->stream) != hipSuccess) rc = DCA_ERR_HIP;
    if (rc == DCA_OK) rc = dca_comm_native(ctx, DCA_COMM_ALL_REDUCE, d, total, DCA_F64);
    if (rc == DCA_OK && (hipMemcpyAsync(all, d, total * sizeof(double), hipMemcpyDeviceToHost, ctx->stream) != hipSuccess ||
                         hipStreamSynchronize(ctx->stream) != hipSuccess)) rc = DCA_ERR_HIP;
    dca_dev_free(d);
    return rc;
}
int dca_mf_set_reduce_hook(dca_ctx* ctx, dca_reduce_hook hook, void* user)
{
    CHECK_CTX(ctx);
    DCA_TRY(need_mf(ctx));
    dca_mf_engine_set_hook(ctx->mf, hook, user);
    return DCA_OK;
}
int dca_mf_fields(dca_ctx* ctx, double* out) { CHECK_CTX(ctx); DCA_TRY(need_mf(ctx)); if (!out) return DCA_ERR_ARG; return dca_mf_engine_fields(ctx->mf, out); }
int dca_mf_pair_couplings(dca_ctx* ctx, const int* pairs, int npairs, int shift, double* out)
{
    CHECK_CTX(ctx);
    DCA_TRY(need_mf(ctx));
    if (npairs < 0 || (npairs > 0 && (!pairs || !out))) return DCA_ERR_ARG;
    return dca_mf_engine_pair_couplings(ctx->mf, pairs, npairs, shift, out);
}
int dca_scores_order(dca_ctx* ctx, int32_t* order_out, int capacity)
{
    CHECK_CTX(ctx);
    if (!ctx->dLastScores) { dca_set_error("no score vector has been computed on this context"); return DCA_ERR_STATE; }
    if (!order_out || capacity < ctx->nLastScores) return DCA_ERR_ARG;
    return dca_scores_order_device(ctx, ctx->dLastScores, ctx->nLastScores, order_out);
}
int dca_mf_di_scores(dca_ctx* ctx, int apc, double* out) { CHECK_CTX(ctx); DCA_TRY(need_mf(ctx)); return dca_mf_engine_di(ctx->mf, apc, out); }
int dca_mf_run(dca_ctx* ctx, double pseudocount, int apc, double* scores_out, double* couplings_out)
{
    CHECK_CTX(ctx);
    DCA_TRY(need_mf(ctx));
    DCA_TRY(dca_mf_engine_corr(ctx->mf, pseudocount, nullptr));
    DCA_TRY(dca_mf_engine_couplings(ctx->mf, couplings_out));
    return dca_mf_engine_scores(ctx->mf, apc, scores_out);
}

// The inverse overlaps three streams (cholinv.hip, block sweep); the HIP runtime's default of four hardware queues makes streams of a
// process share queues as soon as it holds more than four.  Raised when the library is loaded -- it takes effect if the
// runtime has not been initialised yet (it reads the variable at its first call); the sweep also probes the streams it uses.
namespace { struct HwQueues { HwQueues() { setenv("GPU_MAX_HW_QUEUES", "8", 0); } } g_hwQueues; }

int dca_spd_inverse(dca_ctx* ctx, const double* A, int n, double* Ainv_out)
{
    CHECK_CTX(ctx);
    if (!A || !Ainv_out || n <= 0) return DCA_ERR_ARG;
    const int np = (int)round_up((size_t)n, 64);
    std::vector<double> padded((size_t)np * np, 0.0);
    // the LOWER triangle of A is what is inverted (as LAPACK's 'L' routines read it): mirrored here, because the device
    // path reads both halves of the matrix
    for (int r = 0; r < n; ++r) {
        memcpy(padded.data() + (size_t)r * np, A + (size_t)r * n, (size_t)(r + 1) * sizeof(double));
        for (int c = 0; c < r; ++c) padded[(size_t)c * np + r] = A[(size_t)r * n + c];
    }
    for (int r = n; r < np; ++r) padded[(size_t)r * np + r] = 1.0;
    double *dA = nullptr, *dWork = nullptr;
    HIP_TRY(dca_dev_malloc(reinterpret_cast<void**>(&dA), padded.size() * sizeof(double)));
    if (dca_dev_malloc(reinterpret_cast<void**>(&dWork), 2 * padded.size() * sizeof(double)) != hipSuccess) { dca_dev_free(dA); dca_set_error("out of device memory"); return DCA_ERR_NOMEM; }
    int info = 0;
    int rc = DCA_OK;
    if (hipMemcpy(dA, padded.data(), padded.size() * sizeof(double), hipMemcpyHostToDevice) != hipSuccess) rc = DCA_ERR_HIP;
    double* dInv = nullptr;
    if (rc == DCA_OK) rc = dca_spd_inverse_device(ctx, dA, np, dWork, &info, 1.0, &dInv);
    if (rc == DCA_OK && info != 0) { dca_set_error("matrix is not positive definite (pivot %d)", info); rc = DCA_ERR_NOT_SPD; }
    if (rc == DCA_OK) {
        if (hipMemcpy(padded.data(), dInv, padded.size() * sizeof(double), hipMemcpyDeviceToHost) != hipSuccess) rc = DCA_ERR_HIP;
        else for (int r = 0; r < n; ++r) memcpy(Ainv_out + (size_t)r * n, padded.data() + (size_t)r * np, (size_t)n * sizeof(double));
    }
    dca_dev_free(dA); dca_dev_free(dWork);
    return rc;
}

// ------------------------------------------------------------------ timing
int dca_set_profiling(dca_ctx* ctx, int on) { if (!ctx) return DCA_ERR_ARG; ctx->profiling = on != 0; ctx->profile_only.clear(); return DCA_OK; }
int dca_set_profiling_only(dca_ctx* ctx, const char* stage)
{
    if (!ctx) return DCA_ERR_ARG;
    ctx->profiling = stage && *stage;
    ctx->profile_only = stage ? stage : "";
    return DCA_OK;
}
int dca_reset_kernel_times(dca_ctx* ctx)
{
    CHECK_CTX(ctx);
    hipStreamSynchronize(ctx->stream);
    dca_flush_clocks(ctx);
    ctx->clocks.clear();
    return DCA_OK;
}
int dca_get_kernel_time(dca_ctx* ctx, const char* tag, double* ms_out, int* launches_out)
{
    CHECK_CTX(ctx);
    hipStreamSynchronize(ctx->stream);
    dca_flush_clocks(ctx);
    auto it = ctx->clocks.find(tag ? tag : "");
    if (ms_out) *ms_out = it == ctx->clocks.end() ? 0.0 : it->second.ms;
    if (launches_out) *launches_out = it == ctx->clocks.end() ? 0 : it->second.launches;
    return DCA_OK;
}

// ------------------------------------------------------------------ drop-in FFI
// plmdcaBackend.cpp:151-201: read + dedup, weights, initial x, L-BFGS; returns the
// parameter vector.  malloc/free are paired here (the reference mixes malloc/delete[]).
float* plmdcaBackend(unsigned short biomolecule, unsigned short num_site_states, const char* msa_file,
                     unsigned int seqs_len, float seqid, float lambda_h, float lambda_J,
                     unsigned int max_iteration, unsigned int num_threads, bool verbose)
{
    (void)num_threads;
    const int L = (int)seqs_len, q = (int)num_site_states;
    uint8_t* X = nullptr;
    int raw = 0;
    const int N = dca_read_msa_owned(msa_file, biomolecule, L, &X, &raw);       // one pass over the file
    if (N <= 0) { free(X); if (N == 0) dca_set_error("no sequences in %s", msa_file); return nullptr; }
    dca_ctx* ctx = nullptr;
    float* result = nullptr;
    dca_plm_stats st;
    memset(&st, 0, sizeof(st));
    const size_t P = dca_plm_num_params(L, q);
    if (dca_create(&ctx, 0, DCA_F32) != DCA_OK) { free(X); return nullptr; }
    const int rc_msa = dca_set_msa(ctx, X, N, L, q);
    free(X);
    if (rc_msa == DCA_OK &&
        dca_compute_weights(ctx, (double)seqid, DCA_F32) == DCA_OK &&
        dca_plm_configure(ctx, (double)lambda_h, (double)lambda_J, DCA_CARRY_CHUNKED, 0, 0, 0, 1) == DCA_OK &&
        dca_plm_init_x(ctx) == DCA_OK &&
        dca_plm_lbfgs_begin(ctx, (int)max_iteration, verbose ? 1 : 0) == DCA_OK &&
        dca_plm_lbfgs_iterate(ctx, max_iteration ? (int)max_iteration : 1 << 30, &st) == DCA_OK) {
        result = static_cast<float*>(malloc(P * sizeof(float)));
        if (!result) dca_set_error("out of host memory");
        else if (dca_plm_get_x(ctx, result, DCA_F32) != DCA_OK) { free(result); result = nullptr; }
    }
    if (verbose && result) {
        if (st.status == -1001) fprintf(stderr, "L-BFGS optimization completed\n");
        else { fprintf(stderr, "L-BFGS optimization terminated with status code = %d\n", st.status); fprintf(stderr, "fx = %f\n", st.fx); }
    }
    dca_destroy(ctx);
    return result;
}

void freeFieldsAndCouplings(void* h_and_J) { free(h_and_J); }

}  // extern "C"
