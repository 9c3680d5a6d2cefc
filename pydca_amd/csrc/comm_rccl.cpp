// Native collectives for the sharded paths: RCCL (librccl.so, the library behind torch.distributed's "nccl" backend on
// ROCm) is opened with dlopen and its collectives are enqueued on the context's own stream, in place on the library's
// device buffers -- an evaluation then has no host callback and no device synchronisation around its exchange step.
// One communicator per context (= one per process / GPU): rank 0 makes a 128-byte unique id (dca_comm_unique_id), the
// caller distributes it by any means it has (torch.distributed store, MPI, a file), every rank calls dca_comm_init.
// No reference counterpart: pydca is single-process.
#include <dlfcn.h>

#include <algorithm>
#include <mutex>

#include "dca_internal.h"

namespace {

// the part of rccl.h's ABI that is used here (stable since NCCL 2.x; RCCL keeps it)
struct RcclUniqueId { char internal[128]; };
typedef void* RcclComm;
enum { kRcclUint32 = 3, kRcclFloat32 = 7, kRcclFloat64 = 8, kRcclSum = 0 };

struct RcclApi {
    void* handle = nullptr;
    int (*GetUniqueId)(RcclUniqueId*) = nullptr;
    int (*CommInitRank)(RcclComm*, int, RcclUniqueId, int) = nullptr;
    int (*CommDestroy)(RcclComm) = nullptr;
    const char* (*GetErrorString)(int) = nullptr;
    int (*AllReduce)(const void*, void*, size_t, int, int, RcclComm, hipStream_t) = nullptr;
    int (*ReduceScatter)(const void*, void*, size_t, int, int, RcclComm, hipStream_t) = nullptr;
    int (*AllGather)(const void*, void*, size_t, int, RcclComm, hipStream_t) = nullptr;
    int (*GroupStart)() = nullptr;
    int (*GroupEnd)() = nullptr;
    int (*Send)(const void*, size_t, int, int, RcclComm, hipStream_t) = nullptr;
    int (*Recv)(void*, size_t, int, int, RcclComm, hipStream_t) = nullptr;
    int (*CommCount)(RcclComm, int*) = nullptr;         // optional: what the communicator itself reports (dca_comm_info)
    int (*CommUserRank)(RcclComm, int*) = nullptr;
    int (*CommAbort)(RcclComm) = nullptr;               // optional: dca_comm_abort
};
RcclApi g_api;

template <typename F> bool bind(void* handle, F& fn, const char* name)
{
    fn = reinterpret_cast<F>(dlsym(handle, name));
    if (!fn) dca_set_error("librccl lacks %s", name);
    return fn != nullptr;
}

// RCCL must run on the SAME HIP runtime as this library: it receives this library's stream handles and device
// pointers.  A process can hold two HIP runtimes (PyTorch wheels bundle their own libamdhip64.so next to the system's;
// which one this library is bound to depends on what was loaded first), so the default is the librccl that lies next to
// the libamdhip64 this library actually resolved its HIP calls to.
std::string rccl_next_to_hip(const char* file)
{
    Dl_info info;
    if (!dladdr(reinterpret_cast<const void*>(&hipStreamSynchronize), &info) || !info.dli_fname) return std::string();
    std::string dir(info.dli_fname);
    const size_t slash = dir.rfind('/');
    if (slash == std::string::npos) return std::string();
    return dir.substr(0, slash + 1) + file;
}

// One thread binds the API; the table is published (handle set) only when every entry point is resolved, so a second
// thread never sees a handle with null function pointers behind it (ctypes releases the GIL around these calls).
std::mutex g_api_mu;
int load_api(const char* path)
{
    std::lock_guard<std::mutex> lk(g_api_mu);
    if (g_api.handle) return DCA_OK;
    const std::string near1 = rccl_next_to_hip("librccl.so.1"), near2 = rccl_next_to_hip("librccl.so");
    const char* candidates[] = {path, getenv("DCA_RCCL_PATH"), near1.c_str(), near2.c_str(), "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so"};
    void* h = nullptr;
    for (const char* c : candidates) {
        if (!c || !*c) continue;
        h = dlopen(c, RTLD_NOW | RTLD_GLOBAL);
        if (h) break;
    }
    if (!h) { dca_set_error("cannot open librccl.so (%s)", dlerror()); return DCA_ERR_IO; }
    RcclApi api;
    bool ok = bind(h, api.GetUniqueId, "ncclGetUniqueId") && bind(h, api.CommInitRank, "ncclCommInitRank") &&
              bind(h, api.CommDestroy, "ncclCommDestroy") && bind(h, api.GetErrorString, "ncclGetErrorString") &&
              bind(h, api.AllReduce, "ncclAllReduce") && bind(h, api.ReduceScatter, "ncclReduceScatter") &&
              bind(h, api.AllGather, "ncclAllGather") && bind(h, api.GroupStart, "ncclGroupStart") && bind(h, api.GroupEnd, "ncclGroupEnd") &&
              bind(h, api.Send, "ncclSend") && bind(h, api.Recv, "ncclRecv");
    if (!ok) { dlclose(h); return DCA_ERR_IO; }
    api.CommCount = reinterpret_cast<decltype(api.CommCount)>(dlsym(h, "ncclCommCount"));
    api.CommUserRank = reinterpret_cast<decltype(api.CommUserRank)>(dlsym(h, "ncclCommUserRank"));
    api.CommAbort = reinterpret_cast<decltype(api.CommAbort)>(dlsym(h, "ncclCommAbort"));
    api.handle = h;
    g_api = api;
    return DCA_OK;
}

int rccl_type(int dtype) { return dtype == DCA_F32 ? kRcclFloat32 : kRcclFloat64; }

#define RCCL_TRY(expr)                                                                  \
    do {                                                                                \
        int _r = (expr);                                                                \
        if (_r != 0) { dca_set_error("%s failed: %s", #expr, g_api.GetErrorString(_r)); return DCA_ERR_HIP; } \
    } while (0)

}  // namespace

int dca_comm_unique_id_impl(const char* rccl_path, void* id128)
{
    DCA_TRY(load_api(rccl_path));
    RcclUniqueId id;
    RCCL_TRY(g_api.GetUniqueId(&id));
    memcpy(id128, id.internal, sizeof(id.internal));
    return DCA_OK;
}

int dca_comm_init_impl(dca_ctx* ctx, const char* rccl_path, const void* id128, int world, int rank)
{
    if (world < 1 || rank < 0 || rank >= world) { dca_set_error("dca_comm_init: bad rank / world"); return DCA_ERR_ARG; }
    DCA_TRY(load_api(rccl_path));
    HIP_TRY(hipSetDevice(ctx->device));
    dca_comm_destroy_impl(ctx);       // drains the stream first: collectives of the old communicator may still be queued
    RcclUniqueId id;
    memcpy(id.internal, id128, sizeof(id.internal));
    RcclComm comm = nullptr;
    RCCL_TRY(g_api.CommInitRank(&comm, world, id, rank));
    ctx->comm = comm;
    ctx->comm_rank = rank;
    ctx->comm_world = world;
    return DCA_OK;
}

// world size and rank as the communicator reports them (ncclCommCount / ncclCommUserRank), so that a caller can assert
// that the N processes it started really form ONE communicator of N ranks; the values handed to dca_comm_init where the
// library does not export the queries
int dca_comm_info_impl(dca_ctx* ctx, int* world, int* rank)
{
    if (!ctx->comm) { dca_set_error("no communicator: dca_comm_init first"); return DCA_ERR_STATE; }
    int w = ctx->comm_world, r = ctx->comm_rank;
    if (g_api.CommCount) RCCL_TRY(g_api.CommCount(static_cast<RcclComm>(ctx->comm), &w));
    if (g_api.CommUserRank) RCCL_TRY(g_api.CommUserRank(static_cast<RcclComm>(ctx->comm), &r));
    if (world) *world = w;
    if (rank) *rank = r;
    return DCA_OK;
}

// From ANOTHER thread than the one that drives the context (a watchdog that has seen a peer process die): ncclCommAbort makes
// the collectives that wait for the dead peer return with an error, so that the driving thread's call comes back instead of
// hanging for ever.  The communicator is gone afterwards (dca_comm_destroy only clears the context's fields).
int dca_comm_abort_impl(dca_ctx* ctx)
{
    if (!ctx->comm) return DCA_OK;
    if (!g_api.CommAbort) { dca_set_error("this librccl has no ncclCommAbort"); return DCA_ERR_STATE; }
    if (ctx->comm_aborted.exchange(true)) return DCA_OK;
    RCCL_TRY(g_api.CommAbort(static_cast<RcclComm>(ctx->comm)));
    return DCA_OK;
}

void dca_comm_destroy_impl(dca_ctx* ctx)
{
    if (ctx->comm && ctx->comm_aborted.load()) {
        ctx->comm = nullptr;               // released by ncclCommAbort
        ctx->comm_aborted.store(false);
    }
    if (ctx->comm && g_api.CommDestroy) {
        hipStreamSynchronize(ctx->stream);
        g_api.CommDestroy(static_cast<RcclComm>(ctx->comm));
    }
    ctx->comm = nullptr;
    ctx->comm_world = 0;
    dca_dev_free(ctx->commStage); ctx->commStage = nullptr; ctx->commStageBytes = 0;
}

// ---- the three collectives of the sharded optimiser, enqueued on ctx->stream (dca_comm_hook semantics: in place,
// `count` = whole vector; slices are count / world elements, rank r owns [r * slice, (r + 1) * slice))
namespace {
// mine[e] = sum over the ranks, IN RANK ORDER, of their pieces of this rank's slice: piece r comes from `stage` (the
// pieces received from the other ranks, packed in rank order without this rank's own) or, for r == rank, from `mine`
// itself.  A fixed order makes the owner's sum reproducible from run to run whatever order the pieces arrived in.
template <typename T>
__global__ __launch_bounds__(256)
void comm_sum_pieces_kernel(T* __restrict__ mine, const T* __restrict__ stage, size_t slice, int world, int rank)
{
    for (size_t e = blockIdx.x * (size_t)blockDim.x + threadIdx.x; e < slice; e += (size_t)gridDim.x * blockDim.x) {
        T acc = rank == 0 ? mine[e] : stage[e];
        for (int r = 1; r < world; ++r)
            acc += r == rank ? mine[e] : stage[(size_t)(r < rank ? r : r - 1) * slice + e];
        mine[e] = acc;
    }
}
}  // namespace

// Direct exchange (dca_plm_set_native_comm mode 3).  xGMI is a full mesh of point-to-point links, so a reduce-scatter can
// send slice j straight to rank j over the link that joins the two GPUs -- world - 1 concurrent transfers of
// count / world elements per rank, every link busy at once -- where a ring pushes (world - 1) / world of the vector
// through ONE link per hop.  The pieces land in a staging buffer ((world - 1) slices) and are summed locally in rank order;
// the all-gather is the same pattern with no sum and no staging (slice j of rank j lands in place).
static int comm_direct(dca_ctx* ctx, int op, void* buf, size_t count, int dtype)
{
    RcclComm comm = static_cast<RcclComm>(ctx->comm);
    const int world = ctx->comm_world, rank = ctx->comm_rank;
    const size_t esz = dtype == DCA_F32 ? 4 : 8;
    const size_t slice = count / (size_t)world;
    char* base = static_cast<char*>(buf);
    if (world == 1) return DCA_OK;
    if (op == DCA_COMM_REDUCE_SCATTER) {
        const size_t need = (size_t)(world - 1) * slice * esz;
        if (ctx->commStageBytes < need) {
            dca_dev_free(ctx->commStage); ctx->commStage = nullptr; ctx->commStageBytes = 0;
            HIP_TRY(dca_dev_malloc(&ctx->commStage, need, false));
            ctx->commStageBytes = need;
        }
        char* stage = static_cast<char*>(ctx->commStage);
        RCCL_TRY(g_api.GroupStart());
        int bad = 0;
        for (int k = 1; k < world && !bad; ++k) {               // partner order rotated by rank: no two ranks start on the same peer
            const int to = (rank + k) % world, from = (rank - k + world) % world;
            bad = g_api.Send(base + (size_t)to * slice * esz, slice, rccl_type(dtype), to, comm, ctx->stream);
            if (!bad) bad = g_api.Recv(stage + (size_t)(from < rank ? from : from - 1) * slice * esz, slice, rccl_type(dtype), from, comm, ctx->stream);
        }
        RCCL_TRY(g_api.GroupEnd());
        RCCL_TRY(bad);
        const unsigned blocks = (unsigned)std::min<size_t>((slice + 255) / 256, 256 * 8);
        if (dtype == DCA_F32)
            hipLaunchKernelGGL(comm_sum_pieces_kernel<float>, dim3(blocks), dim3(256), 0, ctx->stream,
                               reinterpret_cast<float*>(base + (size_t)rank * slice * esz), reinterpret_cast<const float*>(stage), slice, world, rank);
        else
            hipLaunchKernelGGL(comm_sum_pieces_kernel<double>, dim3(blocks), dim3(256), 0, ctx->stream,
                               reinterpret_cast<double*>(base + (size_t)rank * slice * esz), reinterpret_cast<const double*>(stage), slice, world, rank);
        HIP_TRY(hipGetLastError());
        return DCA_OK;
    }
    // all-gather: this rank's slice to every peer, every peer's slice into its place
    RCCL_TRY(g_api.GroupStart());
    int bad = 0;
    for (int k = 1; k < world && !bad; ++k) {
        const int to = (rank + k) % world, from = (rank - k + world) % world;
        bad = g_api.Send(base + (size_t)rank * slice * esz, slice, rccl_type(dtype), to, comm, ctx->stream);
        if (!bad) bad = g_api.Recv(base + (size_t)from * slice * esz, slice, rccl_type(dtype), from, comm, ctx->stream);
    }
    RCCL_TRY(g_api.GroupEnd());
    RCCL_TRY(bad);
    return DCA_OK;
}

int dca_comm_native(dca_ctx* ctx, int op, void* buf, size_t count, int dtype, bool direct)
{
    if (!ctx->comm) { dca_set_error("no communicator: dca_comm_init first"); return DCA_ERR_STATE; }
    RcclComm comm = static_cast<RcclComm>(ctx->comm);
    const size_t esz = dtype == DCA_F32 ? 4 : 8;
    if (op == DCA_COMM_ALL_REDUCE) {
        RCCL_TRY(g_api.AllReduce(buf, buf, count, rccl_type(dtype), kRcclSum, comm, ctx->stream));
        return DCA_OK;
    }
    if (direct) return comm_direct(ctx, op, buf, count, dtype);
    const size_t slice = count / (size_t)ctx->comm_world;
    char* mine = static_cast<char*>(buf) + (size_t)ctx->comm_rank * slice * esz;      // the in-place forms of both collectives
    if (op == DCA_COMM_REDUCE_SCATTER) RCCL_TRY(g_api.ReduceScatter(buf, mine, slice, rccl_type(dtype), kRcclSum, comm, ctx->stream));
    else RCCL_TRY(g_api.AllGather(mine, buf, slice, rccl_type(dtype), comm, ctx->stream));
    return DCA_OK;
}

// ---- grouped point-to-point transfers on ctx->stream (the column-strip decomposition's two exchanges, plm_engine.hip):
// dca_comm_p2p_begin, any number of dca_comm_p2p_send / _recv (matched in issue order per peer, as NCCL does), dca_comm_p2p_end
int dca_comm_p2p_begin(dca_ctx* ctx)
{
    if (!ctx->comm) { dca_set_error("no communicator: dca_comm_init first"); return DCA_ERR_STATE; }
    RCCL_TRY(g_api.GroupStart());
    return DCA_OK;
}
int dca_comm_p2p_send(dca_ctx* ctx, const void* buf, size_t count, int dtype, int peer)
{
    RCCL_TRY(g_api.Send(buf, count, rccl_type(dtype), peer, static_cast<RcclComm>(ctx->comm), ctx->stream));
    return DCA_OK;
}
int dca_comm_p2p_recv(dca_ctx* ctx, void* buf, size_t count, int dtype, int peer)
{
    RCCL_TRY(g_api.Recv(buf, count, rccl_type(dtype), peer, static_cast<RcclComm>(ctx->comm), ctx->stream));
    return DCA_OK;
}
int dca_comm_p2p_end(dca_ctx* ctx)
{
    RCCL_TRY(g_api.GroupEnd());
    return DCA_OK;
}

// all-reduce of a vector and of one double (gradient + objective, pair counts + Meff) as one group
int dca_comm_native_reduce(dca_ctx* ctx, void* vec, size_t count, int dtype, double* scalar_dev)
{
    if (!ctx->comm) { dca_set_error("no communicator: dca_comm_init first"); return DCA_ERR_STATE; }
    RcclComm comm = static_cast<RcclComm>(ctx->comm);
    RCCL_TRY(g_api.GroupStart());
    int r1 = g_api.AllReduce(vec, vec, count, rccl_type(dtype), kRcclSum, comm, ctx->stream);
    int r2 = scalar_dev ? g_api.AllReduce(scalar_dev, scalar_dev, 1, kRcclFloat64, kRcclSum, comm, ctx->stream) : 0;
    RCCL_TRY(g_api.GroupEnd());
    RCCL_TRY(r1);
    RCCL_TRY(r2);
    return DCA_OK;
}

int dca_comm_native_sum_u32(dca_ctx* ctx, uint32_t* buf, size_t count)
{
    if (!ctx->comm) { dca_set_error("no communicator: dca_comm_init first"); return DCA_ERR_STATE; }
    RCCL_TRY(g_api.AllReduce(buf, buf, count, kRcclUint32, kRcclSum, static_cast<RcclComm>(ctx->comm), ctx->stream));
    return DCA_OK;
}
