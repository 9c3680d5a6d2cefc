// TEST INFRASTRUCTURE ONLY -- a stand-in for librccl.so that implements the entry points csrc/comm_rccl.cpp binds
// (same names, same signatures) between THREADS of one process, so that the product's native exchange path (in-place
// reduce-scatter / all-gather / all-reduce on the context's stream, sharded weights, mfDCA counts) can be driven with
// world sizes 2 and 3 on the single GPU a test box has; real RCCL refuses two ranks on one device.
// A collective drains the caller's stream, stages through host memory and sums in RANK ORDER (every rank gets the same
// bits); ranks of one communicator meet at a generation barrier.  Not a performance model of anything.
#include <hip/hip_runtime.h>

#include <condition_variable>
#include <cstring>
#include <deque>
#include <map>
#include <mutex>
#include <string>
#include <utility>
#include <vector>

namespace {

struct Group {
    int nranks = 0, joined = 0;
    std::mutex mu;
    std::condition_variable cv;
    int arrived = 0;
    long generation = 0;
    std::vector<std::vector<char>> slots;
    std::vector<char> result;
};
struct Comm { Group* group; int rank; };

std::mutex g_mu;
std::map<std::string, Group*> g_groups;
int g_next_id = 1;

void barrier(Group* g)
{
    std::unique_lock<std::mutex> lk(g->mu);
    const long gen = g->generation;
    if (++g->arrived == g->nranks) { g->arrived = 0; ++g->generation; g->cv.notify_all(); }
    else g->cv.wait(lk, [&] { return g->generation != gen; });
}

size_t type_size(int t) { return t == 3 /* uint32 */ || t == 7 /* float */ ? 4 : t == 8 /* double */ ? 8 : 0; }

template <typename T> void sum_into(std::vector<char>& out, const std::vector<std::vector<char>>& in, size_t count)
{
    out.assign(count * sizeof(T), 0);
    T* o = reinterpret_cast<T*>(out.data());
    memcpy(o, in[0].data(), count * sizeof(T));
    for (size_t r = 1; r < in.size(); ++r) {
        const T* p = reinterpret_cast<const T*>(in[r].data());
        for (size_t i = 0; i < count; ++i) o[i] += p[i];
    }
}

// every rank contributes `count` elements; afterwards g->result holds the rank-ordered sum on all of them
int reduce_all(Comm* c, const void* send, size_t count, int type, hipStream_t stream)
{
    Group* g = c->group;
    const size_t esz = type_size(type);
    if (!esz) return 4;
    if (hipStreamSynchronize(stream) != hipSuccess) return 1;
    g->slots[c->rank].resize(count * esz);
    if (hipMemcpy(g->slots[c->rank].data(), send, count * esz, hipMemcpyDeviceToHost) != hipSuccess) return 1;
    barrier(g);
    if (c->rank == 0) {
        if (type == 3) sum_into<uint32_t>(g->result, g->slots, count);
        else if (type == 7) sum_into<float>(g->result, g->slots, count);
        else sum_into<double>(g->result, g->slots, count);
    }
    barrier(g);
    return 0;
}

}  // namespace

extern "C" {

struct ncclUniqueId { char internal[128]; };

int ncclGetUniqueId(ncclUniqueId* id)
{
    std::lock_guard<std::mutex> lk(g_mu);
    memset(id->internal, 0, sizeof(id->internal));
    snprintf(id->internal, sizeof(id->internal), "fake-rccl-%d", g_next_id++);
    return 0;
}

int ncclCommInitRank(void** comm, int nranks, ncclUniqueId id, int rank)
{
    Group* g;
    {
        std::lock_guard<std::mutex> lk(g_mu);
        Group*& slot = g_groups[std::string(id.internal)];
        if (!slot) { slot = new Group(); slot->nranks = nranks; slot->slots.resize(nranks); }
        g = slot;
        if (g->nranks != nranks || rank < 0 || rank >= nranks) return 4;
    }
    *comm = new Comm{g, rank};
    barrier(g);                       // like the real call: returns when every rank has joined
    return 0;
}

int ncclCommDestroy(void* comm) { delete static_cast<Comm*>(comm); return 0; }
int ncclCommCount(void* comm, int* count) { *count = static_cast<Comm*>(comm)->group->nranks; return 0; }
int ncclCommUserRank(void* comm, int* rank) { *rank = static_cast<Comm*>(comm)->rank; return 0; }
const char* ncclGetErrorString(int r) { return r == 0 ? "no error" : r == 4 ? "invalid argument" : "fake rccl: hip error"; }

// Point-to-point: only inside ncclGroupStart / ncclGroupEnd, the way the product issues its direct exchange (every rank
// posts its sends and receives, the group end carries them out).  Sends are staged in a host mailbox keyed (from, to);
// all ranks of the communicator must end a group with point-to-point operations together.
namespace {
struct P2P { bool send; void* buf; size_t bytes; int peer; Comm* comm; hipStream_t stream; };
thread_local int t_depth = 0;
thread_local std::vector<P2P> t_ops;
std::mutex g_mail_mu;
std::map<std::pair<Group*, std::pair<int, int>>, std::deque<std::vector<char>>> g_mail;      // FIFO per (from, to): several messages to one peer in a group match in issue order
}  // namespace

int ncclGroupStart() { ++t_depth; return 0; }
int ncclGroupEnd()
{
    if (--t_depth > 0 || t_ops.empty()) return 0;
    std::vector<P2P> ops;
    ops.swap(t_ops);
    Group* g = ops[0].comm->group;
    for (const P2P& o : ops) {
        if (!o.send) continue;
        if (hipStreamSynchronize(o.stream) != hipSuccess) return 1;
        std::vector<char> box(o.bytes);
        if (hipMemcpy(box.data(), o.buf, o.bytes, hipMemcpyDeviceToHost) != hipSuccess) return 1;
        std::lock_guard<std::mutex> lk(g_mail_mu);
        g_mail[{g, {o.comm->rank, o.peer}}].push_back(std::move(box));
    }
    barrier(g);
    int rc = 0;
    for (const P2P& o : ops) {
        if (o.send) continue;
        std::vector<char> box;
        {
            std::lock_guard<std::mutex> lk(g_mail_mu);
            auto it = g_mail.find({g, {o.peer, o.comm->rank}});
            if (it == g_mail.end() || it->second.empty() || it->second.front().size() != o.bytes) { rc = 4; continue; }
            box.swap(it->second.front());
            it->second.pop_front();
        }
        if (hipMemcpy(o.buf, box.data(), o.bytes, hipMemcpyHostToDevice) != hipSuccess) rc = 1;
    }
    barrier(g);
    return rc;
}

int ncclSend(const void* buf, size_t count, int type, int peer, void* comm, hipStream_t stream)
{
    if (t_depth <= 0 || !type_size(type)) return 4;
    t_ops.push_back(P2P{true, const_cast<void*>(buf), count * type_size(type), peer, static_cast<Comm*>(comm), stream});
    return 0;
}

int ncclRecv(void* buf, size_t count, int type, int peer, void* comm, hipStream_t stream)
{
    if (t_depth <= 0 || !type_size(type)) return 4;
    t_ops.push_back(P2P{false, buf, count * type_size(type), peer, static_cast<Comm*>(comm), stream});
    return 0;
}

int ncclAllReduce(const void* send, void* recv, size_t count, int type, int op, void* comm, hipStream_t stream)
{
    Comm* c = static_cast<Comm*>(comm);
    if (op != 0) return 4;
    if (int rc = reduce_all(c, send, count, type, stream)) return rc;
    hipError_t e = hipMemcpy(recv, c->group->result.data(), count * type_size(type), hipMemcpyHostToDevice);
    barrier(c->group);                // the result buffer is reused by the next collective
    return e == hipSuccess ? 0 : 1;
}

int ncclReduceScatter(const void* send, void* recv, size_t recvcount, int type, int op, void* comm, hipStream_t stream)
{
    Comm* c = static_cast<Comm*>(comm);
    if (op != 0) return 4;
    const size_t esz = type_size(type);
    if (int rc = reduce_all(c, send, recvcount * c->group->nranks, type, stream)) return rc;
    hipError_t e = hipMemcpy(recv, c->group->result.data() + (size_t)c->rank * recvcount * esz, recvcount * esz, hipMemcpyHostToDevice);
    barrier(c->group);
    return e == hipSuccess ? 0 : 1;
}

int ncclAllGather(const void* send, void* recv, size_t sendcount, int type, void* comm, hipStream_t stream)
{
    Comm* c = static_cast<Comm*>(comm);
    Group* g = c->group;
    const size_t esz = type_size(type);
    if (!esz) return 4;
    if (hipStreamSynchronize(stream) != hipSuccess) return 1;
    g->slots[c->rank].resize(sendcount * esz);
    if (hipMemcpy(g->slots[c->rank].data(), send, sendcount * esz, hipMemcpyDeviceToHost) != hipSuccess) return 1;
    barrier(g);
    hipError_t e = hipSuccess;
    for (int r = 0; r < g->nranks && e == hipSuccess; ++r)
        e = hipMemcpy(static_cast<char*>(recv) + (size_t)r * sendcount * esz, g->slots[r].data(), sendcount * esz, hipMemcpyHostToDevice);
    barrier(g);
    return e == hipSuccess ? 0 : 1;
}

}  // extern "C"
