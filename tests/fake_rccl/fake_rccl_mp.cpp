// TEST INFRASTRUCTURE ONLY -- a stand-in for librccl.so between PROCESSES that share one GPU box (tests/fake_rccl/fake_rccl.cpp
// is the same between the threads of one process).  Real RCCL refuses two ranks on one device, so this is what lets
// `bench.py --gpus 2` run its NATIVE exchange path -- its own communicators, the start-up timing of the four schemes, the
// column-strip decomposition -- end to end on a single-GPU test box (tests/test_api_gpu.py::test_bench_native_path_two_processes).
// Everything is staged through files in /dev/shm named after the communicator's unique id; ranks meet at a counter barrier in
// a shared mapping; reductions are summed in RANK order by every rank (same bits everywhere).  Not a performance model.
#include <hip/hip_runtime.h>

#include <atomic>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

namespace {

struct Control { std::atomic<int> arrived; std::atomic<long> generation; };
struct Comm { std::string id; int nranks, rank; Control* ctl; long p2pSeq; std::atomic<bool> aborted{false}; };
struct P2P { bool send; void* buf; size_t bytes; int peer; Comm* comm; hipStream_t stream; };
thread_local int t_depth = 0;
thread_local std::vector<P2P> t_ops;

size_t type_size(int t) { return t == 3 || t == 7 ? 4 : t == 8 ? 8 : 0; }
std::string path_of(const Comm* c, const std::string& what) { return "/dev/shm/" + c->id + "." + what; }

// false: the communicator was aborted while this rank waited for its peers (ncclCommAbort from a watchdog thread)
bool barrier(Comm* c)
{
    if (c->aborted.load()) return false;
    const long gen = c->ctl->generation.load();
    if (c->ctl->arrived.fetch_add(1) + 1 == c->nranks) {
        c->ctl->arrived.store(0);
        c->ctl->generation.fetch_add(1);
    } else {
        while (c->ctl->generation.load() == gen) {
            if (c->aborted.load()) return false;
            usleep(50);
        }
    }
    return true;
}

bool write_file(const std::string& p, const void* data, size_t bytes)
{
    FILE* f = fopen(p.c_str(), "wb");
    if (!f) return false;
    const bool ok = fwrite(data, 1, bytes, f) == bytes;
    fclose(f);
    return ok;
}
bool read_file(const std::string& p, void* data, size_t bytes)
{
    FILE* f = fopen(p.c_str(), "rb");
    if (!f) return false;
    const bool ok = fread(data, 1, bytes, f) == bytes;
    fclose(f);
    return ok;
}

template <typename T> void add_into(std::vector<char>& acc, const std::vector<char>& v, size_t count)
{
    T* a = reinterpret_cast<T*>(acc.data());
    const T* b = reinterpret_cast<const T*>(v.data());
    for (size_t i = 0; i < count; ++i) a[i] += b[i];
}

// every rank contributes count elements; result = rank-ordered sum, on every rank
int reduce_all(Comm* c, const void* send, size_t count, int type, hipStream_t stream, std::vector<char>& result)
{
    const size_t esz = type_size(type);
    if (!esz) return 4;
    if (hipStreamSynchronize(stream) != hipSuccess) return 1;
    std::vector<char> mine(count * esz);
    if (hipMemcpy(mine.data(), send, count * esz, hipMemcpyDeviceToHost) != hipSuccess) return 1;
    if (!write_file(path_of(c, "slot" + std::to_string(c->rank)), mine.data(), mine.size())) return 1;
    if (!barrier(c)) return 6;
    result.assign(count * esz, 0);
    std::vector<char> v(count * esz);
    for (int r = 0; r < c->nranks; ++r) {
        if (!read_file(path_of(c, "slot" + std::to_string(r)), v.data(), v.size())) return 1;
        if (r == 0) result = v;
        else if (type == 3) add_into<uint32_t>(result, v, count);
        else if (type == 7) add_into<float>(result, v, count);
        else add_into<double>(result, v, count);
    }
    if (!barrier(c)) return 6;      // nobody overwrites its slot before everyone has read it
    return 0;
}

}  // namespace

extern "C" {

struct ncclUniqueId { char internal[128]; };

int ncclGetUniqueId(ncclUniqueId* id)
{
    static std::atomic<int> counter{0};
    memset(id->internal, 0, sizeof(id->internal));
    snprintf(id->internal, sizeof(id->internal), "fake_rccl_mp_%d_%d", (int)getpid(), counter.fetch_add(1));
    return 0;
}

int ncclCommInitRank(void** comm, int nranks, ncclUniqueId id, int rank)
{
    if (rank < 0 || rank >= nranks) return 4;
    Comm* c = new Comm;
    c->id = id.internal; c->nranks = nranks; c->rank = rank; c->ctl = nullptr; c->p2pSeq = 0;
    const std::string p = path_of(c, "ctl");
    const int fd = open(p.c_str(), O_RDWR | O_CREAT, 0600);
    if (fd < 0) return 1;
    if (ftruncate(fd, 4096) != 0) { close(fd); return 1; }       // a new file reads as zeros: arrived = generation = 0
    void* m = mmap(nullptr, 4096, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    close(fd);
    if (m == MAP_FAILED) return 1;
    c->ctl = static_cast<Control*>(m);
    *comm = c;
    barrier(c);
    return 0;
}

// the real call also releases the communicator; this one only makes every wait on it return (the object is leaked: a test aid)
int ncclCommAbort(void* comm)
{
    static_cast<Comm*>(comm)->aborted.store(true);
    return 0;
}

int ncclCommDestroy(void* comm)
{
    Comm* c = static_cast<Comm*>(comm);
    if (!barrier(c)) return 6;
    unlink(path_of(c, "slot" + std::to_string(c->rank)).c_str());
    if (c->rank == 0) unlink(path_of(c, "ctl").c_str());
    munmap(c->ctl, 4096);
    delete c;
    return 0;
}
int ncclCommCount(void* comm, int* count) { *count = static_cast<Comm*>(comm)->nranks; return 0; }
int ncclCommUserRank(void* comm, int* rank) { *rank = static_cast<Comm*>(comm)->rank; return 0; }
const char* ncclGetErrorString(int r) { return r == 0 ? "no error" : r == 4 ? "invalid argument" : "fake rccl (mp): i/o or hip error"; }

int ncclGroupStart() { ++t_depth; return 0; }
int ncclGroupEnd()
{
    if (--t_depth > 0 || t_ops.empty()) return 0;
    std::vector<P2P> ops;
    ops.swap(t_ops);
    Comm* c = ops[0].comm;
    // messages of one group to one peer are matched in issue order: file <from>.<to>.<k>
    std::vector<int> sent(c->nranks, 0), got(c->nranks, 0);
    for (const P2P& o : ops) {
        if (!o.send) continue;
        if (hipStreamSynchronize(o.stream) != hipSuccess) return 1;
        std::vector<char> box(o.bytes);
        if (hipMemcpy(box.data(), o.buf, o.bytes, hipMemcpyDeviceToHost) != hipSuccess) return 1;
        if (!write_file(path_of(c, "p2p." + std::to_string(c->rank) + "." + std::to_string(o.peer) + "." + std::to_string(sent[o.peer]++)), box.data(), box.size())) return 1;
    }
    if (!barrier(c)) return 6;
    int rc = 0;
    for (const P2P& o : ops) {
        if (o.send) continue;
        const std::string p = path_of(c, "p2p." + std::to_string(o.peer) + "." + std::to_string(c->rank) + "." + std::to_string(got[o.peer]++));
        struct stat st;
        if (stat(p.c_str(), &st) != 0 || (size_t)st.st_size != o.bytes) { rc = 4; continue; }
        std::vector<char> box(o.bytes);
        if (!read_file(p, box.data(), o.bytes) || hipMemcpy(o.buf, box.data(), o.bytes, hipMemcpyHostToDevice) != hipSuccess) rc = 1;
        unlink(p.c_str());
    }
    if (!barrier(c)) return 6;
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
    if (op != 0) return 4;
    std::vector<char> result;
    if (int rc = reduce_all(static_cast<Comm*>(comm), send, count, type, stream, result)) return rc;
    return hipMemcpy(recv, result.data(), result.size(), hipMemcpyHostToDevice) == hipSuccess ? 0 : 1;
}
int ncclReduceScatter(const void* send, void* recv, size_t recvcount, int type, int op, void* comm, hipStream_t stream)
{
    Comm* c = static_cast<Comm*>(comm);
    if (op != 0) return 4;
    std::vector<char> result;
    if (int rc = reduce_all(c, send, recvcount * c->nranks, type, stream, result)) return rc;
    const size_t esz = type_size(type);
    return hipMemcpy(recv, result.data() + (size_t)c->rank * recvcount * esz, recvcount * esz, hipMemcpyHostToDevice) == hipSuccess ? 0 : 1;
}
int ncclAllGather(const void* send, void* recv, size_t sendcount, int type, void* comm, hipStream_t stream)
{
    Comm* c = static_cast<Comm*>(comm);
    const size_t esz = type_size(type);
    if (!esz) return 4;
    if (hipStreamSynchronize(stream) != hipSuccess) return 1;
    std::vector<char> mine(sendcount * esz);
    if (hipMemcpy(mine.data(), send, mine.size(), hipMemcpyDeviceToHost) != hipSuccess) return 1;
    if (!write_file(path_of(c, "slot" + std::to_string(c->rank)), mine.data(), mine.size())) return 1;
    if (!barrier(c)) return 6;
    int rc = 0;
    std::vector<char> v(sendcount * esz);
    for (int r = 0; r < c->nranks && !rc; ++r) {
        if (!read_file(path_of(c, "slot" + std::to_string(r)), v.data(), v.size())) rc = 1;
        else if (hipMemcpy(static_cast<char*>(recv) + (size_t)r * sendcount * esz, v.data(), v.size(), hipMemcpyHostToDevice) != hipSuccess) rc = 1;
    }
    if (!barrier(c)) return 6;
    return rc;
}

}  // extern "C"
