// TEST-ONLY stand-in for librccl.so: the eight entry points ecfft_amd/csrc/transport.h binds with dlsym (ncclGetUniqueId,
// ncclCommInitRank, ncclCommDestroy, ncclCommAbort, ncclSend, ncclRecv, ncclGroupStart, ncclGroupEnd, ncclGetErrorString), moving
// bytes between PROCESSES THAT SHARE ONE GPU.  Real RCCL refuses several ranks on one device and a gpurun lease has one GPU, so
// without this the RcclTransport code path (grouped multi-peer sends / receives, sub-group peer lists, Transport::vote, the
// collective ecfft_build_exit_shard) has only ever run with world = 1.  Selected with ecfft_comm_set_rccl_library(<path of this library>).
//
// Not a performance path and not shipped: messages are staged through POSIX shared memory (device -> shm object -> device with
// blocking hipMemcpy), one shm object per message, a small control segment (named by the unique id) carries per-pair sequence
// numbers.  Semantics kept: every send / receive of a group progresses together (sends never block: unbounded buffering), the
// messages between one pair of ranks match in the order given, work is ordered with the caller's stream (the group drains the
// stream first and completes before it returns — a stronger ordering than RCCL's, never a weaker one).
#include <hip/hip_runtime.h>
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
#include <atomic>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

namespace {
constexpr int kMaxRanks = 16;
struct Control {
    std::atomic<uint32_t> attached, detached, aborted;
    uint32_t nranks;
    std::atomic<uint64_t> posted[kMaxRanks][kMaxRanks];      // [src][dst]: messages published so far
};
struct Comm {
    Control* ctl = nullptr; int rank = 0, nranks = 1; std::string token;
    uint64_t sent[kMaxRanks] = {}, received[kMaxRanks] = {};
};
struct Op { bool send; void* ptr; size_t bytes; int peer; Comm* comm; hipStream_t stream; };
thread_local int g_depth = 0;
thread_local std::vector<Op> g_ops;
const char* g_err = "no error";
double timeout_s() { const char* t = getenv("ECFFT_STUB_RCCL_TIMEOUT_S"); return t ? atof(t) : 120.0; }

std::string msg_name(const Comm* c, int src, int dst, uint64_t seq) {
    char b[160]; snprintf(b, sizeof(b), "/%s_m_%d_%d_%llu", c->token.c_str(), src, dst, (unsigned long long)seq); return b;
}
int do_send(const Op& o) {
    Comm* c = o.comm;
    const uint64_t seq = c->sent[o.peer]++;
    const std::string name = msg_name(c, c->rank, o.peer, seq);
    int fd = shm_open(name.c_str(), O_CREAT | O_EXCL | O_RDWR, 0600);
    if (fd < 0) { g_err = "stub rccl: shm_open (send) failed"; return 2; }
    const size_t len = o.bytes ? o.bytes : 1;
    if (ftruncate(fd, (off_t)len) != 0) { close(fd); g_err = "stub rccl: ftruncate failed"; return 2; }
    void* m = mmap(nullptr, len, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    close(fd);
    if (m == MAP_FAILED) { g_err = "stub rccl: mmap (send) failed"; return 2; }
    hipError_t e = o.bytes ? hipMemcpy(m, o.ptr, o.bytes, hipMemcpyDeviceToHost) : hipSuccess;
    munmap(m, len);
    if (e != hipSuccess) { g_err = "stub rccl: device -> host copy failed"; return 1; }
    c->ctl->posted[c->rank][o.peer].fetch_add(1, std::memory_order_release);
    return 0;
}
int do_recv(const Op& o) {
    Comm* c = o.comm;
    const uint64_t seq = c->received[o.peer]++;
    const auto t0 = std::chrono::steady_clock::now();
    while (c->ctl->posted[o.peer][c->rank].load(std::memory_order_acquire) <= seq) {
        if (c->ctl->aborted.load()) { g_err = "stub rccl: communicator aborted"; return 3; }
        if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > timeout_s()) { g_err = "stub rccl: receive timed out"; return 2; }
        std::this_thread::sleep_for(std::chrono::microseconds(50));
    }
    const std::string name = msg_name(c, o.peer, c->rank, seq);
    int fd = shm_open(name.c_str(), O_RDONLY, 0600);
    if (fd < 0) { g_err = "stub rccl: shm_open (receive) failed"; return 2; }
    struct stat st; fstat(fd, &st);
    const size_t len = (size_t)st.st_size;
    void* m = mmap(nullptr, len, PROT_READ, MAP_SHARED, fd, 0);
    close(fd);
    if (m == MAP_FAILED) { g_err = "stub rccl: mmap (receive) failed"; return 2; }
    int rc = 0;
    if ((o.bytes ? o.bytes : 1) != len) { g_err = "stub rccl: message size mismatch between sender and receiver"; rc = 4; }
    else if (o.bytes && hipMemcpy(o.ptr, m, o.bytes, hipMemcpyHostToDevice) != hipSuccess) { g_err = "stub rccl: host -> device copy failed"; rc = 1; }
    munmap(m, len);
    shm_unlink(name.c_str());
    return rc;
}
int run_ops(std::vector<Op>& ops) {
    for (const Op& o : ops) if (hipStreamSynchronize(o.stream) != hipSuccess) { g_err = "stub rccl: stream synchronise failed"; return 1; }
    int rc = 0;
    for (const Op& o : ops) if (o.send && rc == 0) rc = do_send(o);          // sends never block
    for (const Op& o : ops) if (!o.send && rc == 0) rc = do_recv(o);
    return rc;
}
}  // namespace

extern "C" {
int ncclGetUniqueId(char* id128) {
    memset(id128, 0, 128);
    unsigned long long r = 0;
    FILE* f = fopen("/dev/urandom", "rb");
    if (f) { if (fread(&r, sizeof(r), 1, f) != 1) r = 0; fclose(f); }
    snprintf(id128, 128, "ecfftstub_%d_%llx", (int)getpid(), r);
    return 0;
}
struct NcclUid { char internal[128]; };
int ncclCommInitRank(void** comm, int nranks, NcclUid id, int rank) {
    if (nranks < 1 || nranks > kMaxRanks || rank < 0 || rank >= nranks) { g_err = "stub rccl: bad rank / world"; return 4; }
    Comm* c = new Comm; c->rank = rank; c->nranks = nranks; c->token = std::string(id.internal, strnlen(id.internal, 127));
    const std::string name = "/" + c->token + "_ctl";
    int fd = shm_open(name.c_str(), O_CREAT | O_RDWR, 0600);
    if (fd < 0 || ftruncate(fd, sizeof(Control)) != 0) { g_err = "stub rccl: control segment"; delete c; return 2; }
    c->ctl = (Control*)mmap(nullptr, sizeof(Control), PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);     // fresh shm is zero-filled
    close(fd);
    if (c->ctl == MAP_FAILED) { g_err = "stub rccl: control mmap"; delete c; return 2; }
    c->ctl->nranks = (uint32_t)nranks;
    c->ctl->attached.fetch_add(1);
    const auto t0 = std::chrono::steady_clock::now();
    while (c->ctl->attached.load() < (uint32_t)nranks) {                                            // ncclCommInitRank is collective
        if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > timeout_s()) { g_err = "stub rccl: init timed out"; return 2; }
        std::this_thread::sleep_for(std::chrono::microseconds(200));
    }
    *comm = c;
    return 0;
}
int ncclCommDestroy(void* comm) {
    Comm* c = (Comm*)comm;
    if (!c) return 0;
    const bool last = c->ctl->detached.fetch_add(1) + 1 == (uint32_t)c->nranks;
    munmap(c->ctl, sizeof(Control));
    if (last) shm_unlink(("/" + c->token + "_ctl").c_str());
    delete c;
    return 0;
}
int ncclCommAbort(void* comm) {          // like the real one it also retires the communicator (no ncclCommDestroy afterwards); the
    Comm* c = (Comm*)comm;               // mapping stays: another thread of the caller may still sit in do_recv on it
    if (!c) return 0;
    c->ctl->aborted.store(1);
    if (c->ctl->detached.fetch_add(1) + 1 == (uint32_t)c->nranks) shm_unlink(("/" + c->token + "_ctl").c_str());
    return 0;
}
int ncclGroupStart() { ++g_depth; return 0; }
int ncclGroupEnd() {
    if (g_depth <= 0) { g_err = "stub rccl: ncclGroupEnd without ncclGroupStart"; return 4; }
    if (--g_depth > 0) return 0;
    std::vector<Op> ops; ops.swap(g_ops);
    return run_ops(ops);
}
int ncclSend(const void* buf, size_t count, int datatype, int peer, void* comm, hipStream_t s) {
    (void)datatype;                                              // the transport sends bytes (ncclChar)
    Op o{true, const_cast<void*>(buf), count, peer, (Comm*)comm, s};
    if (g_depth > 0) { g_ops.push_back(o); return 0; }
    std::vector<Op> one{o}; return run_ops(one);
}
int ncclRecv(void* buf, size_t count, int datatype, int peer, void* comm, hipStream_t s) {
    (void)datatype;
    Op o{false, buf, count, peer, (Comm*)comm, s};
    if (g_depth > 0) { g_ops.push_back(o); return 0; }
    std::vector<Op> one{o}; return run_ops(one);
}
const char* ncclGetErrorString(int rc) { return rc == 0 ? "no error" : g_err; }
}
