// Inter-GPU transport of the sharded transforms (one process per GPU).  The data path needs ONE primitive: a set of
// point-to-point sends and receives of device buffers that progress together (an all-to-all inside a group of ranks is the
// special case "one equal piece to and from every member").
//
//   RcclTransport      RCCL (librccl.so, loaded at run time so that single-GPU users carry no dependency) — grouped
//                      ncclSend / ncclRecv on the caller's HIP stream: one message per peer = one per xGMI link.
//   CallbackTransport  the host application moves the buffers (tests: torch.distributed/gloo staged through host memory,
//                      several ranks sharing one GPU — RCCL refuses two ranks on one device).
//
// Per-exchange HIP events (optional) split a sharded transform's time into communication and compute for bench.py.
#pragma once
#include <hip/hip_runtime.h>
#include <dlfcn.h>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <functional>
#include <mutex>
#include <thread>
#include <string>
#include <vector>
#include "../../include/ecfft_hip.h"

namespace ecfft {

struct P2P { int peer; void* ptr; size_t bytes; };

class Transport {
public:
    virtual ~Transport() { for (hipEvent_t e : ev_) (void)hipEventDestroy(e); if (vote_dev_) (void)hipFree(vote_dev_); }
    int rank = 0, world = 1, device = 0;
    // unblocks every exchange in flight and makes the later ones fail (ncclCommAbort): for a caller whose peer is gone.  The
    // communicator can only be destroyed afterwards.  Default: nothing to abort (the callback transport belongs to the host).
    virtual bool abort() { return false; }
    // every send and receive of the call progresses together (ncclGroupStart / ncclGroupEnd semantics); messages between one
    // pair of ranks match in the order given; buffers are device memory; the work is enqueued on `s`
    bool exchange(const P2P* sends, int ns, const P2P* recvs, int nr, hipStream_t s) {
        hipEvent_t a = nullptr, b = nullptr;
        if (stats_on_) { a = next_event(); b = next_event(); (void)hipEventRecord(a, s); }
        carried_ = true;
        bool ok = do_exchange(sends, ns, recvs, nr, s);
        if (stats_on_) { (void)hipEventRecord(b, s); pairs_.push_back({a, b}); }
        ++calls_;
        for (int i = 0; i < ns; ++i) bytes_ += (double)sends[i].bytes;
        return ok;
    }
    // Agreement across ALL ranks of the communicator (one int each way with every peer, then a host wait): true iff every rank
    // passed ok = true.  A collective call whose local preparation failed on one rank (allocation, table build) must not leave
    // its peers blocked in the exchanges that follow — every rank votes BEFORE the first exchange and all of them back out
    // together.  Synchronous; used once per (context, call shape) and between the phases of the collective EXIT-shard build.
    bool vote(bool ok, hipStream_t s) {
        voted_ = true;
        if (world <= 1) return ok;
        carried_ = true;
        // three ints per rank: the vote and the rank's link-striping threshold ("the same value on every rank" was an
        // unchecked contract, and exchange_striped decides locally — ranks that disagree would pair differently sized messages)
        constexpr int kW = 3;
        if (!vote_dev_ && hipMalloc(&vote_dev_, (size_t)(world + 1) * kW * sizeof(int)) != hipSuccess) { (void)hipGetLastError(); vote_dev_ = nullptr; }
        if (!vote_dev_) { fprintf(stderr, "ecfft: no device memory for the agreement buffer\n"); return false; }   // 12*(world+1) bytes: unreachable in practice
        const unsigned long long g = (unsigned long long)stripe_min_gain;
        int mine[kW] = {ok ? 1 : 0, (int)(unsigned)(g & 0xFFFFFFFFull), (int)(unsigned)(g >> 32)};
        if (hipMemcpyAsync(vote_dev_ + world * kW, mine, sizeof(mine), hipMemcpyHostToDevice, s) != hipSuccess) return false;
        if (hipStreamSynchronize(s) != hipSuccess) return false;
        std::vector<P2P> snd, rcv;
        for (int p = 0; p < world; ++p) if (p != rank) { snd.push_back({p, vote_dev_ + world * kW, sizeof(mine)}); rcv.push_back({p, vote_dev_ + p * kW, sizeof(mine)}); }
        if (!do_exchange(snd.data(), (int)snd.size(), rcv.data(), (int)rcv.size(), s)) return false;
        std::vector<int> all((size_t)(world + 1) * kW, 0);
        if (hipStreamSynchronize(s) != hipSuccess) return false;
        if (hipMemcpy(all.data(), vote_dev_, all.size() * sizeof(int), hipMemcpyDeviceToHost) != hipSuccess) return false;
        bool every = ok;
        for (int p = 0; p < world; ++p) {
            if (p == rank) continue;
            if (all[(size_t)p * kW] != 1) every = false;
            if (all[(size_t)p * kW + 1] != mine[1] || all[(size_t)p * kW + 2] != mine[2]) {
                fprintf(stderr, "ecfft: rank %d and rank %d disagree on the link-striping threshold (ecfft_comm_set_link_striping: the same value on every rank)\n", rank, p);
                every = false;
            }
        }
        return every;
    }
    // true once the communicator has carried an exchange or a vote: its link-striping threshold is frozen from then on
    bool used() const { return carried_; }
    // false until the ranks of this communicator have voted once (and thereby compared their striping thresholds): the first sharded
    // call on a communicator votes even when the context already knows the call's shape from another communicator
    bool voted() const { return voted_; }
    // ---- link striping (round 5) --------------------------------------------------------------------------------------------
    // xGMI is a full mesh of point-to-point links (7 per GPU).  The big exchanges of a split ENTER / EXIT are PAIRWISE — a rank hands
    // its whole share to one or two peers (the level's re-distribution, the pair level) — so one or two links carry 8 .. 16 MiB
    // while the other five idle, and the exchange is bound by ONE link.  Striped, a message travels as `world` slices: slice k goes
    // to rank k in a first grouped exchange and from there to its destination in a second one (the slices "k = source" and "k =
    // destination" take the direct link, one in each phase), so every link of the mesh carries 1/world of every message per phase:
    // two exchanges of M/8 per link instead of one of M (8 GPUs, one message per rank), i.e. 4x less time on the wire for one more
    // exchange latency.  Every rank of the communicator takes part in both phases — also ranks outside the group that exchanges,
    // they are the relays — which the SPMD call sequence of the sharded transforms guarantees; the global message pattern is a pure
    // function of the rank (`pat(q)` = the (destination, bytes) list rank q sends, in order), evaluated locally for every q, so no
    // pattern is ever communicated.  An exchange is striped only when the most loaded link gets lighter by `stripe_min_gain` bytes
    // over both phases.  OFF by default (round 6): striping has run bit-exactly over the callback transport, the RCCL
    // stand-in and host threads, but never on xGMI — it doubles the bytes a rank injects (relay traffic), adds one exchange and issues
    // ~2 (W - 1) sends and receives per group, so until an 8-GPU A/B exists a host opts in with ecfft_comm_set_link_striping (4 MiB is
    // the threshold the projection suggests: >= 85 us at 48 GB/s against one more exchange; bench.py --stripe-min-gain).
    struct MsgDesc { int dst; size_t bytes; };
    typedef std::function<void(int, std::vector<MsgDesc>&)> PatternFn;
    size_t stripe_min_gain = ~(size_t)0;
    bool exchange_striped(const PatternFn& pat, const P2P* sends, int ns, const P2P* recvs, int nr, void* stage, size_t stage_bytes, hipStream_t s) {
        const int W = world, me = rank;
        if (W < 4 || W > 64 || !stage || stripe_min_gain == ~(size_t)0) return exchange(sends, ns, recvs, nr, s);      // ~0: striping switched off
        // A mismatch between the pattern and what this rank really sends / receives is a bug in the caller's pattern function.  It is an
        // ERROR, not a fall-back to the plain exchange: the peers evaluate the same pattern and go on with the two-phase
        // exchange, so a rank that quietly sends its messages whole would pair them with their slices.
        auto bad = [&](const char* what) { fprintf(stderr, "ecfft: striped exchange on rank %d: %s\n", me, what); return false; };
        struct GM { int src, dst; size_t bytes, len; bool striped; int sidx, ridx; };      // sidx / ridx: index in my sends (src == me) / my receives (dst == me)
        std::vector<GM> gm; std::vector<MsgDesc> tmp;
        std::vector<size_t> D((size_t)W * W, 0), S1((size_t)W * W, 0), S2((size_t)W * W, 0);
        for (int q = 0; q < W; ++q) {
            tmp.clear(); pat(q, tmp);
            if (q == me) {                                                            // the pattern must describe what this rank really sends
                if ((int)tmp.size() != ns) return bad("the message pattern lists another number of sends than the call makes");
                for (int i = 0; i < ns; ++i) if (tmp[(size_t)i].dst != sends[i].peer || tmp[(size_t)i].bytes != sends[i].bytes) return bad("the message pattern disagrees with a send of the call (peer or size)");
            }
            int seen = 0;                                                             // messages of q to me so far
            for (size_t i = 0; i < tmp.size(); ++i) {
                const int d = tmp[i].dst; const size_t b = tmp[i].bytes;
                if (d < 0 || d >= W) return bad("the message pattern names a rank outside the communicator");
                GM g{q, d, b, b / (size_t)W, q != d && b >= ((size_t)64 << 10) && b % ((size_t)16 * W) == 0, q == me ? (int)i : -1, -1};
                if (d == me) {                                                        // the t-th message q sends me = my t-th receive from q
                    const int t = seen++; int c = 0;
                    for (int j = 0; j < nr; ++j) if (recvs[j].peer == q && c++ == t) { g.ridx = j; break; }
                    if (g.ridx < 0 || recvs[g.ridx].bytes != b) return bad("the message pattern disagrees with a receive of the call (missing or of another size)");
                }
                gm.push_back(g);
                if (q != d) D[(size_t)q * W + d] += b;
                if (g.striped) { for (int k = 0; k < W; ++k) { if (k != q) S1[(size_t)q * W + k] += g.len; if (k != d) S2[(size_t)k * W + d] += g.len; } }
                else if (q != d) S1[(size_t)q * W + d] += b;
            }
        }
        { int mine = 0; for (const GM& g : gm) mine += g.dst == me; if (mine != nr) return bad("the call posts receives the message pattern does not contain"); }
        size_t mD = 0, m1 = 0, m2 = 0, need = 0;
        for (size_t i = 0; i < D.size(); ++i) { if (D[i] > mD) mD = D[i]; if (S1[i] > m1) m1 = S1[i]; if (S2[i] > m2) m2 = S2[i]; }
        bool any = false;
        // the staging bound is rank-independent on purpose (every rank must take the same decision): all relayed slices together, of
        // which a rank stages at most its 1/world-th ... world-th part
        for (const GM& g : gm) { any = any || g.striped; if (g.striped) need += g.len; }
        if (!any || mD <= m1 + m2 || mD - (m1 + m2) < stripe_min_gain || need > stage_bytes) return exchange(sends, ns, recvs, nr, s);
        // phase 1: slice k of every striped message to rank k (slice `dst` lands in place, slice `src` waits for phase 2); messages that
        // are not striped (small, self) whole.  Between one pair of ranks the messages match in the order given: both sides walk the
        // global list (source-major, each source's own order).
        std::vector<P2P> ps, pr; char* st = (char*)stage; size_t so = 0;
        std::vector<size_t> slot(gm.size(), 0);
        for (const GM& g : gm) {
            if (g.src != me) continue;
            char* b = (char*)sends[g.sidx].ptr;
            if (!g.striped) ps.push_back({g.dst, b, g.bytes});
            else for (int k = 0; k < W; ++k) if (k != me) ps.push_back({k, b + (size_t)k * g.len, g.len});
        }
        for (size_t x = 0; x < gm.size(); ++x) {
            const GM& g = gm[x];
            if (!g.striped) { if (g.dst == me) pr.push_back({g.src, recvs[g.ridx].ptr, g.bytes}); continue; }
            if (g.src == me) continue;
            if (g.dst == me) pr.push_back({g.src, (char*)recvs[g.ridx].ptr + (size_t)me * g.len, g.len});
            else { slot[x] = so; pr.push_back({g.src, st + so, g.len}); so += g.len; }
        }
        if (!exchange(ps.data(), (int)ps.size(), pr.data(), (int)pr.size(), s)) return false;
        // phase 2: the relays forward; a source sends its own slice `src`
        ps.clear(); pr.clear();
        for (size_t x = 0; x < gm.size(); ++x) {
            const GM& g = gm[x];
            if (!g.striped || g.dst == me) continue;
            ps.push_back({g.dst, g.src == me ? (char*)sends[g.sidx].ptr + (size_t)me * g.len : st + slot[x], g.len});
        }
        for (int k = 0; k < W; ++k) {                                                // from relay k: its slice of every message destined to me, global order
            if (k == me) continue;
            for (const GM& g : gm) if (g.striped && g.dst == me) pr.push_back({k, (char*)recvs[g.ridx].ptr + (size_t)k * g.len, g.len});
        }
        return exchange(ps.data(), (int)ps.size(), pr.data(), (int)pr.size(), s);
    }
    void stats_enable(bool on) { stats_on_ = on; stats_reset(); }
    void stats_reset() { used_ = 0; pairs_.clear(); calls_ = 0; bytes_ = 0; }
    // caller has synchronised the stream(s)
    void stats_read(double* comm_ms, double* calls, double* bytes) {
        double ms = 0;
        for (auto& p : pairs_) { float t = 0; if (hipEventElapsedTime(&t, p.a, p.b) == hipSuccess) ms += t; }
        if (comm_ms) *comm_ms = ms;
        if (calls) *calls = (double)calls_;
        if (bytes) *bytes = bytes_;
    }
protected:
    virtual bool do_exchange(const P2P* sends, int ns, const P2P* recvs, int nr, hipStream_t s) = 0;
private:
    hipEvent_t next_event() { if (used_ == ev_.size()) { hipEvent_t e; (void)hipEventCreate(&e); ev_.push_back(e); } return ev_[used_++]; }
    struct Pair { hipEvent_t a, b; };
    std::vector<hipEvent_t> ev_; size_t used_ = 0; std::vector<Pair> pairs_;
    bool stats_on_ = false; uint64_t calls_ = 0; double bytes_ = 0;
    int* vote_dev_ = nullptr;
    bool carried_ = false, voted_ = false;
};

// ---- RCCL, bound at run time -------------------------------------------------------------------------------------------
struct RcclApi {
    struct UniqueId { char internal[128]; };             // ncclUniqueId (NCCL_UNIQUE_ID_BYTES = 128)
    typedef void* Comm;
    int (*GetUniqueId)(UniqueId*) = nullptr;
    int (*CommInitRank)(Comm*, int, UniqueId, int) = nullptr;
    int (*CommDestroy)(Comm) = nullptr;
    int (*CommAbort)(Comm) = nullptr;                    // optional
    int (*Send)(const void*, size_t, int, int, Comm, hipStream_t) = nullptr;
    int (*Recv)(void*, size_t, int, int, Comm, hipStream_t) = nullptr;
    int (*GroupStart)() = nullptr;
    int (*GroupEnd)() = nullptr;
    const char* (*GetErrorString)(int) = nullptr;
    void* handle = nullptr;
    bool ok() const { return handle && GetUniqueId && CommInitRank && CommDestroy && Send && Recv && GroupStart && GroupEnd; }

    static RcclApi& get() {
        static RcclApi api = load();
        return api;
    }
    // ecfft_comm_set_rccl_library: the library to bind instead of the mapped / system librccl.  Only before the first communicator
    // (the binding is made once per process); returns false afterwards.  An explicit ABI call, not an environment variable: the
    // shipped library does not dlopen a path it was not handed by its host.
    static bool set_library(const char* path) {
        std::lock_guard<std::mutex> g(path_mu());
        if (bound()) return false;
        override_path() = path ? path : "";
        return true;
    }
private:
    static std::mutex& path_mu() { static std::mutex m; return m; }
    static std::string& override_path() { static std::string p; return p; }
    static bool& bound() { static bool b = false; return b; }
    static RcclApi load() {
        RcclApi a;
        // reuse the copy that is already mapped (PyTorch-ROCm bundles its own librccl and has it loaded), else the system one
        const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1", "/opt/rocm/lib/librccl.so"};
        // an explicit library instead (a differently named RCCL build; the tests' stand-in that lets several ranks share one GPU,
        // tests/stub_rccl) — bound with RTLD_LOCAL so that its symbols do not shadow a librccl that is already mapped
        std::string ovr;
        { std::lock_guard<std::mutex> g(path_mu()); bound() = true; ovr = override_path(); }
        if (!ovr.empty()) {
            a.handle = dlopen(ovr.c_str(), RTLD_NOW | RTLD_LOCAL);
            if (!a.handle) { fprintf(stderr, "ecfft: RCCL library %s could not be loaded (%s)\n", ovr.c_str(), dlerror()); return a; }
        }
        if (!a.handle) for (const char* n : names) { a.handle = dlopen(n, RTLD_NOW | RTLD_NOLOAD | RTLD_GLOBAL); if (a.handle) break; }
        if (!a.handle) for (const char* n : names) { a.handle = dlopen(n, RTLD_NOW | RTLD_GLOBAL); if (a.handle) break; }
        if (!a.handle) { fprintf(stderr, "ecfft: librccl.so not found (%s)\n", dlerror()); return a; }
        a.GetUniqueId = (decltype(a.GetUniqueId))dlsym(a.handle, "ncclGetUniqueId");
        a.CommInitRank = (decltype(a.CommInitRank))dlsym(a.handle, "ncclCommInitRank");
        a.CommDestroy = (decltype(a.CommDestroy))dlsym(a.handle, "ncclCommDestroy");
        a.CommAbort = (decltype(a.CommAbort))dlsym(a.handle, "ncclCommAbort");
        a.Send = (decltype(a.Send))dlsym(a.handle, "ncclSend");
        a.Recv = (decltype(a.Recv))dlsym(a.handle, "ncclRecv");
        a.GroupStart = (decltype(a.GroupStart))dlsym(a.handle, "ncclGroupStart");
        a.GroupEnd = (decltype(a.GroupEnd))dlsym(a.handle, "ncclGroupEnd");
        a.GetErrorString = (decltype(a.GetErrorString))dlsym(a.handle, "ncclGetErrorString");
        return a;
    }
};

class RcclTransport : public Transport {
public:
    ~RcclTransport() override { if (RcclApi::Comm c = comm_.load()) (void)RcclApi::get().CommDestroy(c); }
    // may be called from another thread than the one blocked in an exchange (that is its purpose): the handle is swapped out atomically,
    // so exactly one caller aborts it and an exchange that starts afterwards sees no communicator
    bool abort() override {
        RcclApi& api = RcclApi::get();
        if (!api.CommAbort) return false;
        aborted_.store(true);                             // no new enqueue section starts from here on
        RcclApi::Comm c = comm_.exchange(nullptr);        // ncclCommAbort also frees the communicator: no ncclCommDestroy afterwards
        if (!c) return false;
        // an exchange may be between ncclGroupStart and ncclGroupEnd with `c` in hand: the enqueue section is host work
        // that ends within microseconds on RCCL (what BLOCKS is the stream behind it), so wait for it to leave before the
        // communicator is freed — bounded (0.2 s), because a transport library whose receive blocks on the HOST (the tests' stand-in)
        // is only released by the abort itself
        // (Dekker-style handshake with do_exchange: store comm_ / load in_enqueue_ here, increment in_enqueue_ / load comm_ there —
        // every one of the four accesses is sequentially consistent, which is what makes "either I see the section or it sees no
        // communicator" hold in the C++ memory model.)  If the 0.2 s expire — a transport whose enqueue itself blocks on the host —
        // the communicator is aborted under the section all the same: that is the documented way to release it, and the section's
        // remaining calls then fail inside the library instead of hanging.
        for (int spin = 0; in_enqueue_.load(std::memory_order_seq_cst) != 0 && spin < 2000; ++spin) std::this_thread::sleep_for(std::chrono::microseconds(100));
        return api.CommAbort(c) == 0;
    }
    bool init(const void* id128, int world_, int rank_, int device_) {
        RcclApi& api = RcclApi::get();
        if (!api.ok()) return false;
        RcclApi::UniqueId id; memcpy(id.internal, id128, sizeof(id.internal));
        world = world_; rank = rank_; device = device_;
        RcclApi::Comm c = nullptr;
        int rc = api.CommInitRank(&c, world, id, rank);
        if (rc != 0) { fprintf(stderr, "ecfft: ncclCommInitRank failed: %s\n", api.GetErrorString ? api.GetErrorString(rc) : "?"); return false; }
        comm_.store(c);
        return true;
    }
protected:
    bool do_exchange(const P2P* sends, int ns, const P2P* recvs, int nr, hipStream_t s) override {
        RcclApi& api = RcclApi::get();
        struct Enq { std::atomic<int>& n; explicit Enq(std::atomic<int>& n_) : n(n_) { n.fetch_add(1, std::memory_order_seq_cst); } ~Enq() { n.fetch_sub(1, std::memory_order_seq_cst); } } enq(in_enqueue_);
        RcclApi::Comm c = comm_.load();                   // read INSIDE the section: abort() either sees the section or this sees no communicator
        if (aborted_.load() || !c) return false;
        const int kChar = 0;                              // ncclChar / ncclInt8
        int rc = api.GroupStart();
        for (int i = 0; i < ns && rc == 0; ++i) rc = api.Send(sends[i].ptr, sends[i].bytes, kChar, sends[i].peer, c, s);
        for (int i = 0; i < nr && rc == 0; ++i) rc = api.Recv(recvs[i].ptr, recvs[i].bytes, kChar, recvs[i].peer, c, s);
        int rc2 = api.GroupEnd();
        if (rc == 0) rc = rc2;
        if (rc != 0) fprintf(stderr, "ecfft: RCCL exchange failed: %s\n", api.GetErrorString ? api.GetErrorString(rc) : "?");
        return rc == 0;
    }
private:
    std::atomic<RcclApi::Comm> comm_{nullptr};
    std::atomic<bool> aborted_{false};
    std::atomic<int> in_enqueue_{0};
};

class CallbackTransport : public Transport {
public:
    CallbackTransport(int world_, int rank_, int device_, ecfft_exchange_fn fn, void* user) : fn_(fn), user_(user) { world = world_; rank = rank_; device = device_; }
protected:
    bool do_exchange(const P2P* sends, int ns, const P2P* recvs, int nr, hipStream_t s) override {
        std::vector<int> sp(ns), rp(nr); std::vector<const void*> sptr(ns); std::vector<void*> rptr(nr); std::vector<size_t> sb(ns), rb(nr);
        for (int i = 0; i < ns; ++i) { sp[i] = sends[i].peer; sptr[i] = sends[i].ptr; sb[i] = sends[i].bytes; }
        for (int i = 0; i < nr; ++i) { rp[i] = recvs[i].peer; rptr[i] = recvs[i].ptr; rb[i] = recvs[i].bytes; }
        return fn_(user_, ns, sp.data(), sptr.data(), sb.data(), nr, rp.data(), rptr.data(), rb.data(), (void*)s) == 0;
    }
private:
    ecfft_exchange_fn fn_; void* user_;
};

#ifdef ECFFT_TEST_HOOKS
// ---- projection transport (measurement only) -------------------------------------------------------------------------------
// ONE rank of a `world`-rank job timed on its own: an exchange costs what the model says — `delay_us` of latency plus the largest
// per-peer message at `gbps` GB/s — as a kernel that spins on the constant 100 MHz wall clock ON THE CALLER'S STREAM, and moves
// the rank's own send buffers into its receive buffers (device-to-device), so that every kernel downstream runs on initialised
// memory of the right size.  The RESULTS ARE MEANINGLESS (no data of another rank ever arrives); the TIMELINE of the stream is
// that of a rank whose peers answer after exactly the modelled time: per-rank compute, launches, exchanges, bytes and the exposed
// communication time of the split transforms without the hardware (tools/split_project.py).  The transforms are data oblivious,
// so the wrong data changes no launch.
__global__ void k_spin_us(unsigned long long ticks) {
    const unsigned long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(8);
}
// every piece of one exchange in ONE launch, as RCCL moves a group (a launch per piece would bill the rank ~4 us of launch boundary per
// peer and exchange that no real transport pays)
struct ProjPieces { static constexpr int kMax = 64; void* dst[kMax]; const void* src[kMax]; unsigned long long n16[kMax]; int n; };
__global__ void k_proj_copy(ProjPieces pc) {
    const int piece = blockIdx.y;
    if (piece >= pc.n) return;
    const uint4* s = reinterpret_cast<const uint4*>(pc.src[piece]); uint4* d = reinterpret_cast<uint4*>(pc.dst[piece]);
    for (unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; i < pc.n16[piece]; i += (unsigned long long)gridDim.x * blockDim.x) d[i] = s[i];
}
class ProjectionTransport : public Transport {
public:
    ProjectionTransport(int world_, int rank_, int device_, double delay_us, double gbps) : delay_us_(delay_us), gbps_(gbps) { world = world_; rank = rank_; device = device_; }
protected:
    bool do_exchange(const P2P* sends, int ns, const P2P* recvs, int nr, hipStream_t s) override {
        size_t worst = 0;                                                      // one LINK per peer: the most loaded one sets the time (round 5: the
        {                                                                      // messages to one peer add up — round 4 billed only the largest message)
            size_t per[64] = {0};
            for (int i = 0; i < ns; ++i) if (sends[i].peer != rank && sends[i].peer >= 0 && sends[i].peer < 64) { per[sends[i].peer] += sends[i].bytes; if (per[sends[i].peer] > worst) worst = per[sends[i].peer]; }
        }
        double us = delay_us_ + (gbps_ > 0 ? (double)worst / (gbps_ * 1e3) : 0.0);
        bool remote = false;
        for (int i = 0; i < ns; ++i) remote = remote || sends[i].peer != rank;
        if (!remote) us = 0.0;                                                 // self send / receive only: no link involved
        if (us > 0) hipLaunchKernelGGL(k_spin_us, dim3(1), dim3(1), 0, s, (unsigned long long)(us * 100.0));
        if (ns <= 0) return nr == 0;                                           // nothing of this rank's to stand in for a peer's data
        ProjPieces pc; pc.n = 0; size_t most = 0;
        for (int i = 0; i < nr; ++i) {                                         // i-th receive <- i-th send (sizes agree for the group exchanges of the split transforms)
            const P2P& src = sends[i < ns ? i : ns - 1];
            const size_t b = recvs[i].bytes < src.bytes ? recvs[i].bytes : src.bytes;
            if (!b || recvs[i].ptr == src.ptr) continue;
            const bool vec = pc.n < ProjPieces::kMax && b % 16 == 0 && (reinterpret_cast<uintptr_t>(recvs[i].ptr) | reinterpret_cast<uintptr_t>(src.ptr)) % 16 == 0;
            if (!vec) { if (hipMemcpyAsync(recvs[i].ptr, src.ptr, b, hipMemcpyDeviceToDevice, s) != hipSuccess) return false; continue; }
            pc.dst[pc.n] = recvs[i].ptr; pc.src[pc.n] = src.ptr; pc.n16[pc.n] = b / 16; ++pc.n;
            if (b / 16 > most) most = b / 16;
        }
        if (pc.n) {
            size_t bx = (most + 255) / 256; if (bx > 256) bx = 256; if (bx < 1) bx = 1;
            hipLaunchKernelGGL(k_proj_copy, dim3((unsigned)bx, (unsigned)pc.n), dim3(256), 0, s, pc);
        }
        return hipGetLastError() == hipSuccess;
    }
private:
    double delay_us_, gbps_;
};

#endif  // ECFFT_TEST_HOOKS

}  // namespace ecfft

struct ecfft_comm {
    ecfft::Transport* t = nullptr;
    ~ecfft_comm() { delete t; }
};
