// C-ABI of the MI355X ECFFT hot path — implements include/ecfft_hip.h.
#include <hip/hip_runtime.h>
#include <cstring>
#include <memory>
#include <new>
#include "device_tree.h"
#include "wire_parse.h"
#include "../../include/ecfft_hip.h"
#ifdef ECFFT_TEST_HOOKS
#include "../../include/ecfft_hip_hooks.h"    // test / measurement entry points: never in the shipped library
#endif

using namespace ecfft;

struct ecfft_ctx {
    int field;
    int device;
    std::unique_ptr<DeviceChain<Secp256k1>> secp;
    std::unique_ptr<DeviceChain<M31>> m31;
    void* stage = nullptr;       // device staging for host-pointer calls (in + out), lazily sized
    size_t stage_bytes = 0;
    hipEvent_t last_op = nullptr;   // completion of the previous transform on this context: calls on other streams wait for
                                    // it before touching the shared scratch buffers (contexts serialise, see ecfft_hip.h)
    ~ecfft_ctx() { if (stage) (void)hipFree(stage); if (last_op) (void)hipEventDestroy(last_op); }
};

namespace {

inline bool is_pow2(size_t n) { return n && (n & (n - 1)) == 0; }

bool have_device(int device) {
    int cnt = 0;
    if (hipGetDeviceCount(&cnt) != hipSuccess || cnt <= 0 || device < 0 || device >= cnt) {
        fprintf(stderr, "ecfft: no usable HIP device %d (count=%d) — this library has no CPU fallback\n", device, cnt);
        return false;
    }
    return true;
}

// Every entry point that touches the device selects the context's device and puts the caller's current device back on
// the way out, whatever path it leaves by.
struct DeviceGuard {
    int prev = -1; bool ok = false;
    explicit DeviceGuard(int device) {
        if (hipGetDevice(&prev) != hipSuccess) { (void)hipGetLastError(); prev = -1; }
        ok = hipSetDevice(device) == hipSuccess;
    }
    ~DeviceGuard() { if (prev >= 0) (void)hipSetDevice(prev); }
};
template <class F>
int finish_build(HostTree<F>&& ht, int device, std::unique_ptr<DeviceChain<F>>& slot) {
    slot.reset(new (std::nothrow) DeviceChain<F>());
    if (!slot) return ECFFT_ERR_HIP;
    DeviceGuard dev(device);
    if (!dev.ok) return ECFFT_ERR_HIP;
    if (slot->build(std::move(ht), device)) return ECFFT_OK;
    return slot->bad_points() ? ECFFT_ERR_BAD_ARG : ECFFT_ERR_HIP;       // a leaf that is a pole of its isogeny map is a caller error
}

// plain <-> crate representation on host buffers
void secp_from_mont_host(Fe256* v, size_t n) {   // x*2^256 -> x : multiply by 2^-256 = (2^32+977)^-1
    Fe256 r = Secp256k1::zero(); r.l[0] = 977; r.l[1] = 1;
    Fe256 rinv = Secp256k1::inv(r);
    for (size_t i = 0; i < n; ++i) v[i] = Secp256k1::mul(v[i], rinv);
}
void secp_to_mont_host(Fe256* v, size_t n) { for (size_t i = 0; i < n; ++i) v[i] = Secp256k1::to_mont(v[i]); }

bool ensure_stage(ecfft_ctx* c, size_t bytes) {
    if (c->stage_bytes >= bytes) return true;
    if (c->stage) (void)hipFree(c->stage);
    c->stage = nullptr; c->stage_bytes = 0;
    if (hipMalloc(&c->stage, bytes) != hipSuccess) return false;
    c->stage_bytes = bytes;
    return true;
}

// order this call after the previous one on the same context, whatever streams they use
bool op_begin(ecfft_ctx* c, hipStream_t s) {
    if (!c->last_op) return hipEventCreateWithFlags(&c->last_op, hipEventDisableTiming) == hipSuccess;
    return hipStreamWaitEvent(s, c->last_op, 0) == hipSuccess;
}
bool op_end(ecfft_ctx* c, hipStream_t s) { return hipEventRecord(c->last_op, s) == hipSuccess; }

// Orders a call after the previous one on the context (op_begin) and records its completion (op_end) on EVERY exit path
// after a successful begin — an early error return must not leave `last_op` pointing before work that was enqueued.
struct OpScope {
    ecfft_ctx* c; hipStream_t s; bool ok;
    OpScope(ecfft_ctx* c_, hipStream_t s_) : c(c_), s(s_), ok(op_begin(c_, s_)) {}
    ~OpScope() { if (ok) (void)op_end(c, s); }
};
// no C++ exception (allocation failure inside the chain) crosses the C ABI
template <class Fn>
int guarded(Fn fn) {
    try { return fn(); }
    catch (const std::bad_alloc&) { return ECFFT_ERR_HIP; }
    catch (...) { return ECFFT_ERR_HIP; }
}

enum Op { OP_ENTER, OP_EXIT, OP_EXTEND };

// a context made by ecfft_build_extend_shard / ecfft_build_enter_shard / ecfft_build_exit_shard holds one rank's share of the tables of ONE split transform
// and nothing else: only that transform (ecfft_extend_sharded[_layout] / ecfft_enter_sharded / ecfft_exit_sharded with the same size, world and rank;
// run_sharded checks) plus ecfft_tree_size, ecfft_field, ecfft_ctx_device_bytes, ecfft_profile_* and ecfft_ctx_destroy accept it
inline bool shard_only(const ecfft_ctx* c) { return c->field == ECFFT_FIELD_SECP256K1 ? c->secp->shard_mode() : c->m31->shard_mode(); }

template <class F>
int run_op(ecfft_ctx* c, DeviceChain<F>& ch, Op op, const void* in, void* out, size_t len, size_t count, int moiety,
           int mem, void* stream) {
    using E = typename F::elem;
    if (!in || !out) return ECFFT_ERR_BAD_ARG;
    if (!is_pow2(len)) return ECFFT_ERR_NOT_POW2;
    if (count == 0) return ECFFT_ERR_BAD_ARG;
    size_t need_tree = (op == OP_EXTEND) ? len * 2 : len;
    if (need_tree > ch.size()) return ECFFT_ERR_TREE_TOO_SMALL;          // "FFTree is too small"
    if (op == OP_EXTEND && moiety != ECFFT_S0 && moiety != ECFFT_S1) return ECFFT_ERR_BAD_ARG;
    hipStream_t s = (hipStream_t)stream;
    if (mem != ECFFT_MEM_HOST && mem != ECFFT_MEM_DEVICE) return ECFFT_ERR_BAD_ARG;
    DeviceGuard dev(c->device);
    if (!dev.ok) return ECFFT_ERR_HIP;
    std::lock_guard<std::mutex> guard(ch.lock());
    OpScope scope(c, s);
    if (!scope.ok) return ECFFT_ERR_HIP;
    size_t total = len * count, bytes = total * sizeof(E);
    const E* din = (const E*)in; E* dout = (E*)out;
    if (mem == ECFFT_MEM_HOST) {
        if (!ensure_stage(c, 2 * bytes)) return ECFFT_ERR_HIP;
        din = (const E*)c->stage; dout = (E*)((char*)c->stage + bytes);
        if (hipMemcpyAsync((void*)din, in, bytes, hipMemcpyHostToDevice, s) != hipSuccess) return ECFFT_ERR_HIP;
    }
    bool ok = true;
    switch (op) {
        case OP_ENTER: ok = ch.enter(din, dout, len, count, s); break;
        case OP_EXIT: ok = ch.exit(din, dout, len, count, s); break;
        case OP_EXTEND: ok = ch.extend_api(din, dout, len, count, moiety, s); break;
    }
    if (!ok || hipGetLastError() != hipSuccess) return ECFFT_ERR_HIP;
    if (mem == ECFFT_MEM_HOST) {
        if (hipMemcpyAsync(out, dout, bytes, hipMemcpyDeviceToHost, s) != hipSuccess) return ECFFT_ERR_HIP;
        if (hipStreamSynchronize(s) != hipSuccess) return ECFFT_ERR_HIP;
    }
    return ECFFT_OK;
}

// standard = true: plain standard-form residues (the FFTree wire format) instead of the crate's in-memory representation
template <class F>
int table_of(DeviceChain<F>& ch, size_t m, int which, void* host_out, size_t cap, size_t* count, bool standard = false) {
    using E = typename F::elem;
    if (!is_pow2(m)) return ECFFT_ERR_NOT_POW2;
    if (m > ch.size()) return ECFFT_ERR_TREE_TOO_SMALL;
    unsigned l = ilog2(m);
    const typename DeviceChain<F>::Tree& T = ch.tree(l);
    const E* src = nullptr; size_t cnt = 0;
    switch (which) {
        case ECFFT_TBL_XNN_S: src = T.xnn; cnt = m; break;
        case ECFFT_TBL_XNN_S_INV: src = T.xnn_inv; cnt = m; break;
        case ECFFT_TBL_Z0_S1: src = T.z0_s1; cnt = m / 2; break;
        case ECFFT_TBL_Z1_S0: src = T.z1_s0; cnt = m / 2; break;
        case ECFFT_TBL_Z0_INV_S1: src = T.z0_inv_s1; cnt = m / 2; break;
        case ECFFT_TBL_Z1_INV_S0: src = T.z1_inv_s0; cnt = m / 2; break;
        case ECFFT_TBL_Z0Z0_REM_XNN_S: src = T.z0z0; cnt = m < 2 ? 0 : m; break;   // empty for the 1-leaf tree (src/fftree.rs:459)
        case ECFFT_TBL_Z1Z1_REM_XNN_S: src = T.z1z1; cnt = m < 2 ? 0 : m; break;
        case ECFFT_TBL_F: cnt = 2 * m; break;
        case ECFFT_TBL_RECOMBINE: case ECFFT_TBL_DECOMPOSE: cnt = 4 * m; break;
        default: return ECFFT_ERR_BAD_ARG;
    }
    if (count) *count = cnt;
    if (!host_out) return ECFFT_OK;
    if (cap < cnt) return ECFFT_ERR_BAD_ARG;
    E* o = (E*)host_out;
    if (which == ECFFT_TBL_RECOMBINE || which == ECFFT_TBL_DECOMPOSE) {
        std::lock_guard<std::mutex> guard(ch.lock());
        void* d = nullptr;
        if (hipMalloc(&d, cnt * sizeof(E)) != hipSuccess) return ECFFT_ERR_HIP;
        bool ok = ch.export_matrices(l, which == ECFFT_TBL_DECOMPOSE, (E*)d, nullptr, standard) &&
                  hipMemcpy(o, d, cnt * sizeof(E), hipMemcpyDeviceToHost) == hipSuccess;
        (void)hipFree(d);
        return ok ? ECFFT_OK : ECFFT_ERR_HIP;      // already in the crate representation
    }
    if (which == ECFFT_TBL_F) {
        // f of T_m: every (N/m)-th element of each layer of the top tree (src/fftree.rs:471-478), gathered on the device — only
        // the 2m entries asked for cross the bus
        std::lock_guard<std::mutex> guard(ch.lock());
        void* d = nullptr;
        if (hipMalloc(&d, cnt * sizeof(E)) != hipSuccess) return ECFFT_ERR_HIP;
        bool ok = ch.gather_f(l, (E*)d, nullptr) && hipMemcpy(o, d, cnt * sizeof(E), hipMemcpyDeviceToHost) == hipSuccess;
        (void)hipFree(d);
        if (!ok) return ECFFT_ERR_HIP;
    } else if (cnt) {
        if (!src) return ECFFT_ERR_BAD_ARG;
        if (hipMemcpy(o, src, cnt * sizeof(E), hipMemcpyDeviceToHost) != hipSuccess) return ECFFT_ERR_HIP;
    }
    if (standard) return ECFFT_OK;
    if constexpr (std::is_same<F, Secp256k1>::value) secp_to_mont_host(o, cnt);
    return ECFFT_OK;
}

}  // namespace

namespace {
template <class F>
int run_shard(ecfft_ctx* c, DeviceChain<F>& ch, void* buf, size_t e, int moiety, unsigned log_p, unsigned rank, int which, int mem, void* stream) {
    using E = typename F::elem;
    if (!buf) return ECFFT_ERR_BAD_ARG;
    if (!is_pow2(e)) return ECFFT_ERR_NOT_POW2;
    if (2 * e > ch.size()) return ECFFT_ERR_TREE_TOO_SMALL;
    if (moiety != ECFFT_S0 && moiety != ECFFT_S1) return ECFFT_ERR_BAD_ARG;
    if (((size_t)2 << log_p) > e || rank >= (1u << log_p)) return ECFFT_ERR_BAD_ARG;   // need >= 2 elements per rank
    if (mem != ECFFT_MEM_HOST && mem != ECFFT_MEM_DEVICE) return ECFFT_ERR_BAD_ARG;
    DeviceGuard dev(c->device);
    if (!dev.ok) return ECFFT_ERR_HIP;
    hipStream_t s = (hipStream_t)stream;
    std::lock_guard<std::mutex> guard(ch.lock());
    OpScope scope(c, s);
    if (!scope.ok) return ECFFT_ERR_HIP;
    size_t bytes = (e >> log_p) * sizeof(E);
    E* d = (E*)buf;
    if (mem == ECFFT_MEM_HOST) {
        if (!ensure_stage(c, bytes)) return ECFFT_ERR_HIP;
        d = (E*)c->stage;
        if (hipMemcpyAsync(d, buf, bytes, hipMemcpyHostToDevice, s) != hipSuccess) return ECFFT_ERR_HIP;
    }
    if (which == 2) ch.extend_local_block(d, e, moiety, log_p, s);
    else ch.extend_top_cyclic(d, e, moiety, log_p, rank, which == 1, s);
    if (hipGetLastError() != hipSuccess) return ECFFT_ERR_HIP;
    if (mem == ECFFT_MEM_HOST) {
        if (hipMemcpyAsync(buf, d, bytes, hipMemcpyDeviceToHost, s) != hipSuccess) return ECFFT_ERR_HIP;
        if (hipStreamSynchronize(s) != hipSuccess) return ECFFT_ERR_HIP;
    }
    return ECFFT_OK;
}
}  // namespace


// ---- remaining algorithms: generic driver with up to 3 inputs and 1 output, staged through device temporaries for host buffers
enum Alg { ALG_MEXTEND, ALG_REDC, ALG_MOD, ALG_VANISH, ALG_DEGREE };
template <class F>
int run_alg(ecfft_ctx* c, DeviceChain<F>& ch, Alg alg, const void* in0, const void* in1, const void* in2, void* out, size_t len,
            size_t count, int moiety, int mem, void* stream, size_t* degree) {
    using E = typename F::elem;
    if (!in0 || (alg != ALG_DEGREE && !out)) return ECFFT_ERR_BAD_ARG;
    if ((alg == ALG_REDC || alg == ALG_MOD) && !in1) return ECFFT_ERR_BAD_ARG;
    if (alg == ALG_MOD && !in2) return ECFFT_ERR_BAD_ARG;
    if (alg == ALG_DEGREE && !degree) return ECFFT_ERR_BAD_ARG;
    if (!is_pow2(len)) return ECFFT_ERR_NOT_POW2;
    if (count == 0) return ECFFT_ERR_BAD_ARG;
    size_t need = (alg == ALG_MEXTEND || alg == ALG_VANISH) ? 2 * len : len;
    if (need > ch.size()) return ECFFT_ERR_TREE_TOO_SMALL;
    if ((alg == ALG_MEXTEND || alg == ALG_REDC) && moiety != ECFFT_S0 && moiety != ECFFT_S1) return ECFFT_ERR_BAD_ARG;
    if (mem != ECFFT_MEM_HOST && mem != ECFFT_MEM_DEVICE) return ECFFT_ERR_BAD_ARG;
    if ((alg == ALG_REDC || alg == ALG_MOD) && len < 2) {   // size-1 tree has no moieties: the reference would index out of bounds
        return ECFFT_ERR_BAD_ARG;
    }
    DeviceGuard dev(c->device);
    if (!dev.ok) return ECFFT_ERR_HIP;
    hipStream_t s = (hipStream_t)stream;
    std::lock_guard<std::mutex> guard(ch.lock());
    OpScope scope(c, s);
    if (!scope.ok) return ECFFT_ERR_HIP;
    size_t n_in = len * count, n_out = (alg == ALG_VANISH ? 2 * len : len) * count;
    const E *d0 = (const E*)in0, *d1 = (const E*)in1, *d2 = (const E*)in2; E* dout = (E*)out;
    bool ok = true;
    if (mem == ECFFT_MEM_HOST) {
        // host buffers: inputs and the output are staged through ONE context-owned device buffer that grows on demand
        // and is reused by later calls (no hipMalloc / hipFree per call)
        size_t need = n_in + (in1 ? len : 0) + (in2 ? len : 0) + (alg != ALG_DEGREE ? n_out : 0);
        if (!ensure_stage(c, need * sizeof(E))) return ECFFT_ERR_HIP;
        E* p = (E*)c->stage;
        auto stage_in = [&](const void* h, size_t n, const E** d) -> bool {
            if (!h) return true;
            if (hipMemcpyAsync(p, h, n * sizeof(E), hipMemcpyHostToDevice, s) != hipSuccess) return false;
            *d = p; p += n; return true;
        };
        ok = stage_in(in0, n_in, &d0) && stage_in(in1, len, &d1) && stage_in(in2, len, &d2);
        if (alg != ALG_DEGREE) dout = p;
    }
    if (ok) {
        switch (alg) {
            case ALG_MEXTEND: ok = ch.api_mextend(d0, dout, len, count, moiety, s); break;
            case ALG_REDC: ok = ch.api_redc(d0, d1, dout, len, moiety, s); break;
            case ALG_MOD: ok = ch.api_modular_reduce(d0, d1, d2, dout, len, s); break;
            case ALG_VANISH: ok = ch.api_vanish(d0, dout, len, s); break;
            case ALG_DEGREE: ok = ch.api_degree(d0, len, s, degree); break;
        }
    }
    if (ok && mem == ECFFT_MEM_HOST && alg != ALG_DEGREE)
        ok = hipMemcpyAsync(out, dout, n_out * sizeof(E), hipMemcpyDeviceToHost, s) == hipSuccess && hipStreamSynchronize(s) == hipSuccess;
    return ok ? ECFFT_OK : ECFFT_ERR_HIP;
}
#define ECFFT_DISPATCH_ALG(...) (ctx->field == ECFFT_FIELD_SECP256K1 ? run_alg(ctx, *ctx->secp, __VA_ARGS__) : run_alg(ctx, *ctx->m31, __VA_ARGS__))

namespace {
template <class F>
int run_table_fma(ecfft_ctx* c, DeviceChain<F>& ch, void* out, const void* x, const void* y, size_t cnt, size_t m, int which, size_t t_off,
                  size_t t_stride, int mode, int mem, void* stream) {
    using E = typename F::elem;
    if (!out || !x) return ECFFT_ERR_BAD_ARG;
    if (!is_pow2(m)) return ECFFT_ERR_NOT_POW2;
    if (m > ch.size()) return ECFFT_ERR_TREE_TOO_SMALL;
    if (cnt == 0) return ECFFT_OK;
    DeviceGuard dev(c->device);
    if (!dev.ok) return ECFFT_ERR_HIP;
    hipStream_t s = (hipStream_t)stream;
    std::lock_guard<std::mutex> guard(ch.lock());
    OpScope scope(c, s);                                          // after the context's previous call, whatever stream it ran on (staging buffer, pooled temporaries)
    if (!scope.ok) return ECFFT_ERR_HIP;
    const E *dx = (const E*)x, *dy = (const E*)y; E* dout = (E*)out;
    size_t bytes = cnt * sizeof(E);
    if (mem == ECFFT_MEM_HOST) {
        if (!ensure_stage(c, 3 * bytes)) return ECFFT_ERR_HIP;
        E* st = (E*)c->stage;
        if (hipMemcpyAsync(st, x, bytes, hipMemcpyHostToDevice, s) != hipSuccess) return ECFFT_ERR_HIP;
        dx = st;
        if (y) { if (hipMemcpyAsync(st + cnt, y, bytes, hipMemcpyHostToDevice, s) != hipSuccess) return ECFFT_ERR_HIP; dy = st + cnt; }
        dout = st + 2 * cnt;
    } else if (mem != ECFFT_MEM_DEVICE) return ECFFT_ERR_BAD_ARG;
    if (!ch.table_fma(dout, dx, dy, cnt, ilog2(m), which, t_off, t_stride, mode, s)) return ECFFT_ERR_BAD_ARG;
    if (mem == ECFFT_MEM_HOST) {
        if (hipMemcpyAsync(out, dout, bytes, hipMemcpyDeviceToHost, s) != hipSuccess) return ECFFT_ERR_HIP;
        if (hipStreamSynchronize(s) != hipSuccess) return ECFFT_ERR_HIP;
    }
    return ECFFT_OK;
}
}  // namespace

#ifdef ECFFT_TEST_HOOKS
template <class F>
int run_selftest(int op, const void* a, const void* b, const void* c, void* out, size_t n, int device) {
    using E = typename F::elem;
    if (!a || !b || !out || ((op == 0 || op == 4) && !c) || op < 0 || op > 5) return ECFFT_ERR_BAD_ARG;
    if (!have_device(device)) return ECFFT_ERR_HIP;
    DeviceGuard dev(device);
    if (!dev.ok) return ECFFT_ERR_HIP;
    E *da = nullptr, *db = nullptr, *dc = nullptr, *dout = nullptr;
    size_t bytes = n * sizeof(E);
    bool ok = hipMalloc(&da, bytes) == hipSuccess && hipMalloc(&db, bytes) == hipSuccess && hipMalloc(&dc, bytes) == hipSuccess && hipMalloc(&dout, bytes) == hipSuccess;
    ok = ok && hipMemcpy(da, a, bytes, hipMemcpyHostToDevice) == hipSuccess && hipMemcpy(db, b, bytes, hipMemcpyHostToDevice) == hipSuccess;
    if (ok && c) ok = hipMemcpy(dc, c, bytes, hipMemcpyHostToDevice) == hipSuccess;
    if (ok) {
        const E *pa = da, *pb = db, *pc = dc; E* po = dout;
        foreach_n(nullptr, n, [=] __device__(size_t i) {
            E r;
            if (op == 0) r = F::mul_add(pa[i], pb[i], pc[i]);
            else if (op == 1) r = F::mul(pa[i], pb[i]);
            else if (op == 2) r = F::sub(pa[i], pb[i]);
            else if (op == 3) r = F::add(pa[i], pb[i]);
            else if (op == 4) r = F::tmul_add(F::to_table(pa[i]), pb[i], pc[i]);
            else r = F::tmul(F::to_table(pa[i]), pb[i]);
            po[i] = r;
        });
        ok = hipDeviceSynchronize() == hipSuccess && hipMemcpy(out, dout, bytes, hipMemcpyDeviceToHost) == hipSuccess;
    }
    (void)hipFree(da); (void)hipFree(db); (void)hipFree(dc); (void)hipFree(dout);
    return ok ? ECFFT_OK : ECFFT_ERR_HIP;
}
#endif  // ECFFT_TEST_HOOKS

// dependent chain x <- T*x + c per lane: the table multiply of the butterfly kernels with nothing else around it
template <class F>
__global__ __launch_bounds__(256) void k_mul_chain(const typename F::elem* t, typename F::elem* x, int iters) {
    size_t g = (size_t)blockIdx.x * 256 + threadIdx.x;
    typename F::telem tv = F::to_table(t[g]);
    typename F::elem xv = x[g], cv = t[g];
#pragma unroll 1
    for (int i = 0; i < iters; ++i) xv = F::tmul_add(tv, xv, cv);
    x[g] = F::canon(xv);
}

// the same chain with shader-clock (s_memtime) and constant 100 MHz (wall_clock64) stamps around it: effective shader clock
// of the chip while every SIMD runs the kernels' multiply — what DVFS really grants under this instruction mix
struct ClockStamp { unsigned long long cyc, wall; };
template <class F>
__global__ __launch_bounds__(256) void k_clock_probe(const typename F::elem* t, typename F::elem* x, ClockStamp* st, int iters) {
    size_t g = (size_t)blockIdx.x * 256 + threadIdx.x;
    typename F::telem tv = F::to_table(t[g]);
    typename F::elem xv = x[g], cv = t[g];
    unsigned long long c0 = __builtin_amdgcn_s_memtime(), w0 = wall_clock64();
#pragma unroll 1
    for (int i = 0; i < iters; ++i) xv = F::tmul_add(tv, xv, cv);
    unsigned long long c1 = __builtin_amdgcn_s_memtime(), w1 = wall_clock64();
    x[g] = F::canon(xv);
    if (threadIdx.x == 0) { st[blockIdx.x].cyc = c1 - c0; st[blockIdx.x].wall = w1 - w0; }
}
template <class F>
int run_shader_clock(int device, double* mhz) {
    using E = typename F::elem;
    if (!mhz) return ECFFT_ERR_BAD_ARG;
    if (!have_device(device)) return ECFFT_ERR_HIP;
    DeviceGuard dev(device);
    hipDeviceProp_t p;
    if (!dev.ok || hipGetDeviceProperties(&p, device) != hipSuccess) return ECFFT_ERR_HIP;
    const int blocks = p.multiProcessorCount * 4, iters = sizeof(E) == 32 ? 2048 : 65536;    // a few ms at 4 waves per SIMD
    const size_t n = (size_t)blocks * 256;
    std::vector<E> h(n);
    memset(h.data(), 0x35, n * sizeof(E));
    E *dt = nullptr, *dx = nullptr; ClockStamp* ds = nullptr;
    std::vector<ClockStamp> hs(blocks);
    bool ok = hipMalloc(&dt, n * sizeof(E)) == hipSuccess && hipMalloc(&dx, n * sizeof(E)) == hipSuccess && hipMalloc(&ds, blocks * sizeof(ClockStamp)) == hipSuccess &&
              hipMemcpy(dt, h.data(), n * sizeof(E), hipMemcpyHostToDevice) == hipSuccess && hipMemcpy(dx, h.data(), n * sizeof(E), hipMemcpyHostToDevice) == hipSuccess;
    for (int r = 0; ok && r < 2; ++r) {
        hipLaunchKernelGGL(k_clock_probe<F>, dim3(blocks), dim3(256), 0, nullptr, (const E*)dt, dx, ds, iters);
        ok = hipDeviceSynchronize() == hipSuccess;
    }
    ok = ok && hipMemcpy(hs.data(), ds, blocks * sizeof(ClockStamp), hipMemcpyDeviceToHost) == hipSuccess;
    (void)hipFree(dt); (void)hipFree(dx); (void)hipFree(ds);
    if (!ok) return ECFFT_ERR_HIP;
    double cyc = 0, wall = 0;
    for (const ClockStamp& c : hs) { cyc += (double)c.cyc; wall += (double)c.wall; }
    *mhz = wall > 0 ? cyc / wall * 100.0 : 0.0;
    return ECFFT_OK;
}

template <class F>
int run_mul_ceiling(int device, int waves_per_simd, double* mul_per_s) {
    using E = typename F::elem;
    if (!mul_per_s || waves_per_simd < 1 || waves_per_simd > 8) return ECFFT_ERR_BAD_ARG;
    hipDeviceProp_t p;
    if (!have_device(device)) return ECFFT_ERR_HIP;
    DeviceGuard dev(device);
    if (!dev.ok || hipGetDeviceProperties(&p, device) != hipSuccess) return ECFFT_ERR_HIP;
    const int blocks = p.multiProcessorCount * waves_per_simd, iters = sizeof(E) == 32 ? 512 : 8192;   // one 256-thread block = one wave per SIMD
    const size_t n = (size_t)blocks * 256;
    std::vector<E> h(n);
    uint64_t sd = 88172645463325252ull;
    for (size_t i = 0; i < n; ++i) {
        unsigned char* b = reinterpret_cast<unsigned char*>(&h[i]);
        for (size_t k = 0; k < sizeof(E); ++k) { sd ^= sd << 13; sd ^= sd >> 7; sd ^= sd << 17; b[k] = (unsigned char)(sd >> 24); }
        b[sizeof(E) - 1] &= 0x3F;                                                    // < p for both fields
    }
    E *dt = nullptr, *dx = nullptr;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    bool ok = hipMalloc(&dt, n * sizeof(E)) == hipSuccess && hipMalloc(&dx, n * sizeof(E)) == hipSuccess &&
              hipMemcpy(dt, h.data(), n * sizeof(E), hipMemcpyHostToDevice) == hipSuccess && hipMemcpy(dx, h.data(), n * sizeof(E), hipMemcpyHostToDevice) == hipSuccess &&
              hipEventCreate(&e0) == hipSuccess && hipEventCreate(&e1) == hipSuccess;
    float best = 1e30f;
    for (int r = 0; ok && r < 4; ++r) {
        ok = hipEventRecord(e0, nullptr) == hipSuccess;
        hipLaunchKernelGGL(k_mul_chain<F>, dim3(blocks), dim3(256), 0, nullptr, (const E*)dt, dx, iters);
        ok = ok && hipEventRecord(e1, nullptr) == hipSuccess && hipEventSynchronize(e1) == hipSuccess;
        float ms = 0; ok = ok && hipEventElapsedTime(&ms, e0, e1) == hipSuccess;
        if (ok && r > 0 && ms < best) best = ms;
    }
    if (e0) (void)hipEventDestroy(e0);
    if (e1) (void)hipEventDestroy(e1);
    (void)hipFree(dt); (void)hipFree(dx);
    if (!ok) return ECFFT_ERR_HIP;
    *mul_per_s = (double)n * iters / (best * 1e-3);
    return ECFFT_OK;
}

namespace {
template <class F>
int run_sharded(ecfft_ctx* c, DeviceChain<F>& ch, ecfft_comm* comm, Op op, const void* in, void* out, size_t len, int moiety, void* stream,
                int in_layout = ECFFT_LAYOUT_BLOCK, int out_layout = ECFFT_LAYOUT_BLOCK) {
    using E = typename F::elem;
    if (!in || !out || !comm || !comm->t) return ECFFT_ERR_BAD_ARG;
    if (!is_pow2(len)) return ECFFT_ERR_NOT_POW2;
    Transport& tr = *comm->t;
    const size_t P = (size_t)tr.world;
    if (!is_pow2(P) || P > 64) return ECFFT_ERR_BAD_ARG;
    size_t need_tree = (op == OP_EXTEND) ? len * 2 : len;
    if (need_tree > ch.size()) return ECFFT_ERR_TREE_TOO_SMALL;
    if (op == OP_EXTEND && moiety != ECFFT_S0 && moiety != ECFFT_S1) return ECFFT_ERR_BAD_ARG;
    if (len / P < 2 * P) return ECFFT_ERR_BAD_ARG;                       // every rank needs at least 2P elements
    if (ch.shard_mode()) {                                               // a shard context serves exactly the split it was built for
        const bool fits = P == ((size_t)1 << ch.shard_log_p()) && (unsigned)tr.rank == ch.shard_rank() &&
                          ((ch.shard_kind() == DeviceChain<F>::kShardExtend && op == OP_EXTEND && 2 * len == ch.size()) ||
                           (ch.shard_kind() == DeviceChain<F>::kShardEnter && op == OP_ENTER && len == ch.size()) ||
                           (ch.shard_kind() == DeviceChain<F>::kShardExit && op == OP_EXIT && len == ch.size()));
        if (!fits) return ECFFT_ERR_BAD_ARG;
    }
    if ((in_layout != ECFFT_LAYOUT_BLOCK && in_layout != ECFFT_LAYOUT_CYCLIC) || (out_layout != ECFFT_LAYOUT_BLOCK && out_layout != ECFFT_LAYOUT_CYCLIC)) return ECFFT_ERR_BAD_ARG;
    hipStream_t s = (hipStream_t)stream;
    DeviceGuard dev(c->device);
    if (!dev.ok) return ECFFT_ERR_HIP;
    std::lock_guard<std::mutex> guard(ch.lock());
    OpScope scope(c, s);
    if (!scope.ok) return ECFFT_ERR_HIP;
    bool ok = false;
    switch (op) {
        case OP_EXTEND: ok = ch.api_extend_split(tr, (const E*)in, (E*)out, len, moiety, s, in_layout == ECFFT_LAYOUT_CYCLIC, out_layout == ECFFT_LAYOUT_CYCLIC); break;
        case OP_ENTER: ok = ch.api_enter_split(tr, (const E*)in, (E*)out, len, s); break;
        case OP_EXIT: ok = ch.api_exit_split(tr, (const E*)in, (E*)out, len, s); break;
    }
    return ok && hipGetLastError() == hipSuccess ? ECFFT_OK : ECFFT_ERR_HIP;
}
}  // namespace

// ---- FFTree wire format (ark-serialize 0.4 conventions; hand-written impls at /root/reference/src/fftree.rs:510-660) ----------
// Vec<T> = u64 little-endian length + elements; a field element = its STANDARD-form integer, little endian (32 / 4 bytes);
// [T; N] = elements only; bool = one byte; DensePolynomial = coefficient Vec with trailing zeros trimmed.  Field order of one
// tree (:528-548): f, recombine_matrices, decompose_matrices, rational_maps, xnn_s, z0_s1, z1_s0, [xnn_s_inv, z0_inv_s1,
// z1_inv_s0 with Compress::No only], z0z0_rem_xnn_s, z1z1_rem_xnn_s, bool has_subtree, then the subtree.
// The device tables are plain residues, so every table goes to the file as it lies in HBM.
namespace {
template <class F>
size_t trimmed_len(const typename F::elem* c3) {
    size_t k = 3;
    while (k > 0 && F::is_zero(c3[k - 1])) --k;
    return k;
}
template <class F>
size_t wire_size(const DeviceChain<F>& ch, int compress) {
    const size_t eb = sizeof(typename F::elem);
    size_t total = 0;
    for (size_t m = ch.size();; m >>= 1) {
        const unsigned lm = ilog2(m); const size_t e = m / 2;
        total += 8 + 2 * m * eb + 2 * (8 + 4 * m * eb) + 8;
        for (unsigned k = 0; k < lm; ++k) total += 16 + (trimmed_len<F>(ch.host().maps[k].num) + trimmed_len<F>(ch.host().maps[k].den)) * eb;
        total += ((8 + m * eb) + 2 * (8 + e * eb)) * (compress ? 1 : 2);
        total += 2 * (8 + (m > 1 ? m : 0) * eb) + 1;
        if (m == 1) break;
    }
    return total;
}
inline void put_u64(uint8_t*& p, uint64_t v) { for (int i = 0; i < 8; ++i) *p++ = (uint8_t)(v >> (8 * i)); }
template <class F>
int wire_write(DeviceChain<F>& ch, int compress, uint8_t* buf) {
    using E = typename F::elem;
    const size_t eb = sizeof(E);
    uint8_t* p = buf;
    auto vec = [&](size_t m, int which, size_t per_entry) -> int {      // Vec of cnt / per_entry entries
        size_t cnt = 0;
        int rc = table_of(ch, m, which, nullptr, 0, &cnt, true);
        if (rc != ECFFT_OK) return rc;
        put_u64(p, cnt / per_entry);
        if (cnt) { rc = table_of(ch, m, which, p, cnt, nullptr, true); if (rc != ECFFT_OK) return rc; }
        p += cnt * eb;
        return ECFFT_OK;
    };
    for (size_t m = ch.size();; m >>= 1) {
        const unsigned lm = ilog2(m);
        int rc;
        if ((rc = vec(m, ECFFT_TBL_F, 1)) || (rc = vec(m, ECFFT_TBL_RECOMBINE, 4)) || (rc = vec(m, ECFFT_TBL_DECOMPOSE, 4))) return rc;
        put_u64(p, lm);                                                   // the subtree keeps the first log2(m) maps (split_last, :480)
        for (unsigned k = 0; k < lm; ++k) {
            const RatMap<F>& mp = ch.host().maps[k];
            const size_t ln = trimmed_len<F>(mp.num), ld = trimmed_len<F>(mp.den);
            put_u64(p, ln); memcpy(p, mp.num, ln * eb); p += ln * eb;
            put_u64(p, ld); memcpy(p, mp.den, ld * eb); p += ld * eb;
        }
        if ((rc = vec(m, ECFFT_TBL_XNN_S, 1)) || (rc = vec(m, ECFFT_TBL_Z0_S1, 1)) || (rc = vec(m, ECFFT_TBL_Z1_S0, 1))) return rc;
        if (!compress && ((rc = vec(m, ECFFT_TBL_XNN_S_INV, 1)) || (rc = vec(m, ECFFT_TBL_Z0_INV_S1, 1)) || (rc = vec(m, ECFFT_TBL_Z1_INV_S0, 1)))) return rc;
        if ((rc = vec(m, ECFFT_TBL_Z0Z0_REM_XNN_S, 1)) || (rc = vec(m, ECFFT_TBL_Z1Z1_REM_XNN_S, 1))) return rc;
        *p++ = m > 1 ? 1 : 0;
        if (m == 1) break;
    }
    return ECFFT_OK;
}

// (the bounds-checked parse itself — cursor, canonicality check, level structure — lives in wire_parse.h: pure host C++ that
// tests/cpp/wire_fuzz.cpp also drives under AddressSanitizer + UBSan without a GPU)
template <class F>
int wire_read(int field, const uint8_t* data, size_t len, int compress, int device, int verify, ecfft_ctx** out,
              std::unique_ptr<DeviceChain<F>> ecfft_ctx::*slot) {
    using E = typename F::elem;
    const size_t eb = sizeof(E);
    static_assert(sizeof(E) == 32 || sizeof(E) == 4, "element sizes of the two fields");
    wire::File file;
    { const int prc = wire::parse(field, data, len, compress, file); if (prc != ECFFT_OK) return prc; }
    const std::vector<wire::Level>& levels = file.levels;
    typedef wire::Level WireLevel;
    HostTree<F> ht;
    for (const wire::Map& m : file.maps) {
        RatMap<F> mp;
        for (int i = 0; i < 3; ++i) { memcpy(&mp.num[i], m.num[i], eb); memcpy(&mp.den[i], m.den[i], eb); }
        ht.maps.push_back(mp);
    }
    // FFTree::new on the file's leaves and maps: every other table is recomputed on the GPU (the reference USES the file's
    // tables; with verify != 0 each of them is compared with the recomputed one, so a file whose tables disagree with its own
    // point set is rejected instead of being silently repaired).  Compress::Yes files carry no inverse tables (:620-628).
    const size_t n = levels[0].n;
    ht.n = n; ht.f.assign(2 * n, F::zero());
    memcpy(ht.f.data() + n, levels[0].tbl[ECFFT_TBL_F] + n * eb, n * eb);
    ht.leaves_only = true;
    std::unique_ptr<ecfft_ctx> c(new (std::nothrow) ecfft_ctx());
    if (!c) return ECFFT_ERR_HIP;
    c->field = field; c->device = device;
    int rc = guarded([&] { return finish_build(std::move(ht), device, (*c).*slot); });
    if (rc != ECFFT_OK) return rc;
    {   // verify == 0 still checks the internal layers of every `f` (they follow from the leaves and the maps: cheap, and a file
        // whose layers disagree with its maps would otherwise load as a different tree than the reference's deserialize builds)
        DeviceGuard dev(device);
        if (!dev.ok) return ECFFT_ERR_HIP;
        DeviceChain<F>& ch = *((*c).*slot);
        std::vector<E> got;
        for (const WireLevel& lv : levels)
            for (int which = 0; which < 11; ++which) {
                if (!verify && which != ECFFT_TBL_F) continue;
                if (!lv.tbl[which] && lv.cnt[which] == 0) continue;
                got.resize(lv.cnt[which] ? lv.cnt[which] : 1);
                rc = guarded([&] { return table_of(ch, lv.n, which, got.data(), lv.cnt[which], nullptr, true); });
                if (rc != ECFFT_OK) return rc;
                const size_t skip = which == ECFFT_TBL_F ? 1 : 0;           // heap index 0 is unused (src/utils.rs:228-252)
                if (verify && which == ECFFT_TBL_F && lv.cnt[which] > 0) {  // ... and zero in every tree the crate builds (src/fftree.rs:50, 471): a verified
                    bool zero0 = true;                                      // file is byte for byte what serialize writes (tests: mutated files)
                    for (size_t b = 0; b < eb; ++b) zero0 = zero0 && lv.tbl[which][b] == 0;
                    if (!zero0) { fprintf(stderr, "ecfft: entry 0 of f of the %zu-leaf subtree in the file is not zero\n", lv.n); return ECFFT_ERR_BAD_ARG; }
                }
                if (lv.cnt[which] > skip && memcmp((const uint8_t*)got.data() + skip * eb, lv.tbl[which] + skip * eb, (lv.cnt[which] - skip) * eb) != 0) {
                    fprintf(stderr, "ecfft: table %d of the %zu-leaf subtree in the file differs from the one rebuilt from its point set\n", which, lv.n);
                    return ECFFT_ERR_BAD_ARG;
                }
            }
    }
    *out = c.release();
    return ECFFT_OK;
}
}  // namespace

extern "C" {

size_t ecfft_elem_size(int field) { return field == ECFFT_FIELD_SECP256K1 ? 32 : (field == ECFFT_FIELD_M31 ? 4 : 0); }

int ecfft_build_fftree(int field, size_t n, int device, ecfft_ctx** out) {
    if (!out) return ECFFT_ERR_BAD_ARG;
    *out = nullptr;
    if (!is_pow2(n)) return ECFFT_ERR_NOT_POW2;                           // assert!(n.is_power_of_two())
    if (field != ECFFT_FIELD_SECP256K1 && field != ECFFT_FIELD_M31) return ECFFT_ERR_BAD_ARG;
    unsigned log_n = ilog2(n);
    // size limits first (they do not need a device): src/lib.rs:62-64, src/ec.rs:510-515
    if (field == ECFFT_FIELD_SECP256K1 && log_n >= 36) return ECFFT_ERR_TREE_TOO_LARGE;
    if (field == ECFFT_FIELD_M31 && log_n > 28) return ECFFT_ERR_TREE_TOO_LARGE;
    if (!have_device(device)) return ECFFT_ERR_HIP;
    std::unique_ptr<ecfft_ctx> c(new (std::nothrow) ecfft_ctx());
    if (!c) return ECFFT_ERR_HIP;
    c->field = field; c->device = device;
    int rc;
    if (field == ECFFT_FIELD_SECP256K1) {
        HostTree<Secp256k1> ht;
        int r = build_host_tree<Secp256k1>(log_n, ht, /*points=*/false);   // leaves and layers are computed on the GPU (points_on_device)
        if (r == 1) return ECFFT_ERR_TREE_TOO_LARGE;
        if (r) return ECFFT_ERR_BAD_ARG;
        rc = guarded([&] { return finish_build(std::move(ht), device, c->secp); });
    } else {
        HostTree<M31> ht;
        int r = build_host_tree<M31>(log_n, ht, /*points=*/false);   // leaves and layers are computed on the GPU (points_on_device)
        if (r == 1) return ECFFT_ERR_TREE_TOO_LARGE;
        if (r) return ECFFT_ERR_BAD_ARG;
        rc = guarded([&] { return finish_build(std::move(ht), device, c->m31); });
    }
    if (rc != ECFFT_OK) return rc;
    *out = c.release();
    return ECFFT_OK;
}

namespace {
// kind 1: EXTEND-only shard context for e = len evaluations (tree T_2e); kind 2: ENTER-only for n = len coefficients (tree T_n)
int build_shard_ctx(int kind, int field, size_t len, int device, int world, int rank, ecfft_ctx** out, ecfft_comm* comm = nullptr, int flags = 0) {
    if (!out) return ECFFT_ERR_BAD_ARG;
    *out = nullptr;
    if (!is_pow2(len) || !is_pow2((size_t)(world > 0 ? world : 0))) return ECFFT_ERR_NOT_POW2;
    if (field != ECFFT_FIELD_SECP256K1 && field != ECFFT_FIELD_M31) return ECFFT_ERR_BAD_ARG;
    if (world > 64 || rank < 0 || rank >= world || (kind >= 2 && world < 2)) return ECFFT_ERR_BAD_ARG;
    if (kind == 3 && (!comm || !comm->t)) return ECFFT_ERR_BAD_ARG;
    if (len / (size_t)world < 2 * (size_t)world) return ECFFT_ERR_BAD_ARG;   // same bound as the sharded transforms
    const unsigned log_n = ilog2(len) + (kind == 1 ? 1 : 0), log_p = ilog2((size_t)world);
    if (field == ECFFT_FIELD_SECP256K1 && log_n >= 36) return ECFFT_ERR_TREE_TOO_LARGE;
    if (field == ECFFT_FIELD_M31 && log_n > 28) return ECFFT_ERR_TREE_TOO_LARGE;
    if (!have_device(device)) return ECFFT_ERR_HIP;
    std::unique_ptr<ecfft_ctx> c(new (std::nothrow) ecfft_ctx());
    if (!c) return ECFFT_ERR_HIP;
    c->field = field; c->device = device;
    auto finish = [&](auto&& ht, auto& slot) {
        using Chain = typename std::remove_reference<decltype(*slot)>::type;
        slot.reset(new (std::nothrow) Chain());
        if (!slot) return (int)ECFFT_ERR_HIP;
        DeviceGuard dev(device);
        if (!dev.ok) return (int)ECFFT_ERR_HIP;
        const bool ok = kind == 1 ? slot->build_extend_shard(std::move(ht), device, log_p, (unsigned)rank)
                      : kind == 2 ? slot->build_enter_shard(std::move(ht), device, log_p, (unsigned)rank)
                                  : slot->build_exit_shard(std::move(ht), device, *comm->t, (flags & ECFFT_EXIT_SHARD_MIN_MEMORY) != 0);
        return ok ? (int)ECFFT_OK : (int)ECFFT_ERR_HIP;
    };
    int rc;
    if (field == ECFFT_FIELD_SECP256K1) {
        HostTree<Secp256k1> ht;
        int r = build_host_tree<Secp256k1>(log_n, ht, /*points=*/false);   // leaves and layers are computed on the GPU (points_on_device)
        if (r == 1) return ECFFT_ERR_TREE_TOO_LARGE;
        if (r) return ECFFT_ERR_BAD_ARG;
        rc = guarded([&] { return finish(std::move(ht), c->secp); });
    } else {
        HostTree<M31> ht;
        int r = build_host_tree<M31>(log_n, ht, /*points=*/false);   // leaves and layers are computed on the GPU (points_on_device)
        if (r == 1) return ECFFT_ERR_TREE_TOO_LARGE;
        if (r) return ECFFT_ERR_BAD_ARG;
        rc = guarded([&] { return finish(std::move(ht), c->m31); });
    }
    if (rc != ECFFT_OK) return rc;
    *out = c.release();
    return ECFFT_OK;
}
}  // namespace
int ecfft_build_extend_shard(int field, size_t e, int device, int world, int rank, ecfft_ctx** out) { return build_shard_ctx(1, field, e, device, world, rank, out); }
int ecfft_build_enter_shard(int field, size_t n, int device, int world, int rank, ecfft_ctx** out) { return build_shard_ctx(2, field, n, device, world, rank, out); }
int ecfft_build_exit_shard_opts(int field, size_t n, int device, ecfft_comm* comm, int flags, ecfft_ctx** out) {
    if (!comm || !comm->t || (flags & ~ECFFT_EXIT_SHARD_MIN_MEMORY)) { if (out) *out = nullptr; return ECFFT_ERR_BAD_ARG; }
    return build_shard_ctx(3, field, n, device, comm->t->world, comm->t->rank, out, comm, flags);
}
int ecfft_build_exit_shard(int field, size_t n, int device, ecfft_comm* comm, ecfft_ctx** out) {
    return ecfft_build_exit_shard_opts(field, n, device, comm, 0, out);
}

int ecfft_fftree_new(int field, const void* leaves, size_t n, const void* map_num3, const void* map_den3, int device,
                     ecfft_ctx** out) {
    if (!out) return ECFFT_ERR_BAD_ARG;
    *out = nullptr;
    if (!leaves || (n > 1 && (!map_num3 || !map_den3))) return ECFFT_ERR_BAD_ARG;
    if (!is_pow2(n)) return ECFFT_ERR_NOT_POW2;
    if (field != ECFFT_FIELD_SECP256K1 && field != ECFFT_FIELD_M31) return ECFFT_ERR_BAD_ARG;
    if (!have_device(device)) return ECFFT_ERR_HIP;
    unsigned log_n = ilog2(n);
    std::unique_ptr<ecfft_ctx> c(new (std::nothrow) ecfft_ctx());
    if (!c) return ECFFT_ERR_HIP;
    c->field = field; c->device = device;
    int rc;
    if (field == ECFFT_FIELD_SECP256K1) {
        HostTree<Secp256k1> ht; ht.n = n; ht.f.assign(2 * n, Secp256k1::zero()); ht.maps.resize(log_n);
        memcpy(ht.f.data() + n, leaves, n * 32);
        secp_from_mont_host(ht.f.data() + n, n);
        for (unsigned k = 0; k < log_n; ++k) {
            memcpy(ht.maps[k].num, (const char*)map_num3 + 96 * k, 96); memcpy(ht.maps[k].den, (const char*)map_den3 + 96 * k, 96);
            secp_from_mont_host(ht.maps[k].num, 3); secp_from_mont_host(ht.maps[k].den, 3);
            if (!Secp256k1::is_zero(ht.maps[k].den[2])) return ECFFT_ERR_BAD_ARG;   // x-map denominators have degree 1
        }
        ht.leaves_only = true;                                     // the layers psi_k(L_k) are computed on the GPU (points_on_device)
        rc = guarded([&] { return finish_build(std::move(ht), device, c->secp); });
    } else {
        HostTree<M31> ht; ht.n = n; ht.f.assign(2 * n, 0); ht.maps.resize(log_n);
        memcpy(ht.f.data() + n, leaves, n * 4);
        for (unsigned k = 0; k < log_n; ++k) {
            memcpy(ht.maps[k].num, (const char*)map_num3 + 12 * k, 12); memcpy(ht.maps[k].den, (const char*)map_den3 + 12 * k, 12);
            if (ht.maps[k].den[2] != 0) return ECFFT_ERR_BAD_ARG;
        }
        ht.leaves_only = true;
        rc = guarded([&] { return finish_build(std::move(ht), device, c->m31); });
    }
    if (rc != ECFFT_OK) return rc;
    *out = c.release();
    return ECFFT_OK;
}

int ecfft_build_points(int field, size_t n, void* f_out, void* map_num3_out, void* map_den3_out) {
    if (!f_out) return ECFFT_ERR_BAD_ARG;
    if (!is_pow2(n)) return ECFFT_ERR_NOT_POW2;
    unsigned log_n = ilog2(n);
    if (field == ECFFT_FIELD_SECP256K1) {
        HostTree<Secp256k1> ht;
        int r = build_host_tree<Secp256k1>(log_n, ht);
        if (r == 1) return ECFFT_ERR_TREE_TOO_LARGE;
        if (r) return ECFFT_ERR_BAD_ARG;
        secp_to_mont_host(ht.f.data(), 2 * n);
        memcpy(f_out, ht.f.data(), 2 * n * 32);
        for (unsigned k = 0; k < log_n; ++k) {
            secp_to_mont_host(ht.maps[k].num, 3); secp_to_mont_host(ht.maps[k].den, 3);
            if (map_num3_out) memcpy((char*)map_num3_out + 96 * k, ht.maps[k].num, 96);
            if (map_den3_out) memcpy((char*)map_den3_out + 96 * k, ht.maps[k].den, 96);
        }
        return ECFFT_OK;
    }
    if (field == ECFFT_FIELD_M31) {
        HostTree<M31> ht;
        int r = build_host_tree<M31>(log_n, ht);
        if (r == 1) return ECFFT_ERR_TREE_TOO_LARGE;
        if (r) return ECFFT_ERR_BAD_ARG;
        memcpy(f_out, ht.f.data(), 2 * n * 4);
        for (unsigned k = 0; k < log_n; ++k) {
            if (map_num3_out) memcpy((char*)map_num3_out + 12 * k, ht.maps[k].num, 12);
            if (map_den3_out) memcpy((char*)map_den3_out + 12 * k, ht.maps[k].den, 12);
        }
        return ECFFT_OK;
    }
    return ECFFT_ERR_BAD_ARG;
}

void ecfft_ctx_destroy(ecfft_ctx* ctx) {
    if (!ctx) return;
    DeviceGuard dev(ctx->device);
    delete ctx;
}

size_t ecfft_tree_size(const ecfft_ctx* ctx) {
    if (!ctx) return 0;
    return ctx->field == ECFFT_FIELD_SECP256K1 ? ctx->secp->size() : ctx->m31->size();
}
int ecfft_field(const ecfft_ctx* ctx) { return ctx ? ctx->field : -1; }
#ifdef ECFFT_TEST_HOOKS
long ecfft_selfcheck_pointwise_z(ecfft_ctx* ctx, size_t m) {
    if (!ctx || !is_pow2(m)) return -1;
    return guarded([&] {
        DeviceGuard dev(ctx->device);
        if (!dev.ok) return -1;
        std::lock_guard<std::mutex> guard(ctx->field == ECFFT_FIELD_SECP256K1 ? ctx->secp->lock() : ctx->m31->lock());
        OpScope scope(ctx, nullptr);                              // its pooled temporaries may still be in use by the previous asynchronous call
        if (!scope.ok) return -1;
        return (int)(ctx->field == ECFFT_FIELD_SECP256K1 ? ctx->secp->selfcheck_pointwise_z(m) : ctx->m31->selfcheck_pointwise_z(m));
    });
}
int ecfft_test_fail_next_collective(ecfft_ctx* ctx) {
    if (!ctx) return ECFFT_ERR_BAD_ARG;
    if (ctx->field == ECFFT_FIELD_SECP256K1) ctx->secp->test_fail_next_collective(); else ctx->m31->test_fail_next_collective();
    return ECFFT_OK;
}
int ecfft_test_fail_build_rank(int rank) {
    DeviceChain<Secp256k1>::test_fail_build_rank().store(rank);
    DeviceChain<M31>::test_fail_build_rank().store(rank);
    return ECFFT_OK;
}
#endif  // ECFFT_TEST_HOOKS
int ecfft_ctx_trim(ecfft_ctx* ctx) {
    if (!ctx) return ECFFT_ERR_BAD_ARG;
    DeviceGuard dev(ctx->device);
    if (!dev.ok) return ECFFT_ERR_HIP;
    // the staging buffer is used under the chain lock by every host-memory call: it is freed under that lock too
    auto free_stage = [ctx] { if (ctx->stage) { (void)hipFree(ctx->stage); ctx->stage = nullptr; ctx->stage_bytes = 0; } };
    if (ctx->field == ECFFT_FIELD_SECP256K1) ctx->secp->trim(free_stage); else ctx->m31->trim(free_stage);
    return ECFFT_OK;
}
size_t ecfft_ctx_device_bytes(const ecfft_ctx* ctx) {
    if (!ctx) return 0;
    return ctx->field == ECFFT_FIELD_SECP256K1 ? ctx->secp->device_bytes() : ctx->m31->device_bytes();
}

int ecfft_enter(ecfft_ctx* ctx, const void* coeffs, void* evals, size_t n, int mem, void* stream) {
    if (!ctx || shard_only(ctx)) return ECFFT_ERR_BAD_ARG;
    return guarded([&] { return ctx->field == ECFFT_FIELD_SECP256K1 ? run_op(ctx, *ctx->secp, OP_ENTER, coeffs, evals, n, 1, 0, mem, stream)
                                               : run_op(ctx, *ctx->m31, OP_ENTER, coeffs, evals, n, 1, 0, mem, stream); });
}
int ecfft_exit(ecfft_ctx* ctx, const void* evals, void* coeffs, size_t n, int mem, void* stream) {
    if (!ctx || shard_only(ctx)) return ECFFT_ERR_BAD_ARG;
    return guarded([&] { return ctx->field == ECFFT_FIELD_SECP256K1 ? run_op(ctx, *ctx->secp, OP_EXIT, evals, coeffs, n, 1, 0, mem, stream)
                                               : run_op(ctx, *ctx->m31, OP_EXIT, evals, coeffs, n, 1, 0, mem, stream); });
}
int ecfft_enter_many(ecfft_ctx* ctx, const void* coeffs, void* evals, size_t n, size_t count, int mem, void* stream) {
    if (!ctx || shard_only(ctx)) return ECFFT_ERR_BAD_ARG;
    return guarded([&] { return ctx->field == ECFFT_FIELD_SECP256K1 ? run_op(ctx, *ctx->secp, OP_ENTER, coeffs, evals, n, count, 0, mem, stream)
                                               : run_op(ctx, *ctx->m31, OP_ENTER, coeffs, evals, n, count, 0, mem, stream); });
}
int ecfft_exit_many(ecfft_ctx* ctx, const void* evals, void* coeffs, size_t n, size_t count, int mem, void* stream) {
    if (!ctx || shard_only(ctx)) return ECFFT_ERR_BAD_ARG;
    return guarded([&] { return ctx->field == ECFFT_FIELD_SECP256K1 ? run_op(ctx, *ctx->secp, OP_EXIT, evals, coeffs, n, count, 0, mem, stream)
                                               : run_op(ctx, *ctx->m31, OP_EXIT, evals, coeffs, n, count, 0, mem, stream); });
}
int ecfft_extend(ecfft_ctx* ctx, const void* in, void* out, size_t e, int moiety, size_t count, int mem, void* stream) {
    if (!ctx || shard_only(ctx)) return ECFFT_ERR_BAD_ARG;
    return guarded([&] { return ctx->field == ECFFT_FIELD_SECP256K1 ? run_op(ctx, *ctx->secp, OP_EXTEND, in, out, e, count, moiety, mem, stream)
                                               : run_op(ctx, *ctx->m31, OP_EXTEND, in, out, e, count, moiety, mem, stream); });
}

int ecfft_extend_top_cyclic(ecfft_ctx* ctx, void* buf, size_t e, int moiety, unsigned log_p, unsigned rank, int recombine, int mem, void* stream) {
    if (!ctx || shard_only(ctx)) return ECFFT_ERR_BAD_ARG;
    return guarded([&] { return ctx->field == ECFFT_FIELD_SECP256K1 ? run_shard(ctx, *ctx->secp, buf, e, moiety, log_p, rank, recombine ? 1 : 0, mem, stream)
                                               : run_shard(ctx, *ctx->m31, buf, e, moiety, log_p, rank, recombine ? 1 : 0, mem, stream); });
}
int ecfft_extend_local_block(ecfft_ctx* ctx, void* buf, size_t e, int moiety, unsigned log_p, int mem, void* stream) {
    if (!ctx || shard_only(ctx)) return ECFFT_ERR_BAD_ARG;
    return guarded([&] { return ctx->field == ECFFT_FIELD_SECP256K1 ? run_shard(ctx, *ctx->secp, buf, e, moiety, log_p, 0, 2, mem, stream)
                                               : run_shard(ctx, *ctx->m31, buf, e, moiety, log_p, 0, 2, mem, stream); });
}

int ecfft_mextend(ecfft_ctx* ctx, const void* in, void* out, size_t e, int moiety, size_t count, int mem, void* stream) {
    if (!ctx || shard_only(ctx)) return ECFFT_ERR_BAD_ARG;
    return guarded([&] { return ECFFT_DISPATCH_ALG(ALG_MEXTEND, in, nullptr, nullptr, out, e, count, moiety, mem, stream, nullptr); });
}
int ecfft_redc(ecfft_ctx* ctx, const void* evals, const void* a, void* out, size_t n, int moiety, int mem, void* stream) {
    if (!ctx || shard_only(ctx)) return ECFFT_ERR_BAD_ARG;
    return guarded([&] { return ECFFT_DISPATCH_ALG(ALG_REDC, evals, a, nullptr, out, n, 1, moiety, mem, stream, nullptr); });
}
int ecfft_modular_reduce(ecfft_ctx* ctx, const void* evals, const void* a, const void* c, void* out, size_t n, int mem, void* stream) {
    if (!ctx || shard_only(ctx)) return ECFFT_ERR_BAD_ARG;
    return guarded([&] { return ECFFT_DISPATCH_ALG(ALG_MOD, evals, a, c, out, n, 1, 0, mem, stream, nullptr); });
}
int ecfft_vanish(ecfft_ctx* ctx, const void* domain, void* out, size_t nd, int mem, void* stream) {
    if (!ctx || shard_only(ctx)) return ECFFT_ERR_BAD_ARG;
    return guarded([&] { return ECFFT_DISPATCH_ALG(ALG_VANISH, domain, nullptr, nullptr, out, nd, 1, 0, mem, stream, nullptr); });
}
int ecfft_degree(ecfft_ctx* ctx, const void* evals, size_t n, int mem, void* stream, size_t* degree) {
    if (!ctx || shard_only(ctx)) return ECFFT_ERR_BAD_ARG;
    return guarded([&] { return ECFFT_DISPATCH_ALG(ALG_DEGREE, evals, nullptr, nullptr, nullptr, n, 1, 0, mem, stream, degree); });
}

// ---- one transform split over several GPUs -------------------------------------------------------------------------
int ecfft_comm_get_unique_id(void* id_out) {
    if (!id_out) return ECFFT_ERR_BAD_ARG;
    RcclApi& api = RcclApi::get();
    if (!api.ok()) return ECFFT_ERR_HIP;
    RcclApi::UniqueId id;
    if (api.GetUniqueId(&id) != 0) return ECFFT_ERR_HIP;
    memcpy(id_out, id.internal, ECFFT_COMM_ID_BYTES);
    return ECFFT_OK;
}
int ecfft_comm_set_link_striping(ecfft_comm* comm, size_t min_gain_bytes) {
    if (!comm || !comm->t) return ECFFT_ERR_BAD_ARG;
    if (comm->t->used()) return ECFFT_ERR_BAD_ARG;      // frozen once the communicator has carried an exchange: the ranks agree on it in their first vote
    comm->t->stripe_min_gain = min_gain_bytes;
    return ECFFT_OK;
}
int ecfft_comm_set_rccl_library(const char* path) {
    return RcclApi::set_library(path) ? ECFFT_OK : ECFFT_ERR_BAD_ARG;
}
int ecfft_comm_init_rank(const void* id, int world, int rank, int device, ecfft_comm** out) {
    if (!out) return ECFFT_ERR_BAD_ARG;
    *out = nullptr;
    if (!id || world < 1 || rank < 0 || rank >= world) return ECFFT_ERR_BAD_ARG;
    if (!have_device(device)) return ECFFT_ERR_HIP;
    DeviceGuard dev(device);
    if (!dev.ok) return ECFFT_ERR_HIP;
    std::unique_ptr<RcclTransport> t(new (std::nothrow) RcclTransport());
    if (!t || !t->init(id, world, rank, device)) return ECFFT_ERR_HIP;
    ecfft_comm* c = new (std::nothrow) ecfft_comm();
    if (!c) return ECFFT_ERR_HIP;
    c->t = t.release();
    *out = c;
    return ECFFT_OK;
}
int ecfft_comm_init_callback(int world, int rank, int device, ecfft_exchange_fn fn, void* user, ecfft_comm** out) {
    if (!out) return ECFFT_ERR_BAD_ARG;
    *out = nullptr;
    if (!fn || world < 1 || rank < 0 || rank >= world) return ECFFT_ERR_BAD_ARG;
    ecfft_comm* c = new (std::nothrow) ecfft_comm();
    if (!c) return ECFFT_ERR_HIP;
    c->t = new (std::nothrow) CallbackTransport(world, rank, device, fn, user);
    if (!c->t) { delete c; return ECFFT_ERR_HIP; }
    *out = c;
    return ECFFT_OK;
}
#ifdef ECFFT_TEST_HOOKS
int ecfft_comm_init_projection(int world, int rank, int device, double delay_us, double link_gbps, ecfft_comm** out) {
    if (!out) return ECFFT_ERR_BAD_ARG;
    *out = nullptr;
    if (world < 1 || rank < 0 || rank >= world || delay_us < 0 || link_gbps < 0) return ECFFT_ERR_BAD_ARG;
    ecfft_comm* c = new (std::nothrow) ecfft_comm();
    if (!c) return ECFFT_ERR_HIP;
    c->t = new (std::nothrow) ProjectionTransport(world, rank, device, delay_us, link_gbps);
    if (!c->t) { delete c; return ECFFT_ERR_HIP; }
    *out = c;
    return ECFFT_OK;
}
#endif  // ECFFT_TEST_HOOKS
void ecfft_comm_destroy(ecfft_comm* comm) {
    if (!comm) return;
    DeviceGuard dev(comm->t ? comm->t->device : 0);
    delete comm;
}
int ecfft_comm_abort(ecfft_comm* comm) {
    if (!comm || !comm->t) return ECFFT_ERR_BAD_ARG;
    return comm->t->abort() ? ECFFT_OK : ECFFT_ERR_HIP;            // callable from another host thread than the one that is blocked
}
int ecfft_comm_rank(const ecfft_comm* comm) { return comm && comm->t ? comm->t->rank : -1; }
int ecfft_comm_world(const ecfft_comm* comm) { return comm && comm->t ? comm->t->world : 0; }
int ecfft_comm_stats_enable(ecfft_comm* comm, int on) {
    if (!comm || !comm->t) return ECFFT_ERR_BAD_ARG;
    DeviceGuard dev(comm->t->device);
    if (!dev.ok || hipDeviceSynchronize() != hipSuccess) return ECFFT_ERR_HIP;
    comm->t->stats_enable(on != 0);
    return ECFFT_OK;
}
int ecfft_comm_stats_read(ecfft_comm* comm, double* comm_ms, double* exchanges, double* bytes_sent) {
    if (!comm || !comm->t) return ECFFT_ERR_BAD_ARG;
    DeviceGuard dev(comm->t->device);
    if (!dev.ok || hipDeviceSynchronize() != hipSuccess) return ECFFT_ERR_HIP;
    comm->t->stats_read(comm_ms, exchanges, bytes_sent);
    comm->t->stats_reset();
    return ECFFT_OK;
}


int ecfft_extend_sharded(ecfft_ctx* ctx, ecfft_comm* comm, const void* in, void* out, size_t e, int moiety, void* stream) {
    if (!ctx) return ECFFT_ERR_BAD_ARG;
    return guarded([&] { return ctx->field == ECFFT_FIELD_SECP256K1 ? run_sharded(ctx, *ctx->secp, comm, OP_EXTEND, in, out, e, moiety, stream)
                                                                    : run_sharded(ctx, *ctx->m31, comm, OP_EXTEND, in, out, e, moiety, stream); });
}
int ecfft_extend_sharded_layout(ecfft_ctx* ctx, ecfft_comm* comm, const void* in, void* out, size_t e, int moiety, int in_layout, int out_layout, void* stream) {
    if (!ctx) return ECFFT_ERR_BAD_ARG;
    return guarded([&] { return ctx->field == ECFFT_FIELD_SECP256K1 ? run_sharded(ctx, *ctx->secp, comm, OP_EXTEND, in, out, e, moiety, stream, in_layout, out_layout)
                                                                    : run_sharded(ctx, *ctx->m31, comm, OP_EXTEND, in, out, e, moiety, stream, in_layout, out_layout); });
}
int ecfft_enter_sharded(ecfft_ctx* ctx, ecfft_comm* comm, const void* coeffs, void* evals, size_t n, void* stream) {
    if (!ctx) return ECFFT_ERR_BAD_ARG;
    return guarded([&] { return ctx->field == ECFFT_FIELD_SECP256K1 ? run_sharded(ctx, *ctx->secp, comm, OP_ENTER, coeffs, evals, n, 0, stream)
                                                                    : run_sharded(ctx, *ctx->m31, comm, OP_ENTER, coeffs, evals, n, 0, stream); });
}
int ecfft_exit_sharded(ecfft_ctx* ctx, ecfft_comm* comm, const void* evals, void* coeffs, size_t n, void* stream) {
    if (!ctx) return ECFFT_ERR_BAD_ARG;
    return guarded([&] { return ctx->field == ECFFT_FIELD_SECP256K1 ? run_sharded(ctx, *ctx->secp, comm, OP_EXIT, evals, coeffs, n, 0, stream)
                                                                    : run_sharded(ctx, *ctx->m31, comm, OP_EXIT, evals, coeffs, n, 0, stream); });
}

int ecfft_table_fma(ecfft_ctx* ctx, void* out, const void* x, const void* y, size_t cnt, size_t m, int which, size_t t_off,
                    size_t t_stride, int mode, int mem, void* stream) {
    if (!ctx || shard_only(ctx)) return ECFFT_ERR_BAD_ARG;
    return guarded([&] { return ctx->field == ECFFT_FIELD_SECP256K1 ? run_table_fma(ctx, *ctx->secp, out, x, y, cnt, m, which, t_off, t_stride, mode, mem, stream)
                                               : run_table_fma(ctx, *ctx->m31, out, x, y, cnt, m, which, t_off, t_stride, mode, mem, stream); });
}

int ecfft_tree_table(ecfft_ctx* ctx, size_t m, int which, void* host_out, size_t cap, size_t* count) {
    if (!ctx || shard_only(ctx)) return ECFFT_ERR_BAD_ARG;
    DeviceGuard dev(ctx->device);
    if (!dev.ok) return ECFFT_ERR_HIP;
    // reads immutable tables only (its device temporaries are its own hipMalloc blocks, not the pool): no ordering against
    // transform calls in flight is needed
    return guarded([&] { return ctx->field == ECFFT_FIELD_SECP256K1 ? table_of(*ctx->secp, m, which, host_out, cap, count)
                                                                    : table_of(*ctx->m31, m, which, host_out, cap, count); });
}

int ecfft_profile_enable(ecfft_ctx* ctx, int on) {
    if (!ctx) return ECFFT_ERR_BAD_ARG;
    Profiler& p = ctx->field == ECFFT_FIELD_SECP256K1 ? ctx->secp->profiler() : ctx->m31->profiler();
    DeviceGuard dev(ctx->device);
    if (!dev.ok || hipDeviceSynchronize() != hipSuccess) return ECFFT_ERR_HIP;
    p.reset(); p.on = on != 0;
    return ECFFT_OK;
}
int ecfft_profile_classes(void) { return KC_COUNT; }
int ecfft_profile_read(ecfft_ctx* ctx, int cls, char* name, size_t cap, uint64_t* launches, double* ms_total,
                       double* alg_bytes_total) {
    if (!ctx || cls < 0 || cls >= KC_COUNT) return ECFFT_ERR_BAD_ARG;
    Profiler& p = ctx->field == ECFFT_FIELD_SECP256K1 ? ctx->secp->profiler() : ctx->m31->profiler();
    DeviceGuard dev(ctx->device);
    if (!dev.ok || hipDeviceSynchronize() != hipSuccess) return ECFFT_ERR_HIP;
    p.collect();
    if (name && cap) snprintf(name, cap, "%s", kKernelClassName[cls]);
    if (launches) *launches = p.launches(cls);
    if (ms_total) *ms_total = p.ms(cls);
    if (alg_bytes_total) *alg_bytes_total = p.bytes(cls);
    return ECFFT_OK;
}

int ecfft_tree_rational_maps(ecfft_ctx* ctx, void* map_num3_out, void* map_den3_out) {
    if (!ctx || shard_only(ctx)) return ECFFT_ERR_BAD_ARG;
    if (ctx->field == ECFFT_FIELD_SECP256K1) {
        const auto& maps = ctx->secp->host().maps;
        for (size_t k = 0; k < maps.size(); ++k) {
            RatMap<Secp256k1> m = maps[k];
            secp_to_mont_host(m.num, 3); secp_to_mont_host(m.den, 3);
            if (map_num3_out) memcpy((char*)map_num3_out + 96 * k, m.num, 96);
            if (map_den3_out) memcpy((char*)map_den3_out + 96 * k, m.den, 96);
        }
    } else {
        const auto& maps = ctx->m31->host().maps;
        for (size_t k = 0; k < maps.size(); ++k) {
            if (map_num3_out) memcpy((char*)map_num3_out + 12 * k, maps[k].num, 12);
            if (map_den3_out) memcpy((char*)map_den3_out + 12 * k, maps[k].den, 12);
        }
    }
    return ECFFT_OK;
}

int ecfft_fftree_serialize(ecfft_ctx* ctx, int compress, void* buf, size_t cap, size_t* len) {
    if (!ctx || shard_only(ctx)) return ECFFT_ERR_BAD_ARG;
    const size_t need = ctx->field == ECFFT_FIELD_SECP256K1 ? wire_size(*ctx->secp, compress) : wire_size(*ctx->m31, compress);
    if (len) *len = need;
    if (!buf) return ECFFT_OK;
    if (cap < need) return ECFFT_ERR_BAD_ARG;
    DeviceGuard dev(ctx->device);
    if (!dev.ok) return ECFFT_ERR_HIP;
    return guarded([&] { return ctx->field == ECFFT_FIELD_SECP256K1 ? wire_write(*ctx->secp, compress, (uint8_t*)buf)
                                                                    : wire_write(*ctx->m31, compress, (uint8_t*)buf); });
}

int ecfft_fftree_deserialize(int field, const void* bytes, size_t len, int compress, int device, int verify, ecfft_ctx** out) {
    if (!out) return ECFFT_ERR_BAD_ARG;
    *out = nullptr;
    if (!bytes) return ECFFT_ERR_BAD_ARG;
    if (field != ECFFT_FIELD_SECP256K1 && field != ECFFT_FIELD_M31) return ECFFT_ERR_BAD_ARG;
    if (!have_device(device)) return ECFFT_ERR_HIP;
    return guarded([&] { return field == ECFFT_FIELD_SECP256K1
        ? wire_read<Secp256k1>(field, (const uint8_t*)bytes, len, compress, device, verify, out, &ecfft_ctx::secp)
        : wire_read<M31>(field, (const uint8_t*)bytes, len, compress, device, verify, out, &ecfft_ctx::m31); });
}

int ecfft_elems_to_standard(int field, const void* in, void* out, size_t n) {
    if (!in || !out) return ECFFT_ERR_BAD_ARG;
    if (field == ECFFT_FIELD_M31) { if (in != out) memmove(out, in, n * 4); return ECFFT_OK; }
    if (field != ECFFT_FIELD_SECP256K1) return ECFFT_ERR_BAD_ARG;
    if (in != out) memmove(out, in, n * 32);
    secp_from_mont_host((Fe256*)out, n);
    return ECFFT_OK;
}
int ecfft_elems_from_standard(int field, const void* in, void* out, size_t n) {
    if (!in || !out) return ECFFT_ERR_BAD_ARG;
    if (field == ECFFT_FIELD_M31) { if (in != out) memmove(out, in, n * 4); return ECFFT_OK; }
    if (field != ECFFT_FIELD_SECP256K1) return ECFFT_ERR_BAD_ARG;
    if (in != out) memmove(out, in, n * 32);
    secp_to_mont_host((Fe256*)out, n);
    return ECFFT_OK;
}

#ifdef ECFFT_TEST_HOOKS
int ecfft_selftest_field(int field, int op, const void* a, const void* b, const void* c, void* out, size_t n, int device) {
    if (field == ECFFT_FIELD_SECP256K1) return run_selftest<Secp256k1>(op, a, b, c, out, n, device);
    if (field == ECFFT_FIELD_M31) return run_selftest<M31>(op, a, b, c, out, n, device);
    return ECFFT_ERR_BAD_ARG;
}

int ecfft_selftest_blk16(const void* matrix256, const void* x, void* out, size_t n, int device) {
    if (!matrix256 || !x || !out || !n || n % Blk16::kSub) return ECFFT_ERR_BAD_ARG;
    if (!have_device(device)) return ECFFT_ERR_HIP;
    DeviceGuard dev(device);
    if (!dev.ok) return ECFFT_ERR_HIP;
    Fe256 *dT = nullptr, *dx = nullptr; uint8_t* dA = nullptr;
    bool ok = hipMalloc(&dT, 256 * sizeof(Fe256)) == hipSuccess && hipMalloc(&dx, n * sizeof(Fe256)) == hipSuccess &&
              hipMalloc(&dA, Blk16::kABytes + Blk16::kKWords * 8) == hipSuccess;
    ok = ok && hipMemcpy(dT, matrix256, 256 * sizeof(Fe256), hipMemcpyHostToDevice) == hipSuccess && hipMemcpy(dx, x, n * sizeof(Fe256), hipMemcpyHostToDevice) == hipSuccess;
    if (ok) {
        unsigned long long* dK = reinterpret_cast<unsigned long long*>(dA + Blk16::kABytes);
        hipLaunchKernelGGL(k_blk16_from_matrix, dim3(1), dim3(256), 0, nullptr, dT, dA, dK, false);
        hipLaunchKernelGGL(k_blk16_apply, dim3((unsigned)(n / Blk16::kSub)), dim3(512), 0, nullptr, dx, dA, dK);
        ok = hipDeviceSynchronize() == hipSuccess && hipMemcpy(out, dx, n * sizeof(Fe256), hipMemcpyDeviceToHost) == hipSuccess;
    }
    (void)hipFree(dT); (void)hipFree(dx); (void)hipFree(dA);
    return ok ? ECFFT_OK : ECFFT_ERR_HIP;
}

int ecfft_selftest_blk16_small(const void* matrix256, const void* x, void* out, size_t n, int mode, int device) {
    if (!matrix256 || !x || !out || !n || n % 256 || mode < 1 || mode > 4) return ECFFT_ERR_BAD_ARG;
    if (!have_device(device)) return ECFFT_ERR_HIP;
    DeviceGuard dev(device);
    if (!dev.ok) return ECFFT_ERR_HIP;
    Fe256 *dT = nullptr, *dx = nullptr; uint8_t* dA = nullptr;
    bool ok = hipMalloc(&dT, 256 * sizeof(Fe256)) == hipSuccess && hipMalloc(&dx, n * sizeof(Fe256)) == hipSuccess &&
              hipMalloc(&dA, Blk16::kABytes + Blk16::kKWords * 8) == hipSuccess;
    ok = ok && hipMemcpy(dT, matrix256, 256 * sizeof(Fe256), hipMemcpyHostToDevice) == hipSuccess && hipMemcpy(dx, x, n * sizeof(Fe256), hipMemcpyHostToDevice) == hipSuccess;
    if (ok) {
        unsigned long long* dK = reinterpret_cast<unsigned long long*>(dA + Blk16::kABytes);
        hipLaunchKernelGGL(k_blk16_from_matrix, dim3(1), dim3(256), 0, nullptr, dT, dA, dK, false);
        if (mode == 1) hipLaunchKernelGGL(k_blk16_apply_n16<1>, dim3((unsigned)(n / 256)), dim3(256), 0, nullptr, dx, dA, dK);
        else if (mode == 2) hipLaunchKernelGGL(k_blk16_apply_n16<2>, dim3((unsigned)(n / 256)), dim3(128), 0, nullptr, dx, dA, dK);
        else if (mode == 3) hipLaunchKernelGGL(k_blk16_apply_n16<3>, dim3((unsigned)(n / 128)), dim3(128), 0, nullptr, dx, dA, dK);
        else hipLaunchKernelGGL(k_blk16_apply_n16<4>, dim3((unsigned)(n / 256)), dim3(256), 0, nullptr, dx, dA, dK);
        ok = hipDeviceSynchronize() == hipSuccess && hipMemcpy(out, dx, n * sizeof(Fe256), hipMemcpyDeviceToHost) == hipSuccess;
    }
    (void)hipFree(dT); (void)hipFree(dx); (void)hipFree(dA);
    return ok ? ECFFT_OK : ECFFT_ERR_HIP;
}

int ecfft_selftest_blk32(const void* matrix1024, const void* x, void* out, size_t n, int device) {
    if (!matrix1024 || !x || !out || !n || n % 1024) return ECFFT_ERR_BAD_ARG;
    if (!have_device(device)) return ECFFT_ERR_HIP;
    DeviceGuard dev(device);
    if (!dev.ok) return ECFFT_ERR_HIP;
    Fe256 *dT = nullptr, *dx = nullptr, *dcs = nullptr; uint8_t* dA = nullptr;
    bool ok = hipMalloc(&dT, 1024 * sizeof(Fe256)) == hipSuccess && hipMalloc(&dcs, 1024 * sizeof(Fe256)) == hipSuccess && hipMalloc(&dx, n * sizeof(Fe256)) == hipSuccess &&
              hipMalloc(&dA, Blk16::kABytes32 + Blk16::kKWords32 * 8) == hipSuccess;
    ok = ok && hipMemcpy(dT, matrix1024, 1024 * sizeof(Fe256), hipMemcpyHostToDevice) == hipSuccess && hipMemcpy(dx, x, n * sizeof(Fe256), hipMemcpyHostToDevice) == hipSuccess;
    if (ok) {
        unsigned long long* dK = reinterpret_cast<unsigned long long*>(dA + Blk16::kABytes32);
        hipLaunchKernelGGL(k_blk32_expand, dim3(4), dim3(256), 0, nullptr, (const Fe256*)dT, dA, dcs, false);
        hipLaunchKernelGGL(k_blk32_seeds, dim3(1), dim3(32), 0, nullptr, (const Fe256*)dcs, dK);
        hipLaunchKernelGGL(k_blk32_apply, dim3((unsigned)(n / 1024)), dim3(512), 0, nullptr, dx, dA, dK);
        ok = hipDeviceSynchronize() == hipSuccess && hipMemcpy(out, dx, n * sizeof(Fe256), hipMemcpyDeviceToHost) == hipSuccess;
    }
    (void)hipFree(dT); (void)hipFree(dx); (void)hipFree(dA); (void)hipFree(dcs);
    return ok ? ECFFT_OK : ECFFT_ERR_HIP;
}
// which composite map the 1024-element low-level kernels of this context run for the lowest levels of ENTER (dir 0) / EXIT (dir 1):
// 32 (levels 1..5), 16 (levels 1..4) or 0 (level code)
int ecfft_ctx_low_map(const ecfft_ctx* ctx, int dir) {
    if (!ctx || dir < 0 || dir > 1 || ctx->field != ECFFT_FIELD_SECP256K1 || !ctx->secp) return 0;
    return ctx->secp->low_map(dir);
}

#endif  // ECFFT_TEST_HOOKS
int ecfft_mul_ceiling(int field, int device, int waves_per_simd, double* mul_per_s) {
    if (field == ECFFT_FIELD_SECP256K1) return run_mul_ceiling<Secp256k1>(device, waves_per_simd, mul_per_s);
    if (field == ECFFT_FIELD_M31) return run_mul_ceiling<M31>(device, waves_per_simd, mul_per_s);
    return ECFFT_ERR_BAD_ARG;
}

int ecfft_device_alloc(int device, size_t bytes, void** out) {
    if (!out) return ECFFT_ERR_BAD_ARG;
    *out = nullptr;
    if (!have_device(device)) return ECFFT_ERR_HIP;
    DeviceGuard dev(device);
    if (!dev.ok) return ECFFT_ERR_HIP;
    return hipMalloc(out, bytes ? bytes : 1) == hipSuccess ? ECFFT_OK : ECFFT_ERR_HIP;
}
int ecfft_device_free(void* ptr) { return !ptr || hipFree(ptr) == hipSuccess ? ECFFT_OK : ECFFT_ERR_HIP; }
int ecfft_device_sync(int device) {
    DeviceGuard dev(device);
    return dev.ok && hipDeviceSynchronize() == hipSuccess ? ECFFT_OK : ECFFT_ERR_HIP;
}

int ecfft_shader_clock(int field, int device, double* mhz) {
    if (field == ECFFT_FIELD_SECP256K1) return run_shader_clock<Secp256k1>(device, mhz);
    if (field == ECFFT_FIELD_M31) return run_shader_clock<M31>(device, mhz);
    return ECFFT_ERR_BAD_ARG;
}

int ecfft_device_copy(void* dst, const void* src, size_t bytes, int kind) {
    hipMemcpyKind k = kind == 0 ? hipMemcpyDeviceToHost : (kind == 1 ? hipMemcpyHostToDevice : hipMemcpyDeviceToDevice);
    if (!dst || !src || kind < 0 || kind > 2) return ECFFT_ERR_BAD_ARG;
    return hipMemcpy(dst, src, bytes, k) == hipSuccess ? ECFFT_OK : ECFFT_ERR_HIP;
}

int ecfft_device_info(int device, char* buf, size_t cap) {
    if (!buf || !cap) return ECFFT_ERR_BAD_ARG;
    hipDeviceProp_t p;
    if (hipGetDeviceProperties(&p, device) != hipSuccess) { snprintf(buf, cap, "no HIP device"); return ECFFT_ERR_HIP; }
    snprintf(buf, cap, "%s (%s), %d CUs, %.1f GiB", p.name, p.gcnArchName, p.multiProcessorCount, p.totalGlobalMem / 1073741824.0);
    return ECFFT_OK;
}

}  // extern "C"
