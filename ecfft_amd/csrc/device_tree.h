// Device-resident FFTree chain T_1 < T_2 < ... < T_N and the iterative, level-by-level drivers of
// EXTEND / ENTER / EXIT (the reference's recursive extend_impl / enter_impl / exit_impl / redc_impl,
// /root/reference/src/fftree.rs:72-120, 143-161, 200-224, 232-259, re-expressed as batched sweeps,
// SURVEY.md Appendix A) plus the on-GPU table precompute (from_tree, src/fftree.rs:318-463).
//
// HBM layout ("Moiety precompute layout"): per tree T_m (m = 2e leaves), per moiety
// parity s in {0,1}, structure-of-arrays tables holding only the half that parity uses:
//   p0[s], p1[s], np0[s], dinv[s]   e-1 entries, stage k at offset e - 2*h_k   (h_k = e >> (k+1))
//   c0t[s] = np0[s]*dinv[s]         likewise    (pair-split decompose of the latency regime: two independent products)
//   w[s], winv[s]                   e entries   (normalisation weights of the parity's leaves)
//   xe, w1x                         e entries   ENTER combine
//   A1, B1, NB2, C1, D1, xie        e entries   EXIT pointwise steps with every inverse pre-fused
// All tables are PLAIN residues; user data stays in the crate's Montgomery form (field_secp256k1.h).  The tables above are
// stored as F::telem — for secp256k1 the pair (t, t*2^128 mod p) that the 12-word multiply wants, for M31 the doubled constant
// 2t (field_m31.h) — and are written by to_tables() from plain temporaries; the reference's own tables (xnn, z*) stay plain
// F::elem arrays.  Below the level drivers: the sharded (multi-GPU) drivers extend_split / api_enter_split / api_exit_split.
#pragma once
#include <hip/hip_runtime.h>
#include <vector>
#include <set>
#include <map>
#include <mutex>
#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <new>
#include "kernels.h"
#include "host_curve.h"
#include "transport.h"

namespace ecfft {

// A/B switches of the tuning experiments (ECFFT_NO_MFMA, ECFFT_NO_LOW16, ECFFT_LOW32, ECFFT_NO_ROW256, ...): a test / tuning build
// (-DECFFT_TEST_HOOKS, tests/hooks/) reads them from the environment, the SHIPPED library has none of them — it reads no environment
// variable at all, so two ranks of one job cannot end up on different exchange patterns because their environments differ.
#ifdef ECFFT_TEST_HOOKS
inline const char* ab_env(const char* name) { return getenv(name); }
#else
inline const char* ab_env(const char*) { return nullptr; }
#endif

#define ECFFT_HIP_TRY(x)                                                                   \
    do { hipError_t e_ = (x); if (e_ != hipSuccess) {                                      \
             fprintf(stderr, "ecfft: HIP error '%s' at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); \
             return false; } } while (0)

// Allocation failures inside the chain (table arena, temporaries) unwind to the C-ABI entry point, which maps them to
// ECFFT_ERR_HIP — a shared library must never abort() its host process.
struct DeviceAllocError : std::bad_alloc {
    const char* what() const noexcept override { return "ecfft: device allocation failed"; }
};

static inline unsigned ilog2(size_t n) { unsigned l = 0; while (n > 1) { n >>= 1; ++l; } return l; }
static inline unsigned nblocks(size_t n) { return (unsigned)((n + kBlock - 1) / kBlock); }

template <class Fn>
static inline void foreach_n(hipStream_t s, size_t n, Fn fn) {
    if (n) hipLaunchKernelGGL(k_foreach<Fn>, dim3(nblocks(n)), dim3(kBlock), 0, s, fn, n);
}


// ---------------------------------------------------------------------------------------------
// Optional per-launch timing with HIP events on the launch stream (bench.py's "roofline" object).
// Every hot-path launch goes through ECFFT_LAUNCH with a kernel class and the ALGORITHMIC bytes of
// that launch in the stage-streaming model of SURVEY.md section 8(d): each butterfly stage it covers
// reads and writes its operand once and reads its stage tables once; each pointwise step reads its
// operands and tables and writes its result once.
// ---------------------------------------------------------------------------------------------
enum KernelClass { KC_DECOMPOSE = 0, KC_RECOMBINE, KC_POINTWISE, KC_ROW, KC_COL, KC_FUSED_ENTER, KC_FUSED_EXIT, KC_COUNT };
static const char* const kKernelClassName[KC_COUNT] = {"k_decompose_stage", "k_recombine_stage", "pointwise",
                                                       "k_stages_lds", "k_stages_col", "k_enter_low", "k_exit_low"};

class Profiler {
public:
    bool on = false;
    ~Profiler() { for (hipEvent_t e : pool_) (void)hipEventDestroy(e); }
    void reset() { used_ = 0; recs_.clear(); for (int i = 0; i < KC_COUNT; ++i) { ms_[i] = 0; bytes_[i] = 0; n_[i] = 0; } }
    hipEvent_t next() {
        if (used_ == pool_.size()) { hipEvent_t e; (void)hipEventCreate(&e); pool_.push_back(e); }
        return pool_[used_++];
    }
    void begin(hipStream_t s, int cls, double bytes) {
        hipEvent_t a = next(); (void)hipEventRecord(a, s);
        recs_.push_back({cls, bytes, a, nullptr});
    }
    void end(hipStream_t s) { hipEvent_t b = next(); (void)hipEventRecord(b, s); recs_.back().stop = b; }
    void collect() {   // caller has synchronised the stream(s)
        for (const Rec& r : recs_) {
            float ms = 0; if (hipEventElapsedTime(&ms, r.start, r.stop) != hipSuccess) continue;
            ms_[r.cls] += ms; bytes_[r.cls] += r.bytes; n_[r.cls] += 1;
        }
        recs_.clear(); used_ = 0;
    }
    double ms(int c) const { return ms_[c]; }
    double bytes(int c) const { return bytes_[c]; }
    uint64_t launches(int c) const { return n_[c]; }
private:
    struct Rec { int cls; double bytes; hipEvent_t start, stop; };
    std::vector<hipEvent_t> pool_; size_t used_ = 0;
    std::vector<Rec> recs_;
    double ms_[KC_COUNT] = {}, bytes_[KC_COUNT] = {}; uint64_t n_[KC_COUNT] = {};
};

#define ECFFT_LAUNCH(cls, alg_bytes, kern, grid, block, lds, stream, ...)               \
    do { if (prof_.on) prof_.begin(stream, cls, (double)(alg_bytes));                  \
         hipLaunchKernelGGL(kern, grid, block, lds, stream, __VA_ARGS__);               \
         if (prof_.on) prof_.end(stream); } while (0)

#ifndef ECFFT_BATCH_SPLIT
#define ECFFT_BATCH_SPLIT 1
#endif
#ifndef ECFFT_BATCH_WAYS
#define ECFFT_BATCH_WAYS 2      // a batch runs as up to this many concurrent parts (whole polynomials each, one stream per part); A/B: profiles/r06/batch_ways_ab.txt
#endif
template <class F>
class DeviceChain {
public:
    using E = typename F::elem;
    using TE = typename F::telem;  // table constant in the form the kernels multiply by (secp256k1: the pair (t, t*2^128))
    using Tree = LevelTables<F>;   // per-tree table set (kernels.h); an array of them is mirrored on the device

    ~DeviceChain() { release(); }

    size_t size() const { return N_; }
    unsigned log_size() const { return L_; }
    const Tree& tree(unsigned log_m) const { return trees_[log_m]; }
    // the table set the EXTEND machinery uses for the tree with 2^log_m leaves: trees_ / sets_, unless a temporary share of that
    // tree is installed (the sharded EXIT build needs two different shares of one tree)
    const Tree& tree_at(unsigned log_m) const { return (ovr_tree_ && ovr_tree_->log_m == log_m) ? *ovr_tree_ : trees_[log_m]; }
    const HostTree<F>& host() const { return host_; }
    int low_map(int dir) const {
        if (trees_.size() > 5 && trees_[5].low32_A[dir]) return 32;
        if (trees_.size() > 4 && trees_[4].low16_A[dir]) return 16;
        return 0;
    }
    const E* f_device() const { return f_; }
    std::mutex& lock() { return mu_; }
    // test hook, process wide: the rank whose local part of the next collective ecfft_build_exit_shard reports failure (-1: none).
    // Set through the ABI only (ecfft_test_fail_build_rank) — no environment variable can reach it.
#ifdef ECFFT_TEST_HOOKS
    static std::atomic<int>& test_fail_build_rank() { static std::atomic<int> r{-1}; return r; }
#endif
    Profiler& profiler() const { return prof_; }

    // ------------------------------------------------------------------------------------------
    // construction
    // ------------------------------------------------------------------------------------------
    bool build(HostTree<F>&& ht, int device) {
        host_ = std::move(ht);
        N_ = host_.n; L_ = ilog2(N_); device_ = device;
        ECFFT_HIP_TRY(hipSetDevice(device_));
        hipStream_t s = nullptr;
        // arena per tree of m leaves: 6m elements of reference tables (xnn, z*) + 11m table constants of the hot path
        // (F::telem each); + f (2N) + den coefficients
        size_t total = 2 * N_ + 64;
        for (unsigned l = 0; l <= L_; ++l) total += (6 + 11 * kTeElems) * ((size_t)1 << l) + 1024 + blk16_elems(l);
        total += low16_elems();
        ECFFT_HIP_TRY(hipMalloc(&arena_, total * sizeof(E)));
        arena_cap_ = total; arena_used_ = 0;
        f_ = take(2 * N_);
        if (!points_on_device(f_, s)) { fprintf(stderr, "ecfft: point set construction failed\n"); return false; }
        // denominators of the maps: v_k(x) = den0 + den1 x (degree 1 for both curve families)
        std::vector<E> den(2 * (L_ ? L_ : 1));
        for (unsigned k = 0; k < L_; ++k) { den[2 * k] = host_.maps[k].den[0]; den[2 * k + 1] = host_.maps[k].den[1]; }
        den_ = take(2 * (L_ ? L_ : 1));
        ECFFT_HIP_TRY(hipMemcpyAsync(den_, den.data(), den.size() * sizeof(E), hipMemcpyHostToDevice, s));
        // transform scratch: 5 N (grown on demand for batched calls); side stream for the two-halves schedule
        if (!ensure_scratch(N_)) return false;
        create_side_streams();
        trees_.assign(L_ + 1, Tree{});
        {   // ~32 m temporaries are alive while T_m is built; one slab instead of ~80 hipMalloc/hipFree pairs per tree
            size_t want = 40 * N_ + 16384, cap_bytes = (size_t)16 << 30;
            if (want * sizeof(E) > cap_bytes) want = cap_bytes / sizeof(E);
            if (hipMalloc(&slab_, want * sizeof(E)) == hipSuccess) slab_cap_ = want; else { (void)hipGetLastError(); slab_ = nullptr; slab_cap_ = 0; }
            slab_used_ = 0;
        }
        for (unsigned l = 0; l <= L_; ++l) {
            if (!build_tree(l, s)) return false;
        }
        if (!build_low16(L_, s)) return false;
        ECFFT_HIP_TRY(hipStreamSynchronize(s));
        if (slab_) { (void)hipFree(slab_); slab_ = nullptr; slab_cap_ = slab_used_ = 0; }
        temps_free();
        ECFFT_HIP_TRY(hipMalloc(&d_trees_, (L_ + 1) * sizeof(Tree)));
        ECFFT_HIP_TRY(hipMemcpy(d_trees_, trees_.data(), (L_ + 1) * sizeof(Tree), hipMemcpyHostToDevice));
        return true;
    }

    // ------------------------------------------------------------------------------------------
    // SHARDED EXTEND context ("matrix tables shard the same way").  For ONE EXTEND of e evaluations split over
    // P = 2^log_p GPUs a rank never touches most of T_2e: the cyclic stages k < log_p read only the stage-table entries
    // i = i'*P + rank, the block-local stages k >= log_p only the last e/P entries of each stage table (pair distances <= e/2P),
    // the 1/W and W scalings only the rank's e/P positions, and no other tree of the chain and none of the z tables is needed at
    // all.  Everything a rank does need is pointwise in the point set (layers of f, isogeny denominators) plus batched
    // inversions, so it is built directly — no full tree ever exists on any GPU: ~24 e/P table constants instead of the 28*(2e)
    // elements of T_2e plus the 56e of the chain below it, and a build that is O(e/P) kernels' work instead of the chain's.
    // Layout: trees_[log2(2e)] holds BIASED pointers (base - first_needed_index) for the block-local stage tables and for
    // w / winv, so the single-GPU kernels index them unchanged; the cyclic entries live in compact arrays cyc_* laid out like the
    // stage tables of a length-e/P vector (stage k at offset e/P - 2*h_k/P).
    // ------------------------------------------------------------------------------------------
    // HBM held by this context between calls: table arena + transform scratch (pooled temporaries of the algorithm wrappers
    // and the host-call staging buffer come and go)
    size_t device_bytes() const { return (arena_cap_ + scratch_cap_) * sizeof(E) + pool_bytes() + full_cyc_bytes_; }
    // ecfft_ctx_trim: give every idle pooled temporary back to the device (between calls; takes the context lock)
    // `also`: runs under the same lock (the ABI layer frees its host-call staging buffer there: run_op / run_alg use it under lock())
    template <class Fn>
    void trim(Fn&& also) { std::lock_guard<std::mutex> g(mu_); temps_trim(0); (void)hipDeviceSynchronize(); full_cyclic_free(); also(); }
    enum { kShardNone = 0, kShardExtend = 1, kShardEnter = 2, kShardExit = 3 };
    int shard_kind() const { return shard_kind_; }
    bool shard_mode() const { return shard_kind_ != kShardNone; }
    unsigned shard_log_p() const { return shard_log_p_; }
    unsigned shard_rank() const { return shard_rank_; }
    // side streams of the two-halves schedule (enter_rec / exit_rec)
    void create_side_streams() {
        const int want = ((1 << kSplitDepth) - 1) > (ECFFT_BATCH_WAYS - 1) ? ((1 << kSplitDepth) - 1) : (ECFFT_BATCH_WAYS - 1);   // halves of one transform / parts of a batch
        for (nside_ = 0; nside_ < want && nside_ < kMaxSides; ++nside_) {
            if (hipStreamCreateWithFlags(&sides_[nside_], hipStreamNonBlocking) != hipSuccess || hipEventCreateWithFlags(&ev_fork_[nside_], hipEventDisableTiming) != hipSuccess ||
                hipEventCreateWithFlags(&ev_join_[nside_], hipEventDisableTiming) != hipSuccess) break;
        }
    }
    // One rank's share of the EXTEND tables of the tree with 2^log_m leaves (a tree of the chain of T_N: its layer k is every
    // (N/m)-th point of the top tree's layer k) for a split over 2^log_p ranks; f = the device copy of the point set.
    struct ShardSet {
        bool valid = false; unsigned log_p = 0, rank = 0;
        TE* cyc[2][4] = {};      // per parity: np0, dinv, p0, p1 of the cyclic stages, compact
        TE* cycw[2][2] = {};     // per parity: w, winv of the rank's cyclic positions
    };
    // only_target = 0 / 1: keep only what EXTENDs towards that moiety read (decompose-side tables of the source parity,
    // recombine-side tables of the target parity): half the constants; -1: both directions
    static size_t shard_set_elems(size_t c, int only_target = -1) {
        return (only_target < 0 ? 2 : 1) * ((13 * c + 128) * kTeElems + (sizeof(E) == 32 ? Blk16::kArenaElems + 8 : 0)) + 64;   // + the matrix-core tables of the rank's row passes
    }
    bool build_shard_set(unsigned log_m, unsigned log_p, unsigned rank, const E* f, hipStream_t s, int only_target = -1,
                         Tree* tout = nullptr, ShardSet* sout = nullptr) {
        const size_t m = (size_t)1 << log_m, e = m / 2, P = (size_t)1 << log_p, c = e >> log_p, stride = N_ / m;
        if (log_m < 2 || c < P || c < 2 || rank >= P) return false;
        const unsigned le = log_m - 1;
        Tree& T = tout ? *tout : trees_[log_m]; ShardSet& S = sout ? *sout : sets_[log_m];
        T.m = m; T.e = e; T.log_m = log_m;
        S.valid = true; S.log_p = log_p; S.rank = rank;
        const size_t N = N_;
        const size_t loc0 = e - 2 * (e >> (log_p + 1));          // first entry of the block-local stages: e - 2*h_{log_p} = e - c
        const size_t nloc = c;                                   // entries [loc0, e) (the last one is padding, as in the full tables)
        E* plain[2][4];                                          // local p0, p1, np0, dinv (plain) for the inner constants
        for (int sg = 0; sg < 2; ++sg) {
            const bool dec = only_target < 0 || sg == 1 - only_target;       // this parity is an EXTEND source
            const bool rec = only_target < 0 || sg == only_target;           // ... an EXTEND target
            // ---- block-local stages k >= log_p: the last c entries of every stage table
            E *p0 = temp(nloc), *p1 = temp(nloc), *np0 = temp(nloc), *dinv = temp(nloc), *c0 = temp(nloc);
            (void)hipMemsetAsync(p0 + (nloc - 1), 0, sizeof(E), s); (void)hipMemsetAsync(p1 + (nloc - 1), 0, sizeof(E), s);
            (void)hipMemsetAsync(np0 + (nloc - 1), 0, sizeof(E), s); (void)hipMemsetAsync(dinv + (nloc - 1), 0, sizeof(E), s);
            foreach_n(s, nloc - 1, [=] __device__(size_t gl) {
                const size_t g = loc0 + gl, rem = e - g;         // same entry as build_tree's concatenated stage tables
                unsigned k = 0; size_t h = e >> 1;
                while (rem <= h) { h >>= 1; ++k; }
                const size_t i = g - (e - 2 * h), lay = N >> k;
                E a = f[lay + (2 * i + sg) * stride], b = f[lay + (2 * i + sg + 2 * h) * stride];
                p0[gl] = a; p1[gl] = b; np0[gl] = F::neg(a); dinv[gl] = F::sub(b, a);
            });
            batch_inv(dinv, dinv, nloc - 1, s);
            (void)hipMemsetAsync(c0 + (nloc - 1), 0, sizeof(E), s);
            foreach_n(s, nloc - 1, [=] __device__(size_t gl) { c0[gl] = F::mul(np0[gl], dinv[gl]); });
            plain[sg][0] = p0; plain[sg][1] = p1; plain[sg][2] = np0; plain[sg][3] = dinv;
            if (rec) { T.p0[sg] = to_tables(p0, nloc, s) - loc0; T.p1[sg] = to_tables(p1, nloc, s) - loc0; }
            if (dec) { T.np0[sg] = to_tables(np0, nloc, s) - loc0; T.dinv[sg] = to_tables(dinv, nloc, s) - loc0; T.c0t[sg] = to_tables(c0, nloc, s) - loc0; }
            // ---- cyclic stages k < log_p: entries i = i'*P + rank, compact, laid out like a length-c vector's stage tables
            const size_t ncyc = c;
            E *q0 = temp(ncyc), *q1 = temp(ncyc), *nq0 = temp(ncyc), *qd = temp(ncyc);
            (void)hipMemsetAsync(q0, 0, ncyc * sizeof(E), s); (void)hipMemsetAsync(q1, 0, ncyc * sizeof(E), s);
            (void)hipMemsetAsync(nq0, 0, ncyc * sizeof(E), s); (void)hipMemsetAsync(qd, 0, ncyc * sizeof(E), s);
            for (unsigned k = 0; k < log_p; ++k) {
                const size_t h = e >> (k + 1), hl = h >> log_p, offl = c - 2 * hl, lay = N >> k;
                foreach_n(s, hl, [=] __device__(size_t il) {
                    const size_t i = il * P + rank;
                    E a = f[lay + (2 * i + sg) * stride], b = f[lay + (2 * i + sg + 2 * h) * stride];
                    q0[offl + il] = a; q1[offl + il] = b; nq0[offl + il] = F::neg(a); qd[offl + il] = F::sub(b, a);
                });
            }
            batch_inv(qd, qd, ncyc, s);                          // zero padding stays zero
            if (dec) { S.cyc[sg][0] = to_tables(nq0, ncyc, s); S.cyc[sg][1] = to_tables(qd, ncyc, s); }
            if (rec) { S.cyc[sg][2] = to_tables(q0, ncyc, s); S.cyc[sg][3] = to_tables(q1, ncyc, s); }
            // ---- normalisation weights (DESIGN.md "Normalised butterflies") of the rank's c BLOCK positions g0 + il and of its
            // c CYCLIC positions il*P + rank (cyclic-in / cyclic-out calls)
            const E* dn = den_; const size_t g0 = (size_t)rank * c;
            for (int cyc = 0; cyc < 2; ++cyc) {
                E *w = temp(c), *wi = temp(c);
                foreach_n(s, c, [=] __device__(size_t il) {
                    const size_t j = 2 * (cyc ? il * P + rank : g0 + il) + sg;
                    E U = F::one(), C = F::one();
                    for (unsigned b = 0; b + 1 < le; ++b) {
                        const size_t lsz = m >> b;
                        E sb = f[(N >> b) + (j & (lsz - 1)) * stride];
                        E V = F::mul_add(dn[2 * b + 1], sb, dn[2 * b]);
                        C = F::mul(C, V);
                        U = F::mul(F::sqr(U), C);
                    }
                    w[il] = U;
                });
                batch_inv(w, wi, c, s);
                if (cyc) { if (rec) S.cycw[sg][0] = to_tables(w, c, s); if (dec) S.cycw[sg][1] = to_tables(wi, c, s); }
                else { if (rec) T.w[sg] = to_tables(w, c, s) - g0; if (dec) T.winv[sg] = to_tables(wi, c, s) - g0; }
            }
        }
        for (int sg = 0; sg < 2; ++sg) {                         // merged innermost stage pair (build_tree): entries at index e-2
            if (only_target >= 0 && sg != 1 - only_target) continue;   // indexed by the SOURCE parity
            E* in = temp(2);
            const size_t o = (e - 2) - loc0;
            const E *sp0 = plain[sg][0] + o, *sdi = plain[sg][3] + o, *tp0 = plain[1 - sg][0] + o, *tp1 = plain[1 - sg][1] + o;
            foreach_n(s, 1, [=] __device__(size_t) {
                in[0] = F::mul(F::sub(tp0[0], sp0[0]), sdi[0]);
                in[1] = F::mul(F::sub(tp1[0], sp0[0]), sdi[0]);
            });
            T.inner[sg] = to_tables(in, 2, s);
        }
        // matrix-core form of the innermost stages for the rank's block-local row passes (their table entries e-16 .. e-2 lie in the
        // stored range [e - c, e) whenever c >= 16); an EXTEND from parity sg reads the decompose tables of sg and the recombine
        // tables of 1 - sg
        T.blk16_A[0] = T.blk16_A[1] = nullptr; T.blk16_K[0] = T.blk16_K[1] = nullptr;
        if constexpr (sizeof(E) == 32) {
            if (c >= 16 && !mfma_off_) {
                for (int sg = 0; sg < 2; ++sg) {
                    if (only_target >= 0 && sg != 1 - only_target) continue;
                    uint8_t* A = reinterpret_cast<uint8_t*>(take(Blk16::kArenaElems));
                    unsigned long long* K = reinterpret_cast<unsigned long long*>(A + Blk16::kABytes);
                    const Blk16BuildArgs ba{T.np0[sg], T.dinv[sg], T.p0[1 - sg], T.p1[1 - sg], T.inner[sg], A, K};
                    hipLaunchKernelGGL(k_blk16_build, dim3(1), dim3(256), 0, s, ba, ba, e);
                    T.blk16_A[sg] = A; T.blk16_K[sg] = K;
                }
            }
        }
        return hipGetLastError() == hipSuccess;
    }
    // z0_s1[i] = Z_0(s1_i) (and z1_s0[i] = Z_1(s0_i)) of the tree with 2^log_m leaves WITHOUT any EXTEND: S0 is the leaf set of
    // the subtree, and the vanishing polynomial of a leaf set factors through the isogeny chain — with A_k the image of the set in
    // layer k (|A_k| = 2^(K-k), K = log m - 1) and psi_k = u_k / v_k, Z_{A_k}(x) = Z_{A_{k+1}}(psi_k(x)) * v_k(x)^|A_{k+1}| /
    // lc(u_k)^|A_{k+1}|, down to the single point A_K.  The images psi_{k-1}(..psi_0(x)) of a LEAF x are the stored layers of the
    // point set, so Z(x) = (x_K - root) * prod_k (v_k(x_k) / lc(u_k))^(2^(K-1-k)): K squarings and 2K multiplies per point.
    // (The same holds for S1: its images are the odd-indexed points of every layer.)  which = 0: out[j] = z0_s1[i0 + j*istride];
    // which = 1: out[j] = z1_s0[i0 + j*istride].  lcinv = 1 / lc(u_k), k < log N.  Used by the sharded EXIT build; build_tree keeps the
    // reference's EXTEND-based construction (src/fftree.rs:386-397) and ecfft_selfcheck_pointwise_z compares the two.
    void pointwise_z(unsigned log_m, int which, size_t i0, size_t cnt, E* out, const E* f, const E* lcinv, hipStream_t s, size_t istride = 1) const {
        const size_t m = (size_t)1 << log_m, stride = N_ / m, N = N_;
        const unsigned K = log_m - 1; const E* dn = den_;
        foreach_n(s, cnt, [=] __device__(size_t j) {
            const size_t leaf = (2 * (i0 + j * istride) + (which == 0 ? 1 : 0)) * stride;  // index in the top tree's layer 0
            E U = F::one();
            for (unsigned k = 0; k < K; ++k) {
                const size_t lsz = N >> k;
                const E xk = f[lsz + (leaf & (lsz - 1))];
                const E ck = F::mul(F::mul_add(dn[2 * k + 1], xk, dn[2 * k]), lcinv[k]);
                U = k == 0 ? ck : F::mul(F::sqr(U), ck);
            }
            const size_t lsz = N >> K;
            const E xK = f[lsz + (leaf & (lsz - 1))], root = f[lsz + (which == 0 ? 0 : (stride & (lsz - 1)))];
            out[j] = F::mul(U, F::sub(xK, root));
        });
    }
    // 1 / lc(u_k) for every map of the chain, on the device (temporary)
    E* upload_lcinv(hipStream_t s) {
        std::vector<E> h(L_ ? L_ : 1, F::one());
        for (unsigned k = 0; k < L_; ++k) h[k] = F::inv(host_.maps[k].num[2]);
        E* d = temp(h.size());
        (void)hipMemcpyAsync(d, h.data(), h.size() * sizeof(E), hipMemcpyHostToDevice, s);
        (void)hipStreamSynchronize(s);
        return d;
    }
    // test hook: number of entries of z0_s1 / z1_s0 of T_m (full context) that differ from the pointwise formula
    long selfcheck_pointwise_z(size_t m) {
        if (shard_mode() || m < 2 || m > N_) return -1;
        const unsigned lm = ilog2(m); const size_t e = m / 2; hipStream_t s = nullptr;
        E* lc = upload_lcinv(s); E* a = temp(e); E* b = temp(e);
        unsigned long long* bad = reinterpret_cast<unsigned long long*>(temp(8));
        (void)hipMemsetAsync(bad, 0, sizeof(unsigned long long), s);
        pointwise_z(lm, 0, 0, e, a, f_, lc, s); pointwise_z(lm, 1, 0, e, b, f_, lc, s);
        const E *z0 = trees_[lm].z0_s1, *z1 = trees_[lm].z1_s0;
        foreach_n(s, e, [=] __device__(size_t i) {
            if (!F::eq(F::canon(a[i]), F::canon(z0[i])) || !F::eq(F::canon(b[i]), F::canon(z1[i]))) atomicAdd(bad, 1ull);
        });
        unsigned long long h = 0;
        bool ok = hipMemcpyAsync(&h, bad, sizeof(h), hipMemcpyDeviceToHost, s) == hipSuccess && hipStreamSynchronize(s) == hipSuccess;
        temps_done();
        return ok ? (long)h : -1;
    }
    // The point set f (2N elements: layer k at [N >> k, 2*(N >> k))) into fdev: uploaded when the host tree carries it (trees made
    // from user leaves, ecfft_fftree_new), otherwise COMPUTED HERE from the generator data — the FftreeField::build_fftree front
    // end (src/lib.rs:66-81, src/ec.rs:545-551) and FFTree::new's layers (src/fftree.rs:42-70) as GPU passes: leaf i = x(off + i*gen)
    // through log N rounds of batched affine additions P_{r+j} = P_r + P_j (one batched inversion per round), then layer k+1 =
    // psi_k(layer k) pointwise with a batched inversion of the denominators.  Same field elements as host_curve.h produces (the
    // host path stays for ecfft_build_points and as the cross-check in the tests); 2^23 points: milliseconds instead of seconds.
    bool points_on_device(E* fdev, hipStream_t s) {
        const size_t n = N_;
        if (!host_.f.empty() && !host_.leaves_only) { ECFFT_HIP_TRY(hipMemcpyAsync(fdev, host_.f.data(), 2 * n * sizeof(E), hipMemcpyHostToDevice, s)); return true; }
        if (!host_.have_gen && !host_.leaves_only) return false;
        const E a2 = host_.curve.a2, a4 = host_.curve.a4, offx = host_.off.x, offy = host_.off.y;
        E* den = temp(n);
        unsigned long long* bad = reinterpret_cast<unsigned long long*>(temp(8));
        (void)hipMemsetAsync(bad, 0, sizeof(unsigned long long), s);
        (void)hipMemsetAsync(fdev, 0, n * sizeof(E), s);          // f[0] is unused; the layers below the leaves are written next
        E* leaves = fdev + n;
        if (host_.leaves_only) {
            ECFFT_HIP_TRY(hipMemcpyAsync(leaves, host_.f.data() + n, n * sizeof(E), hipMemcpyHostToDevice, s));
        } else {
        E *px = temp(n), *py = temp(n);
        if (n > 1) {
            ECFFT_HIP_TRY(hipMemcpyAsync(px + 1, &host_.gen.x, sizeof(E), hipMemcpyHostToDevice, s));
            ECFFT_HIP_TRY(hipMemcpyAsync(py + 1, &host_.gen.y, sizeof(E), hipMemcpyHostToDevice, s));
        }
        for (size_t r = 1; 2 * r <= n && r < n; r <<= 1) {        // compute_leaves (host_curve.h): i*gen for every 0 < i < n
            const size_t cnt = r - 1;
            if (cnt) {
                foreach_n(s, cnt, [=] __device__(size_t t) { den[t] = F::sub(px[t + 1], px[r]); });
                batch_inv(den, den, cnt, s);
                foreach_n(s, cnt, [=] __device__(size_t t) {
                    const size_t j = t + 1;
                    const E xr = px[r], yr = py[r];
                    const E lambda = F::mul(F::sub(py[j], yr), den[t]);
                    const E x3 = F::sub(F::sub(F::sub(F::sqr(lambda), a2), xr), px[j]);
                    px[r + j] = x3;
                    py[r + j] = F::sub(F::mul(lambda, F::sub(xr, x3)), yr);
                });
            }
            if (2 * r < n) foreach_n(s, 1, [=] __device__(size_t) {   // P_{2r} = 2 * P_r (pt_add with p == q)
                const E x = px[r], y = py[r], xx = F::sqr(x);
                const E num = F::add(F::add(F::add(xx, xx), xx), F::add(F::mul(F::add(a2, a2), x), a4));
                const E lambda = F::mul(num, F::inv(F::add(y, y)));
                const E x3 = F::sub(F::sub(F::sub(F::sqr(lambda), a2), x), x);
                px[2 * r] = x3; py[2 * r] = F::sub(F::mul(lambda, F::sub(x, x3)), y);
            });
        }
        foreach_n(s, 1, [=] __device__(size_t) { leaves[0] = offx; });
        if (n > 1) {
            foreach_n(s, n - 1, [=] __device__(size_t t) { den[t] = F::sub(px[t + 1], offx); });
            batch_inv(den, den, n - 1, s);
            foreach_n(s, n - 1, [=] __device__(size_t t) {
                const size_t i = t + 1;
                const E lambda = F::mul(F::sub(py[i], offy), den[t]);
                leaves[i] = F::sub(F::sub(F::sub(F::sqr(lambda), a2), offx), px[i]);
            });
        }
        }
        unsigned k = 0;
        for (size_t sz = n; sz > 1; ++k, sz >>= 1) {              // fill_layers (host_curve.h / src/fftree.rs:42-70)
            const size_t half = sz / 2;
            const E* prev = fdev + sz; E* layer = fdev + half;
            const RatMap<F> m = host_.maps[k];
            foreach_n(s, half, [=] __device__(size_t j) {
                const E x = prev[j];
                const E d = F::mul_add(F::mul_add(m.den[2], x, m.den[1]), x, m.den[0]);
                if (F::is_zero(d)) atomicAdd(bad, 1ull);
                den[j] = d;
            });
            batch_inv(den, den, half, s);
            foreach_n(s, half, [=] __device__(size_t j) {
                const E x = prev[j];
                layer[j] = F::mul(F::mul_add(F::mul_add(m.num[2], x, m.num[1]), x, m.num[0]), den[j]);
            });
        }
        unsigned long long h = 0;
        ECFFT_HIP_TRY(hipMemcpyAsync(&h, bad, sizeof(h), hipMemcpyDeviceToHost, s));
        ECFFT_HIP_TRY(hipStreamSynchronize(s));
        temps_done();
        bad_points_ = h != 0;                                      // a leaf is a pole of its isogeny map: not a valid point set
        return h == 0 && hipGetLastError() == hipSuccess;
    }
    bool bad_points() const { return bad_points_; }
    bool upload_points(E*& fdev, hipStream_t s) {               // the caller hipFree()s fdev
        ECFFT_HIP_TRY(hipMalloc(&fdev, 2 * N_ * sizeof(E)));
        if (!points_on_device(fdev, s)) { (void)hipFree(fdev); fdev = nullptr; return false; }
        std::vector<E> den(2 * (L_ ? L_ : 1));
        for (unsigned k = 0; k < L_; ++k) { den[2 * k] = host_.maps[k].den[0]; den[2 * k + 1] = host_.maps[k].den[1]; }
        den_ = take(2 * (L_ ? L_ : 1));
        if (hipMemcpyAsync(den_, den.data(), den.size() * sizeof(E), hipMemcpyHostToDevice, s) != hipSuccess ||
            hipStreamSynchronize(s) != hipSuccess) {             // `den` is a stack vector; fdev must not leak on this path either
            (void)hipGetLastError(); (void)hipFree(fdev); fdev = nullptr; return false;
        }
        return true;
    }
    bool build_extend_shard(HostTree<F>&& ht, int device, unsigned log_p, unsigned rank) {
        host_ = std::move(ht);
        N_ = host_.n; L_ = ilog2(N_); device_ = device;
        const size_t c = (N_ / 2) >> log_p;
        if (L_ < 2 || c < ((size_t)1 << log_p) || c < 2 || rank >= ((size_t)1 << log_p)) return false;
        ECFFT_HIP_TRY(hipSetDevice(device_));
        hipStream_t s = nullptr;
        // arena: den + per parity {5 local tables + 4 cyclic tables + w + winv of the block and of the cyclic positions} of c
        // constants each + inner.  The point set f (all layers, 2N elements) is needed only while the tables are computed: a
        // temporary, freed before returning.
        size_t total = 64 + 2 * L_ + shard_set_elems(c) + 4096;
        ECFFT_HIP_TRY(hipMalloc(&arena_, total * sizeof(E)));
        arena_cap_ = total; arena_used_ = 0;
        E* fdev = nullptr;
        if (!upload_points(fdev, s)) return false;
        trees_.assign(L_ + 1, Tree{}); sets_.assign(L_ + 1, ShardSet{});
        const bool built = build_shard_set(L_, log_p, rank, fdev, s);
        const bool drained = hipStreamSynchronize(s) == hipSuccess;
        (void)hipFree(fdev);
        temps_free();
        if (!built || !drained) { fprintf(stderr, "ecfft: shard table build failed\n"); return false; }
        host_.f.clear(); host_.f.shrink_to_fit();                // the host copy of the point set is not needed either
        shard_kind_ = kShardExtend; shard_log_p_ = log_p; shard_rank_ = rank;
        return true;
    }

    // SHARDED ENTER context for ONE ENTER of n coefficients split over P = 2^log_p GPUs (api_enter_split): the full chain
    // T_1 .. T_c, c = n/P, for the rank-local low levels (every rank runs the same ENTER of its chunk on T_c), and for each of
    // the log P top levels m = c*Q only the rank's share of T_m: the EXTEND tables of the split over its half-group (a ShardSet
    // with 2^log_p' = Q/2, rank' = rank mod Q/2; Q = 2 is a local EXTEND = the "split" over one rank) and the c entries of
    // xnn_s = leaf^(m/2) its combine step reads (the positions i'*Q + a it owns in the level's cyclic order).  All pointwise in the point set: no tree above T_c is materialised anywhere.
    // ~56 c elements for the chain + 13 c constants per top level (ENTER only extends towards S1) instead of 56 n.
    bool build_enter_shard(HostTree<F>&& ht, int device, unsigned log_p, unsigned rank) {
        host_ = std::move(ht);
        N_ = host_.n; L_ = ilog2(N_); device_ = device;
        const size_t P = (size_t)1 << log_p, c = N_ >> log_p;
        if (log_p == 0 || L_ < 2 || c < 2 * P || rank >= P) return false;
        const unsigned lc = ilog2(c);
        ECFFT_HIP_TRY(hipSetDevice(device_));
        hipStream_t s = nullptr;
        size_t total = 64 + 2 * L_ + 4096;
        for (unsigned l = 0; l <= lc; ++l) total += (6 + 11 * kTeElems) * ((size_t)1 << l) + 1024 + blk16_elems(l);
        total += log_p * (shard_set_elems(c, 1) + c + 64);
        total += low16_elems();
        ECFFT_HIP_TRY(hipMalloc(&arena_, total * sizeof(E)));
        arena_cap_ = total; arena_used_ = 0;
        E* fdev = nullptr;
        if (!upload_points(fdev, s)) return false;
        struct Free { E* p; ~Free() { (void)hipFree(p); } } free_f{fdev};
        f_ = fdev;                                               // build_tree reads f_; reset below
        if (!ensure_scratch(c)) return false;
        create_side_streams();                                   // the rank-local ENTER / EXIT of the chunk runs the two-halves schedule too
        trees_.assign(L_ + 1, Tree{}); sets_.assign(L_ + 1, ShardSet{});
        for (unsigned l = 0; l <= lc; ++l) { if (!build_tree(l, s)) { f_ = nullptr; return false; } }
        if (!build_low16(lc, s)) { f_ = nullptr; return false; }
        for (size_t Q = 2; Q <= P; Q *= 2) {
            const size_t half = Q / 2, m = c * Q;
            const unsigned lm = ilog2(m);
            const size_t a = rank % Q, stride = N_ / m;
            if (!build_shard_set(lm, ilog2(half), (unsigned)(a % half), fdev, s, 1)) { f_ = nullptr; return false; }   // ENTER extends towards S1 only
            E* x = take(c); const E* f = fdev; const size_t N = N_; const uint64_t ex = m / 2;
            foreach_n(s, c, [=] __device__(size_t j) { x[j] = F::pow_u64(f[N + (j * Q + a) * stride], ex); });   // the rank's positions in the level's cyclic order
            trees_[lm].xnn = x;
            if (hipStreamSynchronize(s) != hipSuccess) { f_ = nullptr; return false; }
            temps_free();
        }
        f_ = nullptr;
        hipError_t err = hipGetLastError();
        if (err != hipSuccess) { fprintf(stderr, "ecfft: kernel launch failed: %s\n", hipGetErrorString(err)); return false; }
        ECFFT_HIP_TRY(hipStreamSynchronize(s));
        temps_free();
        host_.f.clear(); host_.f.shrink_to_fit();
        ECFFT_HIP_TRY(hipMalloc(&d_trees_, (L_ + 1) * sizeof(Tree)));
        ECFFT_HIP_TRY(hipMemcpy(d_trees_, trees_.data(), (L_ + 1) * sizeof(Tree), hipMemcpyHostToDevice));
        shard_kind_ = kShardEnter; shard_log_p_ = log_p; shard_rank_ = rank;
        return true;
    }

    // SHARDED EXIT context for ONE EXIT of n evaluations split over the P ranks of `tr` (api_exit_split) — a COLLECTIVE build: the
    // full chain T_1 .. T_c (c = n/P) for the rank-local low levels, and for each top level m = c*Q the rank's share of T_m: the
    // EXTEND tables of the split over its group of Q ranks (both directions, hc = c/2 entries per rank), its c entries of xnn_s
    // and 1/xnn_s, its hc entries of 1/z0_s1 (pointwise_z), and its c entries of z0z0_rem_xnn_s.  The last is (Z_0^2 mod X^(m/2))
    // on the leaves — a truncation in the monomial basis, not pointwise — and is built the way the reference builds it
    // (src/fftree.rs:418-452) but distributed, level by level, with the very operators the split EXIT consists of:
    //   zz0 = modular_reduce_{T_(m/2)}(z0z0' * z1z1') over the half-group (on the blocks of T_(m/2)'s tables from the level below),
    //   zz1 = EXTEND_{T_m}(zz0 -> S1) inside the half-group (a temporary one-direction share of T_m), one exchange re-blocks
    //   interleave(zz0, zz1) over the whole group, then two modular reductions on T_m over the group give the rank's blocks of
    //   z0z0_rem_xnn_s and z1z1_rem_xnn_s (the latter only feeds the next level).  No tree above T_c exists on any GPU.
    struct TempArena {           // to_tables() / take() allocate from a scratch region while this lives
        DeviceChain* ch; E* a; size_t cap, used; E* mine = nullptr;
        TempArena(DeviceChain* c, size_t elems) : ch(c), a(c->arena_), cap(c->arena_cap_), used(c->arena_used_) {
            if (hipMalloc(&mine, elems * sizeof(E)) != hipSuccess) { (void)hipGetLastError(); throw DeviceAllocError(); }
            ch->arena_ = mine; ch->arena_cap_ = elems; ch->arena_used_ = 0;
        }
        ~TempArena() { ch->arena_ = a; ch->arena_cap_ = cap; ch->arena_used_ = used; (void)hipFree(mine); }
    };
    // min_memory: never keep the full tree T_2c — the pair level then runs as four split EXTENDs (9 exchanges instead of 1) and the
    // context holds no tree above T_c at all.  Otherwise T_2c is kept when it fits this rank's free memory, and the ranks AGREE on
    // the form (one rank short of memory => every rank takes the split form): ADVICE r04.
    bool build_exit_shard(HostTree<F>&& ht, int device, Transport& tr, bool min_memory = false) {
        host_ = std::move(ht);
        N_ = host_.n; L_ = ilog2(N_); device_ = device;
        const size_t P = (size_t)tr.world;
        if (P < 2 || (P & (P - 1))) return false;
        const unsigned log_p = ilog2(P), rank = (unsigned)tr.rank;
        const size_t c = N_ >> log_p, hc = c / 2;
        if (L_ < 2 || c < 2 * P || hc < P || rank >= P) return false;
        const unsigned lc = ilog2(c);
        ECFFT_HIP_TRY(hipSetDevice(device_));
        hipStream_t s = nullptr;
        E* fdev = nullptr; E* lc_inv = nullptr;
        struct Free { E*& p; DeviceChain* ch; ~Free() { if (p) (void)hipFree(p); ch->f_ = nullptr; ch->ovr_tree_ = nullptr; ch->ovr_set_ = nullptr; } } free_f{fdev, this};
        // Form of the pair level (groups of two ranks), agreed before anything is allocated.  Redundant (round 4): the chain goes up
        // to T_2c and both ranks of a pair run the level on the whole 2c block from ONE exchange (api_exit_split) — for P = 2 the
        // context is then as large as a full one.  Split: no tree above T_c, the level is four split EXTENDs on the ranks' shares (9
        // exchanges) — the form for transforms whose tables exceed a GPU, chosen with min_memory or when T_2c does not fit the free
        // memory of ANY rank (Transport::vote is an AND over the ranks).
        auto arena_elems = [&](bool red) {
            size_t total = 64 + 3 * L_ + 4096;
            for (unsigned l = 0; l <= lc + (red ? 1u : 0u); ++l) total += (6 + 11 * kTeElems) * ((size_t)1 << l) + 1024 + blk16_elems(l);
            if (P > 2 || !red) total += log_p * (shard_set_elems(hc) + 5 * c + 256);   // P = 2, redundant: the pair level is the only top level — no share at all
            return total + low16_elems();
        };
        bool want_red = !min_memory;
        if (want_red) {
            size_t fr = 0, tt = 0;
            // arena + point set (2N) + transform scratch and pooled temporaries of the block level (~16 c) + headroom
            const size_t need = (arena_elems(true) + 2 * N_ + 16 * c) * sizeof(E) + ((size_t)256 << 20);
            if (hipMemGetInfo(&fr, &tt) != hipSuccess) { (void)hipGetLastError(); want_red = false; } else want_red = need <= fr;
        }
        if (const char* tr_rank = ab_env("ECFFT_TEST_PAIR_SPLIT_RANK")) if (atoi(tr_rank) == (int)rank) want_red = false;   // test builds: this rank "does not fit"
        const bool red = tr.vote(want_red, s);
        // everything up to the first exchange is LOCAL work that can fail on one rank only (allocations, the chain up to n/P):
        // the ranks agree on its outcome before any of them enters the collective part
        auto local_part = [&]() -> bool {
            try {
                const size_t total = arena_elems(red);
                ECFFT_HIP_TRY(hipMalloc(&arena_, total * sizeof(E)));
                arena_cap_ = total; arena_used_ = 0;
                if (!upload_points(fdev, s)) return false;
                f_ = fdev;                                               // build_tree reads f_
                if (!ensure_scratch(2 * c)) return false;
                create_side_streams();                                   // the rank-local ENTER / EXIT of the chunk runs the two-halves schedule too
                trees_.assign(L_ + 1, Tree{}); sets_.assign(L_ + 1, ShardSet{});
                for (unsigned l = 0; l <= lc + (red ? 1u : 0u); ++l) if (!build_tree(l, s)) return false;
                if (red) { pair_full_ = trees_[lc + 1]; have_pair_full_ = true; }   // the level-Q=2 iteration below re-points trees_[lc + 1] at the rank's shares (the distributed
                                                                         // build of the level above still splits T_2c over the pairs); the full tables stay in the arena
                if (!build_low16(lc, s)) return false;
                lc_inv = take(L_);
                std::vector<E> h(L_, F::one());
                for (unsigned k = 0; k < L_; ++k) h[k] = F::inv(host_.maps[k].num[2]);
                ECFFT_HIP_TRY(hipMemcpy(lc_inv, h.data(), L_ * sizeof(E), hipMemcpyHostToDevice));
                return !fail_next_collective_;
            } catch (const DeviceAllocError&) { return false; }
        };
        bool local_ok = local_part();
        fail_next_collective_ = false;
#ifdef ECFFT_TEST_HOOKS
        if (test_fail_build_rank().load() == (int)rank) local_ok = false;   // test hook (ecfft_test_fail_build_rank): this rank's local part "fails"
#endif
        if (!tr.vote(local_ok, s)) { fprintf(stderr, "ecfft: sharded EXIT build: the local part failed on %s rank\n", local_ok ? "another" : "this"); return false; }
        shard_kind_ = kShardExit; shard_log_p_ = log_p; shard_rank_ = rank;      // extend_split must read the shares from here on
        const E* f = fdev; const size_t N = N_;
        bool ok = true;
        E *pz0[2] = {nullptr, nullptr}, *pz1[2] = {nullptr, nullptr};     // the level below: z0z0 / z1z1 _rem_xnn_s on the rank's S0 / S1 positions
        for (size_t Q = 2; ok && Q <= P; Q *= 2) {
            if (P == 2 && red) break;                                      // redundant pair level: the shares of T_2c only feed the distributed build of the level above
            const size_t half = Q / 2, m = c * Q, stride = N_ / m;
            const unsigned lm = ilog2(m), lq = ilog2(Q), lh = ilog2(half);
            const int base = (int)((rank / Q) * Q), a = (int)rank - base, g = a / (int)half, ap = a % (int)half, subbase = base + g * (int)half;
            // Every length-m/2 vector of the level is CYCLIC over the group: entry j of the rank = position j*Q + a (api_exit_split).
            // ---- permanent: the rank's share of T_m for the level's split EXTENDs + its pointwise entries, compact
            bool lvl_ok = false;
            try { lvl_ok = build_shard_set(lm, lq, (unsigned)a, fdev, s, -1); } catch (const DeviceAllocError&) { lvl_ok = false; }
            if (!tr.vote(lvl_ok, s)) return false;                         // the level's exchanges follow: all ranks or none
            Tree& T = trees_[lm];
            E *xe = temp(hc), *xei = take(hc), *xo = take(hc), *zib = take(hc), *c0 = take(hc), *c1 = take(hc), *q0 = take(hc), *q1 = take(hc);
            { const uint64_t ex = m / 2; const size_t qq = Q, aa = (size_t)a;
              foreach_n(s, hc, [=] __device__(size_t j) { const size_t i = j * qq + aa; xe[j] = F::pow_u64(f[N + (2 * i) * stride], ex); xo[j] = F::pow_u64(f[N + (2 * i + 1) * stride], ex); }); }
            batch_inv(xe, xei, hc, s);
            E* zb = temp(hc);                                                      // z0_s1 on the rank's S1 positions (plain), also used below
            pointwise_z(lm, 0, (size_t)a, hc, zb, fdev, lc_inv, s, Q);
            batch_inv(zb, zib, hc, s);
            // in an EXIT-shard context these fields of a top tree are the rank's COMPACT views: 1/xnn_s on its S0 positions, xnn_s on
            // its S1 positions, 1/z0_s1, z0z0_rem_xnn_s on its S0 positions (z0z0) and on its S1 positions (z1z1 field)
            T.xnn_inv = xei; T.xnn = xo; T.z0_inv_s1 = zib; T.z0z0 = c0; T.z1z1 = c1;
            // ---- zz0 = modular_reduce_{T_(m/2)}(z0z0' * z1z1', c = z0z0') (:421-425), then as a length-m/2 vector cyclic over the HALF-group
            E *Z = temp(c), *ZZ1 = temp(c);
            E *e0 = temp(hc), *e1 = temp(hc), *h0 = temp(hc), *h1 = temp(hc), *t0 = temp(hc), *x0 = temp(hc), *x1 = temp(hc), *A = temp(c), *B = temp(c);
            if (half == 1) {
                const Tree& S = trees_[lc];
                E* sq = temp(c);
                ew_mul(sq, S.z0z0, S.z1z1, c, s);
                { const E *xi = S.xnn_inv, *x = S.xnn; foreach_n(s, hc, [=] __device__(size_t i) { e0[i] = xi[2 * i]; e1[i] = x[2 * i + 1]; }); }
                b_modular_reduce(lc, sq, e0, e1, S.z0z0, Z, s);                      // natural order = cyclic over one rank
            } else {
                const Tree& S = trees_[lm - 1];                                   // compact views of T_(m/2), built one level down (rank ap of the half-group)
                { const E *a0 = pz0[0], *a1 = pz0[1], *b0 = pz1[0], *b1 = pz1[1]; foreach_n(s, hc, [=] __device__(size_t j) { e0[j] = F::mul(a0[j], b0[j]); e1[j] = F::mul(a1[j], b1[j]); }); }
                ok = modred_split(tr, subbase, lh, m / 2, e0, e1, h0, h1, S.xnn_inv, S.xnn, 1, S.z0_inv_s1, 1, S.z0z0, S.z1z1, 1, t0, x0, x1, A, B, s, true);
                if (!ok) break;
                // (h0, h1)[j] = zz0 at index i = j*Q + 2*ap + {0, 1} of the length-m/2 vector; as a vector cyclic over the half-group,
                // index i sits on sub-rank i mod half at slot i / half = 2j + t, t = (2*ap + b) / half
                const int hq = (int)half;
                P2P snd[2] = {{subbase + (2 * ap) % hq, h0, hc * sizeof(E)}, {subbase + (2 * ap + 1) % hq, h1, hc * sizeof(E)}};
                const int par = ap & 1, s1 = (ap - par) / 2, s2 = s1 + hq / 2;       // the two senders whose (2*a' + par) mod half == ap
                P2P rcv[2] = {{subbase + s1, e0, hc * sizeof(E)}, {subbase + s2, e1, hc * sizeof(E)}};
                ok = tr.exchange(snd, 2, rcv, 2, s);
                if (!ok) break;
                const size_t ta = (size_t)((2 * s1 + par) / hq), tb = (size_t)((2 * s2 + par) / hq);
                foreach_n(s, hc, [=] __device__(size_t j) { Z[2 * j + ta] = e0[j]; Z[2 * j + tb] = e1[j]; });
            }
            // ---- zz1 = EXTEND_{T_m}(zz0 -> S1) inside the half-group, c entries per rank, cyclic  (:426)
            {
                TempArena ta(this, shard_set_elems(c, 1) + 4096);
                Tree tt{}; ShardSet ts{};
                if (!build_shard_set(lm, lh, (unsigned)ap, fdev, s, 1, &tt, &ts)) return false;
                ovr_tree_ = &tt; ovr_set_ = &ts;
                ok = half == 1 ? extend(Z, ZZ1, c, 1, 1, s) : extend_split(tr, subbase, lh, Z, ZZ1, m / 2, 1, s, A, B, true, true);
                ok = (hipStreamSynchronize(s) == hipSuccess) && ok;
                ovr_tree_ = nullptr; ovr_set_ = nullptr;
            }
            if (!ok) break;
            // ---- zz = interleave(zz0, zz1) (:427-429) on the rank's positions 2*(j*Q + a) + {0, 1}: index i = j*Q + a of zz0 / zz1 sits
            // on this very rank (i mod half = ap) at slot i / half = 2j + g — no exchange
            E *zc0 = temp(hc), *zc1 = temp(hc);
            { const size_t gg = (size_t)g; foreach_n(s, hc, [=] __device__(size_t j) { zc0[j] = Z[2 * j + gg]; zc1[j] = ZZ1[2 * j + gg]; }); }
            // ---- z0z0_rem_xnn_s on the rank's positions  (:430-446)
            E *xqe = temp(hc), *xqo = temp(hc), *xqei = temp(hc), *xqoi = temp(hc);
            { const uint64_t ex = m / 4; const size_t qq = Q, aa = (size_t)a;
              foreach_n(s, hc, [=] __device__(size_t j) { const size_t i = j * qq + aa; xqe[j] = F::pow_u64(f[N + (2 * i) * stride], ex); xqo[j] = F::pow_u64(f[N + (2 * i + 1) * stride], ex); }); }
            batch_inv(xqe, xqei, hc, s); batch_inv(xqo, xqoi, hc, s);
            foreach_n(s, hc, [=] __device__(size_t j) {                          // even leaf: z0 = 0; odd leaf: z0 = z0_s1
                e0[j] = F::mul(F::sub(F::sqr(xe[j]), zc0[j]), xqei[j]);
                e1[j] = F::mul(F::sub(F::sqr(F::sub(zb[j], xo[j])), zc1[j]), xqoi[j]);
            });
            ok = modred_split(tr, base, lq, m, e0, e1, h0, h1, xqei, xqo, 1, zib, 1, zc0, zc1, 1, t0, x0, x1, A, B, s, true);
            if (!ok) break;
            foreach_n(s, hc, [=] __device__(size_t j) { c0[j] = F::mul_add(xqe[j], h0[j], zc0[j]); c1[j] = F::mul_add(xqo[j], h1[j], zc1[j]); });
            // ---- z1z1_rem_xnn_s on the rank's positions (:449-452): only the next level's zz0 reads it
            E* z1s = temp(hc);
            pointwise_z(lm, 1, (size_t)a, hc, z1s, fdev, lc_inv, s, Q);
            foreach_n(s, hc, [=] __device__(size_t j) { e0[j] = F::sqr(F::sub(z1s[j], xe[j])); e1[j] = F::sqr(xo[j]); });
            ok = modred_split(tr, base, lq, m, e0, e1, q0, q1, xei, xo, 1, zib, 1, c0, c1, 1, t0, x0, x1, A, B, s, true);
            if (!ok) break;
            pz0[0] = c0; pz0[1] = c1; pz1[0] = q0; pz1[1] = q1;
            if (hipStreamSynchronize(s) != hipSuccess) { ok = false; break; }
            temps_free();
        }
        ok = ok && hipGetLastError() == hipSuccess && hipStreamSynchronize(s) == hipSuccess;
        temps_free();
        ok = tr.vote(ok, s);                                                // every rank reports the same outcome
        if (!ok) { fprintf(stderr, "ecfft: sharded EXIT table build failed\n"); return false; }
        host_.f.clear(); host_.f.shrink_to_fit();
        ECFFT_HIP_TRY(hipMalloc(&d_trees_, (L_ + 1) * sizeof(Tree)));
        ECFFT_HIP_TRY(hipMemcpy(d_trees_, trees_.data(), (L_ + 1) * sizeof(Tree), hipMemcpyHostToDevice));
        return true;
    }

    // ------------------------------------------------------------------------------------------
    // EXTEND core: all 2*log(e) normalised stages on `total` elements = count vectors of length
    // e = m/2 laid end to end, as a chain of fused passes:
    //     [column passes: top decompose stages, <= 4 per pass] -> row pass (every stage with
    //     2h <= tile, decompose then recombine) -> [column passes: top recombine stages].
    // `io` describes where the first pass loads from (with an optional fused pointwise op) and what
    // the last pass does with its result; intermediate passes run in place on `buf`.
    // srcpar = parity of the moiety the data lives on.
    // ------------------------------------------------------------------------------------------
#ifndef ECFFT_LOG_TILE_BYTES
#define ECFFT_LOG_TILE_BYTES 15
#endif
#ifndef ECFFT_CT_ALL
#define ECFFT_CT_ALL 0      // 1: compile-time tile sizes for every field (default: 4-byte fields only)
#endif
    static constexpr unsigned kLogTileMax = (sizeof(E) == 32) ? ECFFT_LOG_TILE_BYTES - 5 : ECFFT_LOG_TILE_BYTES - 2;   // 32 KiB LDS tiles by default (A/B on MI355X: 512 threads x 32 KiB beat 64 KiB tiles by ~5%)
#ifndef ECFFT_COL_STAGES
#define ECFFT_COL_STAGES 9      // round 5: 9 (was 8) together with ECFFT_COL_MIN_LOGC 1 — the nine column stages of an EXTEND of 2^19 (every core of
#endif                          // level 20) run as ONE pass on tiles of 2^9 rows x 2 elements instead of two passes of 5 + 4 stages: 2^20 -1.4 %, 2^21 -0.8 %
#ifndef ECFFT_LOG_COL_TILE_BYTES
#define ECFFT_LOG_COL_TILE_BYTES 15
#endif
#ifndef ECFFT_COL_MIN_LOGC
#define ECFFT_COL_MIN_LOGC 1     // log2 of the shortest column-tile row of a full-size launch (2: rows of >= 4 elements = 128 B, rounds 1-4; 1: 64-byte rows
                                 // are accepted where they save a whole pass, i.e. for exactly nine column stages)
#endif
#ifndef ECFFT_COL_STAGES_4B
#define ECFFT_COL_STAGES_4B 9
#endif
    // max stages per column pass (A/B on MI355X: 8 stages x 4-element rows beat 5 x 32 on secp256k1; 9 x 16-element rows are 1 %
    // better than 8 on M31 at 2^24 and equal below, profiles/r02/knob_sweep.txt)
    static constexpr unsigned kColStages = sizeof(E) == 4 ? ECFFT_COL_STAGES_4B : ECFFT_COL_STAGES;
    static constexpr unsigned kLogColTileMax = (sizeof(E) == 32) ? ECFFT_LOG_COL_TILE_BYTES - 5 : ECFFT_LOG_COL_TILE_BYTES - 2;
    // Fusion of consecutive cores (EXIT): `next_ld` != nullptr says that another core of the same tree and size, opposite
    // direction, follows on `buf` with that load operator; if this core ends in a column pass, that pass and the next
    // core's first column pass run as ONE launch (k_stages_col_mid) and the function returns true; the next core is then
    // called with skip_first_col = true.
    struct NextLoad { int ld_mode; const TE* ld_tbl; double extra_first; };
    // ENTER fusion: `ef` != nullptr says that this core is the EXTEND of an ENTER level (total = whole [u0 | v0] blocks, target
    // S1) and that its LAST pass must also do the level's combine (src/fftree.rs:155-159) and write the level's output to
    // ef->dst — k_stages_col_enter when the core ends in a column pass, the row kernel's ST_ENTER store operator otherwise.
    struct EnterFuse { const E* src; E* dst; double extra; };
    bool extend_core(unsigned log_m, IoDesc<F> io, E* buf, size_t total, int srcpar, hipStream_t s,
                     double extra_first = 0.0, double extra_last = 0.0, unsigned k_begin = 0,
                     const NextLoad* next_ld = nullptr, bool skip_first_col = false, const EnterFuse* ef = nullptr) const {
        // k_begin > 0: only stages k >= k_begin (block-distributed shard of a split EXTEND)
        const Tree& T = tree_at(log_m);
        size_t e = T.e; unsigned le = ilog2(e);
        int tgt = 1 - srcpar;
        unsigned tz = (unsigned)__builtin_ctzll((unsigned long long)total);   // tiles must divide count*e
        unsigned log_tile = tz < kLogTileMax ? tz : kLogTileMax;
        const bool small = small_launch(total) && !ef_small_off_;           // latency regime: 4x smaller tiles, 4x as many workgroups
        if (small && log_tile > kLogLowSmall) log_tile = kLogLowSmall;
        unsigned k_first = le > log_tile ? le - log_tile : 0;                 // first stage with 2h <= tile
        if (k_first < k_begin) k_first = k_begin;
        if (ef && k_first == 0 && le + 1 > log_tile) log_tile = le + 1;       // ST_ENTER wants whole [U | V] blocks in the tile (64 KiB at e = tile)
        // pass list: (kind, ka, kb)
        struct Pass { int kind; unsigned ka, kb; };                           // kind 0 col-decompose, 1 row, 2 col-recombine
        Pass passes[2 * 8 + 1]; int np = 0;
        if (k_first > k_begin) {                                              // balanced groups of <= kColStages stages
            unsigned log_ct0 = tz < kLogColTileMax ? tz : kLogColTileMax;
            if (small && log_ct0 > kLogLowSmall) log_ct0 = kLogLowSmall;
            const unsigned minc = (small && log_ct0 == kLogLowSmall) ? small_min_logc_ : (unsigned)ECFFT_COL_MIN_LOGC;
            unsigned rmax = kColStages < log_ct0 - minc ? kColStages : log_ct0 - minc;    // keep rows >= 4 elements (128 B)
            if (rmax < 1) rmax = 1;
            unsigned ncol = k_first - k_begin, ngrp = (ncol + rmax - 1) / rmax;
            unsigned k = k_begin;
            for (unsigned g = 0; g < ngrp; ++g) { unsigned sz = ncol / ngrp + (g < ncol % ngrp ? 1 : 0); passes[np++] = {0, k, k + sz - 1}; k += sz; }
        }
        int nd = np;
        passes[np++] = {1, k_first, le};
        for (int g = nd - 1; g >= 0; --g) passes[np++] = {2, passes[g].ka, passes[g].kb};
        // results of launches of at most 2^kNtStoreMaxLog elements leave with non-temporal stores (kernels.h data_st; 32-byte fields)
        const uint32_t nt_st = (sizeof(E) == 32 && kNtStoreMaxLog && total <= ((size_t)1 << kNtStoreMaxLog)) ? 1u : 0u;
        const E* plain_src = buf;
        const bool fuse_tail = next_ld && nd >= 1 && (io.st_mode == ST_PLAIN || io.st_mode == ST_SCALE || io.st_mode == ST_AXPBY) && io.dst == buf;
        const int pi0 = (skip_first_col && nd >= 1) ? 1 : 0;
        for (int pi = pi0; pi < np; ++pi) {
            IoDesc<F> d;
            bool first = pi == 0, last = pi == np - 1;
            if (last && fuse_tail) {
                // this core's last recombine group + the next core's first decompose group (same stages, same tiles)
                const Pass& P = passes[pi];
                d = io; d.src = buf; d.dst = buf; d.ld_mode = next_ld->ld_mode; d.ld_tbl = next_ld->ld_tbl; d.nt_st = nt_st;
                unsigned log_ct = tz < kLogColTileMax ? tz : kLogColTileMax;
                if (small && log_ct > kLogLowSmall) log_ct = kLogLowSmall;
                unsigned R = P.kb - P.ka + 1, log_c = log_ct - R;
                double hsum = 0; for (unsigned k = P.ka; k <= P.kb; ++k) hsum += (double)(e >> (k + 1));
                double bytes = 2.0 * sizeof(E) * (2.0 * R * total + 4.0 * hsum * tblw_) + extra_last + next_ld->extra_first;
                const unsigned lv = pair_spans(total, le, P.ka, log_ct, d);
                if (col256_ok(log_ct, le) && T.c0t[tgt])      // latency regime: one element per thread in registers (reg_col_stages)
                    ECFFT_LAUNCH(KC_COL, bytes, k_stages_col_mid256<F>, dim3((unsigned)(total >> log_ct)), dim3(256), 0, s, d, T.p0[tgt], T.p1[tgt], T.c0t[tgt], T.dinv[tgt], le, P.ka, P.kb, log_c);
                else
                ECFFT_LAUNCH(KC_COL, bytes, k_stages_col_mid<F>, dim3((unsigned)(total >> (log_ct + lv))), dim3(kBlockLds), (sizeof(E) * ((size_t)col_row_stride<E>(1u << log_c) << R)) << lv, s,
                             d, T.p0[tgt], T.p1[tgt], T.np0[tgt], T.dinv[tgt], le, P.ka, P.kb, log_c, T.c0t[tgt], (uint32_t)lv);
                return true;
            }
            if (last && ef && passes[pi].kind == 2) {
                // last recombine group (stages kb..0) + the ENTER level's combine, two vectors per workgroup
                const Pass& P = passes[pi];
                unsigned log_ct = le < kLogColTileMax ? le : kLogColTileMax;
                if (small && log_ct > kLogLowSmall) log_ct = kLogLowSmall;
                unsigned R = P.kb + 1, log_c = log_ct - R;
                double hsum = 0; for (unsigned k = P.ka; k <= P.kb; ++k) hsum += (double)(e >> (k + 1));
                double bytes = sizeof(E) * (2.0 * R * total + 4.0 * hsum * tblw_) + ef->extra;
                if (col256_ok(log_ct, le))
                    ECFFT_LAUNCH(KC_COL, bytes, k_stages_col_enter256<F>, dim3((unsigned)(total >> (log_ct + 1))), dim3(512), 0, s,
                                 (const E*)buf, ef->src, ef->dst, T.p0[tgt], T.p1[tgt], T.xe, T.w[1], T.w1x, le, P.kb, log_c);
                else
                ECFFT_LAUNCH(KC_COL, bytes, k_stages_col_enter<F>, dim3((unsigned)(total >> (log_ct + 1))), dim3(kBlockLds),
                             2 * sizeof(E) * ((size_t)col_row_stride<E>(1u << log_c) << R), s,
                             (const E*)buf, ef->src, ef->dst, T.p0[tgt], T.p1[tgt], T.xe, T.w[1], T.w1x, le, P.kb, log_c);
                return false;
            }
            // load side
            if (first) { d = io; d.st_tr_logp = 0; } else { d = IoDesc<F>{}; d.src = plain_src; d.src_stride = 1; d.src_off = 0; d.ld_mode = LD_PLAIN; d.ld_tbl = nullptr; }
            // store side
            if (last) { d.dst = io.dst; d.st_mode = io.st_mode; d.st_a = io.st_a; d.st_b = io.st_b; d.aux = io.aux; d.aux_stride = io.aux_stride; d.aux_off = io.aux_off; d.aux_out = io.aux_out;
                        d.st_tr_logp = io.st_tr_logp; d.tr_chunk = io.tr_chunk; }
            else { d.dst = buf; d.st_mode = ST_PLAIN; d.st_a = d.st_b = nullptr; d.aux = nullptr; d.aux_stride = d.aux_off = 0; d.aux_out = nullptr; }
            d.nt_st = nt_st;
            double extra = (first ? extra_first : 0.0) + (last ? extra_last : 0.0);
            const Pass& P = passes[pi];
            if (last && ef) {   // row pass is the last one: pair store operator
                d.dst = ef->dst; d.st_mode = ST_ENTER; d.st_a = T.xe; d.st_b = T.w1x; d.st_c = T.w[1]; d.aux = ef->src; d.aux_stride = 1; d.aux_off = 0; d.aux_out = nullptr;
                extra += ef->extra;
            }
            if (P.kind == 1) {
                unsigned nst = le - k_first;
                double hsum = (double)((e >> k_first) - 1);                   // sum of h over the fused stages
                double bytes = sizeof(E) * (2.0 * nst * 2.0 * total + 8.0 * hsum * tblw_) + extra;
                // compile-time tiles of 4-byte fields run the register-resident stage engine, which works on chunks of 16 elements
                // per thread: only valid when the default tile IS such a chunk (non-default ECFFT_LOG_TILE_BYTES / ECFFT_BLOCK_ROW
                // builds fall back to the run-time-tile kernel)
                constexpr bool ct_row = sizeof(E) == 4 ? ((size_t)1 << kLogTileMax) == (size_t)kBlockRow * 16 : (ECFFT_CT_ALL != 0);
                // matrix-core form of the stages with pair distance <= 8 (mfma_blk16.h): whole 1024-element sub-tiles, >= 4 in-tile stages
                const bool use_blk16 = sizeof(E) == 32 && kBlockRow == 512 && kLogTileMax == 10 && !mfma_off_ && T.blk16_A[srcpar] && log_tile == kLogTileMax && le - k_first >= 4;
                const uint8_t* bA = use_blk16 ? T.blk16_A[srcpar] : nullptr;
                const unsigned long long* bK = use_blk16 ? T.blk16_K[srcpar] : nullptr;
                if (sizeof(E) == 32 && log_tile == kLogLowSmall && !row256_off_ && d.st_mode != ST_ENTER && T.c0t[srcpar]) {
                    // latency regime: one element per thread in registers; round 4: the stages with pair distance <= 8 on the matrix
                    // cores here too (v_mfma_i32_16x16x64_i8: the 16 blocks of a 256-element tile are one MFMA's columns)
                    const bool mf = !mfma_off_ && T.blk16_A[srcpar] && le >= 4 && le - k_first >= 4;
                    ECFFT_LAUNCH(KC_ROW, bytes, (k_stages_row256<F>), dim3((unsigned)(total >> kLogLowSmall)), dim3(256), 0, s, d, T.dinv[srcpar], T.p0[tgt], T.p1[tgt], T.inner[srcpar],
                                 le, k_first, T.c0t[srcpar], mf ? T.blk16_A[srcpar] : (const uint8_t*)nullptr, mf ? T.blk16_K[srcpar] : (const unsigned long long*)nullptr);
                } else
                if (log_tile == kLogTileMax + 1 && ct_row)
                    ECFFT_LAUNCH(KC_ROW, bytes, (k_stages_lds<F, (int)kLogTileMax + 1>), dim3((unsigned)(total >> log_tile)), dim3(kBlockRow),
                                 ((size_t)sizeof(E)) << log_tile, s, d, T.np0[srcpar], T.dinv[srcpar], T.p0[tgt], T.p1[tgt], T.inner[srcpar], le, k_first, log_tile, T.c0t[srcpar], bA, bK);
                else if (log_tile == kLogTileMax && (ct_row || bA))   // compile-time tile (always for the matrix-core row passes): +16% on M31 (8 pairs/thread unroll), -3% on secp256k1
                    ECFFT_LAUNCH(KC_ROW, bytes, (k_stages_lds<F, (int)kLogTileMax>), dim3((unsigned)(total >> log_tile)), dim3(kBlockRow),
                                 ((size_t)sizeof(E)) << log_tile, s, d, T.np0[srcpar], T.dinv[srcpar], T.p0[tgt], T.p1[tgt], T.inner[srcpar], le, k_first, log_tile, T.c0t[srcpar], bA, bK);
                else
                    ECFFT_LAUNCH(KC_ROW, bytes, (k_stages_lds<F, 0>), dim3((unsigned)(total >> log_tile)), dim3(kBlockRow),
                                 ((size_t)sizeof(E)) << log_tile, s, d, T.np0[srcpar], T.dinv[srcpar], T.p0[tgt], T.p1[tgt], T.inner[srcpar], le, k_first, log_tile, T.c0t[srcpar], bA, bK);
            } else {
                unsigned log_ct = tz < kLogColTileMax ? tz : kLogColTileMax;           // column tiles may be larger than row tiles
                if (small && log_ct > kLogLowSmall) log_ct = kLogLowSmall;
                unsigned R = P.kb - P.ka + 1, log_c = log_ct - R;
                double hsum = 0; for (unsigned k = P.ka; k <= P.kb; ++k) hsum += (double)(e >> (k + 1));
                double bytes = sizeof(E) * (2.0 * R * total + 4.0 * hsum * tblw_) + extra;
                const bool ct = (log_ct == kLogColTileMax && (sizeof(E) == 4 || ECFFT_CT_ALL));
                const unsigned lv = ct ? pair_spans(total, le, P.ka, log_ct, d) : 0;
                dim3 grid((unsigned)(total >> (log_ct + lv))); size_t lds = (sizeof(E) * ((size_t)col_row_stride<E>(1u << log_c) << R)) << lv;
                // compile-time column tiles exist for 4-byte fields only (or with ECFFT_CT_ALL): the 32-byte instantiation is not even
                // compiled — it was dead code with 24 B of scratch in the library's code object
                constexpr bool kCtCol = sizeof(E) == 4 || ECFFT_CT_ALL;
                if (col256_ok(log_ct, le) && lv == 0 && (P.kind != 0 || T.c0t[srcpar])) {
                    if (P.kind == 0) ECFFT_LAUNCH(KC_COL, bytes, (k_stages_col256<F, true>), dim3((unsigned)(total >> log_ct)), dim3(256), 0, s, d, T.c0t[srcpar], T.dinv[srcpar], le, P.ka, P.kb, log_c);
                    else ECFFT_LAUNCH(KC_COL, bytes, (k_stages_col256<F, false>), dim3((unsigned)(total >> log_ct)), dim3(256), 0, s, d, T.p0[tgt], T.p1[tgt], le, P.ka, P.kb, log_c);
                } else
                if (P.kind == 0) {
                    if constexpr (kCtCol) { if (ct) { ECFFT_LAUNCH(KC_COL, bytes, (k_stages_col<F, true, (int)kLogColTileMax>), grid, dim3(kBlockLds), lds, s, d, T.np0[srcpar], T.dinv[srcpar], le, P.ka, P.kb, log_c, T.c0t[srcpar], (uint32_t)lv); continue; } }
                    ECFFT_LAUNCH(KC_COL, bytes, (k_stages_col<F, true, 0>), grid, dim3(kBlockLds), lds, s, d, T.np0[srcpar], T.dinv[srcpar], le, P.ka, P.kb, log_c, T.c0t[srcpar], (uint32_t)lv);
                } else {
                    if constexpr (kCtCol) { if (ct) { ECFFT_LAUNCH(KC_COL, bytes, (k_stages_col<F, false, (int)kLogColTileMax>), grid, dim3(kBlockLds), lds, s, d, T.p0[tgt], T.p1[tgt], le, P.ka, P.kb, log_c, (const TE*)nullptr, (uint32_t)lv); continue; } }
                    ECFFT_LAUNCH(KC_COL, bytes, (k_stages_col<F, false, 0>), grid, dim3(kBlockLds), lds, s, d, T.p0[tgt], T.p1[tgt], le, P.ka, P.kb, log_c, (const TE*)nullptr, (uint32_t)lv);
                }
            }
        }
        return false;
    }
    // 4-byte fields, full-size column tiles: two consecutive spans (2h_ka-blocks; they read the same table entries) per workgroup
    // when the number of spans is even and the halved grid still fills the chip twice over
    static unsigned pair_spans(size_t total, unsigned le, unsigned ka, unsigned log_ct, const IoDesc<F>& d) {
        if (sizeof(E) != 4 || ECFFT_COL_PAD != 0 || log_ct != kLogColTileMax) return 0;
        if (((size_t)1 << log_ct) != (size_t)kBlockLds * 16) return 0;    // the kernels' vector path (kFast) exists for 16 elements per thread only
        // the paired form exists only on the kernels' 16-byte vector path: every buffer the pass touches must be 16-byte aligned
        // (user buffers may be only element-aligned) and the operator must be one the vector path implements
        auto al = [](const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; };
        if (!al(d.src) || !al(d.dst) || !al(d.aux) || !al(d.aux_out) || d.ld_tr_logp || d.st_tr_logp || le < 2) return 0;
        if (!((d.src_stride == 1 && d.src_off == 0) || (d.src_stride == 2 && d.src_off < 2))) return 0;
        const size_t nspans = total >> (le - ka);
        return (nspans >= 2 && (nspans & 1) == 0 && (total >> (log_ct + 1)) >= 512) ? 1u : 0u;
    }
    // latency regime, 32-byte fields: 256-element column tiles run on the register-resident engine (k_stages_col256 & co.)
    bool col256_ok(unsigned log_ct, unsigned le) const { return sizeof(E) == 32 && log_ct == kLogLowSmall && le < 31 && !col256_off_; }
    static IoDesc<F> io_plain(const E* src, E* dst) {
        IoDesc<F> d{}; d.src = src; d.src_stride = 1; d.src_off = 0; d.ld_mode = LD_PLAIN; d.ld_tbl = nullptr;
        d.dst = dst; d.st_mode = ST_PLAIN; d.st_a = d.st_b = nullptr; d.aux = nullptr; d.aux_stride = d.aux_off = 0; d.aux_out = nullptr;
        return d;
    }

    // FFTree::extend (src/fftree.rs:123-126) on `count` vectors of length e: uses T_{2e}; `target`
    // names the TARGET moiety.  in/out device pointers (may alias).
    bool extend(const E* in, E* out, size_t e, size_t count, int target, hipStream_t s) const {
        unsigned log_m = ilog2(e) + 1;
        const Tree& T = tree_at(log_m);
        size_t total = e * count; int src = 1 - target;
        IoDesc<F> io = io_plain(in, out);
        io.ld_mode = LD_SCALE; io.ld_tbl = T.winv[src];
        io.st_mode = ST_SCALE; io.st_a = T.w[target];
        extend_core(log_m, io, out, total, src, s);
        return hipGetLastError() == hipSuccess;
    }

    // ecfft_extend with count > 1 (the low-degree extension of many columns): like a batched ENTER / EXIT, a batch whose parts have at
    // least 2^kSplitMinLog elements runs as concurrent parts of whole vectors, one stream each (round 6, batch_ways / run_ways)
    bool extend_api(const E* in, E* out, size_t e, size_t count, int target, hipStream_t s) {
        const size_t total = e * count;
        // only for LONG vectors (e >= 2^kSplitMinLog): their launches are one or two rounds of tiles, which is what a second stream fills
        // (-2.5 .. -4.4 % at e = 2^19 .. 2^22); many short vectors are deep launches already (2^16 x 32: +1.3 %, profiles/r06/extend_split_ab.txt)
        const int ways = (e >> kSplitMinLog) == 0 ? 1 : batch_ways(total, count);
        if (ways <= 1) return extend(in, out, e, count, target, s);
        const size_t part = total / (size_t)ways, cnt = count / (size_t)ways;
        bool ok = true;
        run_ways(ways, s, [&](int i, hipStream_t si) { ok = extend(in + part * i, out + part * i, e, cnt, target, si) && ok; });
        return ok;
    }

    // ------------------------------------------------------------------------------------------
    // Building blocks of ONE EXTEND split over P = 2^log_p GPUs.  The vector of
    // length e lives on T_{2e}; `target` is the target moiety.
    //   cyclic shard: local element j' <-> global position j'*P + rank, local length e/P; stage k < log
    //     e - log_p pairs (j', j' + h_k/P) and reads table entry (j' mod h_k/P)*P + rank.
    //   block shard: local element j' <-> global position rank*e/P + j'; stages k >= log_p are local and
    //     use exactly the single-GPU kernels (table index = local pair index mod h).
    // ------------------------------------------------------------------------------------------
    // cyclic shard, decompose side: multiply by 1/W_src, then stages 0 .. log_p-1
    // cyclic shard, recombine side: stages log_p-1 .. 0, then multiply by W_tgt
    void extend_top_cyclic(E* buf, size_t e, int target, unsigned log_p, unsigned rank, bool recombine, hipStream_t s) const {
        unsigned log_m = ilog2(e) + 1;
        const Tree& T = trees_[log_m];
        int src = 1 - target;
        size_t P = (size_t)1 << log_p, el = e >> log_p, npairs = el / 2;
        if (!recombine) {
            ECFFT_LAUNCH(KC_POINTWISE, 0.0, k_scale_by_table<F>, dim3(nblocks(el)), dim3(kBlock), 0, s, buf, (const E*)buf, T.winv[src], el - 1, el, (uint32_t)P, rank);
            for (unsigned k = 0; k < log_p; ++k) {
                size_t h = e >> (k + 1), off = e - 2 * h;
                ECFFT_LAUNCH(KC_DECOMPOSE, sizeof(E) * (2.0 * el + 4.0 * (h >> log_p)), k_decompose_stage<F>, dim3(nblocks(npairs)), dim3(kBlock), 0, s,
                             buf, T.np0[src] + off, T.dinv[src] + off, ilog2(h >> log_p), npairs, (uint32_t)P, rank);
            }
        } else {
            for (unsigned k = log_p; k-- > 0;) {
                size_t h = e >> (k + 1), off = e - 2 * h;
                ECFFT_LAUNCH(KC_RECOMBINE, sizeof(E) * (2.0 * el + 4.0 * (h >> log_p)), k_recombine_stage<F>, dim3(nblocks(npairs)), dim3(kBlock), 0, s,
                             buf, T.p0[target] + off, T.p1[target] + off, ilog2(h >> log_p), npairs, (uint32_t)P, rank);
            }
            ECFFT_LAUNCH(KC_POINTWISE, 0.0, k_scale_by_table<F>, dim3(nblocks(el)), dim3(kBlock), 0, s, buf, (const E*)buf, T.w[target], el - 1, el, (uint32_t)P, rank);
        }
    }
    // block shard: decompose stages k >= log_p then recombine stages back down to log_p, fused passes, in place
    void extend_local_block(E* buf, size_t e, int target, unsigned log_p, hipStream_t s) const {
        unsigned log_m = ilog2(e) + 1;
        extend_core(log_m, io_plain(buf, buf), buf, e >> log_p, 1 - target, s, 0.0, 0.0, log_p);
    }

    // ------------------------------------------------------------------------------------------
    // ONE transform split over the ranks of a Transport (one process per GPU).
    // Everything below the exchange calls is the single-GPU kernels; the block <-> cyclic re-distributions cost no pass of
    // their own: the pack / unpack index maps ride on the 1/W and W scalings that an EXTEND needs anyway and on the
    // load / store operators (IoDesc::ld_tr_logp / st_tr_logp) of the block-local fused passes.
    // ------------------------------------------------------------------------------------------
    // equal pieces of `piece` elements to / from every rank of the group [gbase, gbase + P)
    bool exchange_group(Transport& tr, int gbase, size_t P, E* from, E* to, size_t piece, hipStream_t s) const {
        P2P snd[64], rcv[64];
        if (P > 64) return false;
        for (size_t q = 0; q < P; ++q) { snd[q] = {gbase + (int)q, from + q * piece, piece * sizeof(E)}; rcv[q] = {gbase + (int)q, to + q * piece, piece * sizeof(E)}; }
        // every rank of the communicator is in such a group of P at this point of the call sequence (aligned groups): rank q sends one
        // piece to each member of its own group — the pattern the link striping needs (small groups leave most links of the mesh idle)
        const size_t pb = piece * sizeof(E); const int Pg = (int)P;
        return xchg(tr, [pb, Pg](int q, std::vector<Transport::MsgDesc>& m) { const int gb = (q / Pg) * Pg; for (int j = 0; j < Pg; ++j) m.push_back({gb + j, pb}); },
                    snd, (int)P, rcv, (int)P, s, P < (size_t)tr.world);
    }
    // one grouped exchange of a split transform, striped over the links of the mesh when that pays (Transport::exchange_striped);
    // `pat(q)` = what rank q sends at this point of the call sequence.  Only inside api_enter_split / api_exit_split, which provide
    // the relays' staging buffer; everywhere else (and with ECFFT_NO_STRIPE in a test build) the plain exchange.
    bool xchg(Transport& tr, const Transport::PatternFn& pat, const P2P* snd, int ns, const P2P* rcv, int nr, hipStream_t s, bool worth = true) const {
        if (!worth || !stripe_stage_ || stripe_off_) return tr.exchange(snd, ns, rcv, nr, s);
        if (stripe_gain_ != ~(size_t)0) tr.stripe_min_gain = stripe_gain_;
        return tr.exchange_striped(pat, snd, ns, rcv, nr, stripe_stage_, stripe_bytes_, s);
    }
    // FULL contexts: the same compact cyclic tables, gathered once per (tree, P, rank, table) from the full stage tables on first
    // use (entry i'*P + rank of stage k -> offset c - 2*(h_k/P) + i', the layout build_shard_set writes) and kept until the context
    // is destroyed or trimmed — so the split EXTEND of a full context also runs its log P cyclic stages as one fused pass instead
    // of log P one-stage launches with stride-P table reads.  nullptr (allocation failed): the caller runs the one-stage kernels.
    // which: 0 np0, 1 dinv, 2 p0, 3 p1 (stage tables), 4 w, 5 winv (the c cyclic positions i'*P + rank).
    const TE* full_cyclic_table(const Tree& T, unsigned log_p, unsigned r, int sg, int which, hipStream_t s) const {
        const uint64_t key = ((uint64_t)T.log_m << 40) | ((uint64_t)log_p << 32) | ((uint64_t)r << 8) | ((uint64_t)sg << 4) | (uint64_t)which;
        auto it = full_cyc_.find(key);
        if (it != full_cyc_.end()) return it->second;
        const size_t e = T.e, P = (size_t)1 << log_p, c = e >> log_p;
        TE* dst = nullptr;
        if (hipMalloc(&dst, c * sizeof(TE)) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
        const TE* src = which == 0 ? T.np0[sg] : which == 1 ? T.dinv[sg] : which == 2 ? T.p0[sg] : which == 3 ? T.p1[sg] : which == 4 ? T.w[sg] : T.winv[sg];
        if (which >= 4) {
            foreach_n(s, c, [=] __device__(size_t il) { dst[il] = src[il * P + r]; });
        } else {
            (void)hipMemsetAsync(dst, 0, c * sizeof(TE), s);
            for (unsigned k = 0; k < log_p; ++k) {
                const size_t h = e >> (k + 1), hl = h >> log_p, offl = c - 2 * hl, off = e - 2 * h;
                foreach_n(s, hl, [=] __device__(size_t il) { dst[offl + il] = src[off + il * P + r]; });
            }
        }
        full_cyc_[key] = dst; full_cyc_bytes_ += c * sizeof(TE);
        return dst;
    }
    void full_cyclic_free() { for (auto& kv : full_cyc_) (void)hipFree(kv.second); full_cyc_.clear(); full_cyc_bytes_ = 0; }

    // Shard contexts keep the cyclic stages' table entries compact and laid out like the stage tables of a length-c vector, so the
    // log_p cyclic stages are the TOP log_p stages of a "length-c EXTEND" on those tables: ONE fused column pass (k_stages_col)
    // instead of log_p one-stage launches, with the 1/W or W scaling of a cyclic-in / cyclic-out call riding on its load / store.
    // Returns false when the stage count does not fit one column tile (the caller then runs the one-stage kernels).
    bool cyclic_stages_fused(const TE* ta, const TE* tb, const E* src, E* dst, size_t c, unsigned log_p, bool dec, const TE* ld_scale,
                             const TE* st_scale, hipStream_t s) const {
        const unsigned lc = ilog2(c);
        unsigned log_ct = lc < kLogColTileMax ? lc : kLogColTileMax;
        if (small_launch(c) && !ef_small_off_ && log_ct > kLogLowSmall) log_ct = kLogLowSmall;
        const unsigned R = log_p;
        if (R == 0 || R > kColStages || R + 2 > log_ct) return false;
        const unsigned log_cc = log_ct - R;
        IoDesc<F> d = io_plain(src, dst);
        if (ld_scale) { d.ld_mode = LD_SCALE; d.ld_tbl = ld_scale; }
        if (st_scale) { d.st_mode = ST_SCALE; d.st_a = st_scale; }
        const bool ct = (log_ct == kLogColTileMax && (sizeof(E) == 4 || ECFFT_CT_ALL));
        dim3 grid((unsigned)(c >> log_ct)); const size_t lds = sizeof(E) * ((size_t)col_row_stride<E>(1u << log_cc) << R);
        double hsum = 0; for (unsigned k = 0; k < log_p; ++k) hsum += (double)(c >> (k + 1));
        const double bytes = sizeof(E) * (2.0 * R * c + 4.0 * hsum);
        constexpr bool kCtCol = sizeof(E) == 4 || ECFFT_CT_ALL;              // compile-time column tiles: 4-byte fields only (see extend_core)
        if (dec) {
            if constexpr (kCtCol) { if (ct) { ECFFT_LAUNCH(KC_COL, bytes, (k_stages_col<F, true, (int)kLogColTileMax>), grid, dim3(kBlockLds), lds, s, d, ta, tb, lc, 0u, log_p - 1, log_cc, (const TE*)nullptr, 0u); return true; } }
            ECFFT_LAUNCH(KC_COL, bytes, (k_stages_col<F, true, 0>), grid, dim3(kBlockLds), lds, s, d, ta, tb, lc, 0u, log_p - 1, log_cc, (const TE*)nullptr, 0u);
        } else {
            if constexpr (kCtCol) { if (ct) { ECFFT_LAUNCH(KC_COL, bytes, (k_stages_col<F, false, (int)kLogColTileMax>), grid, dim3(kBlockLds), lds, s, d, ta, tb, lc, 0u, log_p - 1, log_cc, (const TE*)nullptr, 0u); return true; } }
            ECFFT_LAUNCH(KC_COL, bytes, (k_stages_col<F, false, 0>), grid, dim3(kBlockLds), lds, s, d, ta, tb, lc, 0u, log_p - 1, log_cc, (const TE*)nullptr, 0u);
        }
        return true;
    }
    // FFTree::extend (src/fftree.rs:123-126) of ONE vector of e evaluations held block-distributed by the P = 2^log_p ranks
    // [gbase, gbase + P) of `tr`: rank gbase + r holds positions [r*c, (r+1)*c), c = e/P.  in / out: this rank's shard (may
    // alias).  A, B: scratch of c elements each.  Stage k pairs (i, i + e >> (k+1)): stages k >= log_p are local in the block
    // distribution, stages k < log_p in the cyclic one (position j on rank j mod P).
    // cyc_in / cyc_out: the shard is CYCLIC instead (local j' <-> global position j'*P + r), which drops the first / last of the
    // four exchanges — for callers that chain split EXTENDs or produce / consume the cyclic order anyway.
    bool extend_split(Transport& tr, int gbase, unsigned log_p, const E* in, E* out, size_t e, int target, hipStream_t s, E* A, E* B,
                      bool cyc_in = false, bool cyc_out = false) const {
        const unsigned log_m = ilog2(e) + 1;
        const Tree& T = tree_at(log_m);
        const size_t P = (size_t)1 << log_p, c = e >> log_p, cp = c >> log_p, g0 = (size_t)(tr.rank - gbase) * c;
        const unsigned r = (unsigned)(tr.rank - gbase);
        const int src = 1 - target;
        if (c < P || c < 2) return false;
        const bool sh = shard_mode();                                          // tables of this context hold only this rank's share
        const ShardSet* sp = (ovr_tree_ && ovr_tree_->log_m == log_m) ? ovr_set_ : (log_m < sets_.size() && sets_[log_m].valid ? &sets_[log_m] : nullptr);
        if (sh && (!sp || sp->log_p != log_p || sp->rank != r)) return false;
        static const ShardSet kNoSet{};
        const ShardSet& SS = sh ? *sp : kNoSet;
        const auto& cyc_ = SS.cyc; const auto& cycw_ = SS.cycw;               // [parity][np0, dinv, p0, p1] and [parity][w, winv], compact
        bool dec_done = false;
        // full context: compact copies of the cyclic entries, gathered on first use (full_cyclic_table)
        const TE *fd0 = nullptr, *fd1 = nullptr, *fr0 = nullptr, *fr1 = nullptr, *fwi = nullptr, *fw = nullptr;
        if (!sh && log_p >= 1 && !full_cyc_off_) {
            fd0 = full_cyclic_table(T, log_p, r, src, 0, s); fd1 = full_cyclic_table(T, log_p, r, src, 1, s);
            fr0 = full_cyclic_table(T, log_p, r, target, 2, s); fr1 = full_cyclic_table(T, log_p, r, target, 3, s);
            if (cyc_in) fwi = full_cyclic_table(T, log_p, r, src, 5, s);
            if (cyc_out) fw = full_cyclic_table(T, log_p, r, target, 4, s);
        }
        const bool fdec = fd0 && fd1 && (!cyc_in || fwi), frec = fr0 && fr1 && (!cyc_out || fw);
        if (cyc_in && fdec && cyclic_stages_fused(fd0, fd1, in, B, c, log_p, true, fwi, nullptr, s)) {
            dec_done = true;
        } else if (cyc_in && sh && cyclic_stages_fused(cyc_[src][0], cyc_[src][1], in, B, c, log_p, true, cycw_[src][1], nullptr, s)) {
            dec_done = true;                                                   // 1/W + every cyclic decompose stage in one pass
        } else if (cyc_in) {   // already cyclic: 1/W_src of positions j'*P + r
            if (sh) ECFFT_LAUNCH(KC_POINTWISE, 0.0, k_scale_by_table<F>, dim3(nblocks(c)), dim3(kBlock), 0, s, B, in, (const TE*)cycw_[src][1], c - 1, c, 1u, 0u);
            else ECFFT_LAUNCH(KC_POINTWISE, 0.0, k_scale_by_table<F>, dim3(nblocks(c)), dim3(kBlock), 0, s, B, in, T.winv[src], c - 1, c, (uint32_t)P, r);
        } else {
            {   // 1/W_src scaling + pack for block -> cyclic: element i goes to rank i mod P, slot i / P
                const TE* wi = T.winv[src]; const unsigned lp = log_p;
                foreach_n(s, c, [=] __device__(size_t i) { A[(i & (P - 1)) * cp + (i >> lp)] = F::canon(F::tmul(wi[g0 + i], in[i])); });
            }
            if (!exchange_group(tr, gbase, P, A, B, cp, s)) return false;      // B = cyclic shard, ascending local index
        }
        const size_t npairs = c / 2;
        if (!dec_done && sh && cyclic_stages_fused(cyc_[src][0], cyc_[src][1], B, B, c, log_p, true, nullptr, nullptr, s)) dec_done = true;
        if (!dec_done && fdec && cyclic_stages_fused(fd0, fd1, B, B, c, log_p, true, nullptr, nullptr, s)) dec_done = true;
        for (unsigned k = 0; k < log_p && !dec_done; ++k) {                    // cyclic shard: top decompose stages, table stride P / offset r
            size_t h = e >> (k + 1), off = e - 2 * h;
            if (sh) {
                const size_t offl = c - 2 * (h >> log_p);
                ECFFT_LAUNCH(KC_DECOMPOSE, sizeof(E) * (2.0 * c + 4.0 * (h >> log_p)), k_decompose_stage<F>, dim3(nblocks(npairs)), dim3(kBlock), 0, s,
                             B, (const TE*)(cyc_[src][0] + offl), (const TE*)(cyc_[src][1] + offl), ilog2(h >> log_p), npairs, 1u, 0u);
                continue;
            }
            ECFFT_LAUNCH(KC_DECOMPOSE, sizeof(E) * (2.0 * c + 4.0 * (h >> log_p)), k_decompose_stage<F>, dim3(nblocks(npairs)), dim3(kBlock), 0, s,
                         B, T.np0[src] + off, T.dinv[src] + off, ilog2(h >> log_p), npairs, (uint32_t)P, r);
        }
        if (!exchange_group(tr, gbase, P, B, A, cp, s)) return false;          // A = the P chunks of the block shard, source major
        {   // block shard: every stage k >= log_p, fused passes; unpack on the first load, pack on the last store
            IoDesc<F> io = io_plain(A, B);
            io.ld_tr_logp = log_p; io.st_tr_logp = log_p; io.tr_chunk = cp;
            extend_core(log_m, io, out, c, src, s, 0.0, 0.0, log_p);
        }
        if (!exchange_group(tr, gbase, P, B, A, cp, s)) return false;          // A = cyclic shard
        bool rec_done = false;
        if (sh && cyclic_stages_fused(cyc_[target][2], cyc_[target][3], A, cyc_out ? out : A, c, log_p, false, nullptr, cyc_out ? cycw_[target][0] : nullptr, s)) {
            if (cyc_out) return hipGetLastError() == hipSuccess;               // every cyclic recombine stage + W in one pass
            rec_done = true;
        }
        if (!rec_done && frec && cyclic_stages_fused(fr0, fr1, A, cyc_out ? out : A, c, log_p, false, nullptr, cyc_out ? fw : nullptr, s)) {
            if (cyc_out) return hipGetLastError() == hipSuccess;
            rec_done = true;
        }
        for (unsigned k = log_p; !rec_done && k-- > 0;) {
            size_t h = e >> (k + 1), off = e - 2 * h;
            if (sh) {
                const size_t offl = c - 2 * (h >> log_p);
                ECFFT_LAUNCH(KC_RECOMBINE, sizeof(E) * (2.0 * c + 4.0 * (h >> log_p)), k_recombine_stage<F>, dim3(nblocks(npairs)), dim3(kBlock), 0, s,
                             A, (const TE*)(cyc_[target][2] + offl), (const TE*)(cyc_[target][3] + offl), ilog2(h >> log_p), npairs, 1u, 0u);
                continue;
            }
            ECFFT_LAUNCH(KC_RECOMBINE, sizeof(E) * (2.0 * c + 4.0 * (h >> log_p)), k_recombine_stage<F>, dim3(nblocks(npairs)), dim3(kBlock), 0, s,
                         A, T.p0[target] + off, T.p1[target] + off, ilog2(h >> log_p), npairs, (uint32_t)P, r);
        }
        if (cyc_out) {  // stay cyclic: W_target of positions j'*P + r
            if (sh) ECFFT_LAUNCH(KC_POINTWISE, 0.0, k_scale_by_table<F>, dim3(nblocks(c)), dim3(kBlock), 0, s, out, (const E*)A, (const TE*)cycw_[target][0], c - 1, c, 1u, 0u);
            else ECFFT_LAUNCH(KC_POINTWISE, 0.0, k_scale_by_table<F>, dim3(nblocks(c)), dim3(kBlock), 0, s, out, (const E*)A, T.w[target], c - 1, c, (uint32_t)P, r);
            return hipGetLastError() == hipSuccess;
        }
        if (!exchange_group(tr, gbase, P, A, B, cp, s)) return false;          // B = chunks of the block shard
        {   // unpack + W_target scaling
            const TE* w = T.w[target]; const unsigned lp = log_p;
            foreach_n(s, c, [=] __device__(size_t i) { out[i] = F::canon(F::tmul(w[g0 + i], B[(i & (P - 1)) * cp + (i >> lp)])); });
        }
        return hipGetLastError() == hipSuccess;
    }
    // Local preparation of a collective call (its temporaries) + agreement across the ranks (Transport::vote) the FIRST time a
    // call shape is seen on this context: a rank whose allocation failed must not leave its peers blocked in ncclRecv.  After
    // an agreed first call the shape's temporaries are pinned in the pool, so later calls of that shape allocate nothing, cannot
    // fail locally and stay asynchronous (no vote).  Every rank makes the same calls with the same sizes (the ABI's contract), so
    // "first time" is the same moment on all of them.
    template <class Alloc>
    bool collective_prepare(Transport& tr, int op, size_t len, int variant, hipStream_t s, Alloc alloc) {
        bool local_ok = true;
        try { alloc(); } catch (const DeviceAllocError&) { local_ok = false; }
        const uint64_t key = ((uint64_t)op << 60) | ((uint64_t)variant << 56) | ((uint64_t)(unsigned)tr.world << 48) | (uint64_t)len;
        // An agreed shape cannot fail locally: its temporaries are pinned in the pool.  It therefore never votes again — a vote is a
        // symmetric all-rank exchange, and a rank voting alone would pair its 4-byte messages with its peers' data messages
        // (ADVICE r04).  Should the impossible happen, the rank reports the error and its peers are released by the caller's
        // ecfft_comm_abort; the failure-injection hook only applies to shapes that still vote (it stays armed until one comes).
        if (agreed_shapes_.count(key) && tr.voted()) {
            if (!local_ok) { fprintf(stderr, "ecfft: a pinned temporary of an agreed sharded call shape could not be taken\n"); temps_done(); }
            return local_ok;
        }
#ifdef ECFFT_TEST_HOOKS
        if (fail_next_collective_) { local_ok = false; fail_next_collective_ = false; }        // test hook
#endif
        if (!tr.vote(local_ok, s)) { temps_done(); return false; }
        agreed_shapes_.insert(key);
        for (auto& b : pool_) if (b.busy) b.pinned = true;
        return true;
    }
#ifdef ECFFT_TEST_HOOKS
    void test_fail_next_collective() { std::lock_guard<std::mutex> g(mu_); fail_next_collective_ = true; }
#endif
    bool api_extend_split(Transport& tr, const E* in, E* out, size_t e, int target, hipStream_t s, bool cyc_in = false, bool cyc_out = false) {
        const size_t P = (size_t)tr.world, c = e / P;
        if (P & (P - 1)) return false;
        E *A = nullptr, *B = nullptr;
        if (!collective_prepare(tr, 1, e, (cyc_in ? 1 : 0) | (cyc_out ? 2 : 0), s, [&] { A = temp(c); B = temp(c); })) return false;
        bool ok = extend_split(tr, 0, ilog2(P), in, out, e, target, s, A, B, cyc_in, cyc_out);
        temps_done();
        return ok;
    }
    // FFTree::enter of n coefficients block-distributed over all ranks of `tr` (rank r holds [r*c, (r+1)*c), c = n/P).
    // Levels m <= c are the rank-local ENTER of the chunk; level m = c*Q (Q = 2, 4, .. P) works inside groups of Q consecutive
    // ranks, and every vector of a level stays CYCLIC across it: u0 (on the lower half-group) and v0 (upper) arrive cyclic over
    // their half-group, the split EXTEND runs cyclic-in / cyclic-out (2 exchanges; none for Q = 2), and ONE exchange hands every
    // rank the operands of the outputs it owns in the NEXT level's cyclic order — position i = i'*Q + a of the level's result is
    // out[i] = X + xnn_s[i] * Y (src/fftree.rs:155-159) with (X, Y) = (u0, v0)[i/2] for even i and (u1, v1)[i/2] for odd i, and
    // all positions of a rank have the parity of a, so rank a needs the whole `cur` (a even) or `ext` (a odd) of sub-rank a/2 of
    // both half-groups.  3 exchanges per level (block form: 5), one more at the end to return to the block order.
    bool api_enter_split(Transport& tr, const E* in, E* out, size_t n, hipStream_t s) {
        const size_t P = (size_t)tr.world, c = n / P, cp = c / P;
        if ((P & (P - 1)) || c < 2 * P) return false;
        const bool sh = shard_mode();
        E *cur = nullptr, *ext = nullptr, *U = nullptr, *V = nullptr, *A = nullptr, *B = nullptr;
        E* stg = nullptr;
        const bool stripe = P >= 4 && !stripe_off_;
        if (!collective_prepare(tr, 2, n, stripe ? 1 : 0, s, [&] { cur = temp(c); ext = temp(c); U = temp(c); V = temp(c); A = temp(c); B = temp(c); if (stripe) stg = temp(2 * c); })) return false;
        StripeScope stripe_scope(this, stg, 2 * c * sizeof(E));                 // relays' staging of the striped exchanges (2c elements bound every pattern below)
        bool ok = enter(in, cur, c, 1, s);
        for (size_t Q = 2; ok && Q <= P; Q *= 2) {
            const size_t half = Q / 2, m = c * Q, e = m / 2;
            const int base = (int)((tr.rank / Q) * Q), a = tr.rank - base, g = a / (int)half, ap = a % (int)half;
            if (half == 1) ok = extend(cur, ext, e, 1, 1, s);
            else ok = extend_split(tr, base + g * (int)half, ilog2(half), cur, ext, e, 1, s, A, B, true, true);
            if (!ok) break;
            P2P snd[2] = {{base + 2 * ap, cur, c * sizeof(E)}, {base + 2 * ap + 1, ext, c * sizeof(E)}};
            P2P rcv[2] = {{base + a / 2, U, c * sizeof(E)}, {base + (int)half + a / 2, V, c * sizeof(E)}};
            { const size_t cb = c * sizeof(E); const int Qi = (int)Q, hi = (int)half;
              ok = xchg(tr, [cb, Qi, hi](int q, std::vector<Transport::MsgDesc>& m) { const int b = (q / Qi) * Qi, p = (q - b) % hi; m.push_back({b + 2 * p, cb}); m.push_back({b + 2 * p + 1, cb}); },
                        snd, 2, rcv, 2, s); }
            // a shard context holds exactly the c entries xnn_s[i'*Q + a] of T_m, compact; a full one the whole table
            const E* xnn = trees_[ilog2(m)].xnn + (sh ? 0 : a); const size_t xs = sh ? 1 : Q;
            foreach_n(s, c, [=] __device__(size_t i) { cur[i] = F::mul_add(xnn[i * xs], V[i], U[i]); });      // :157-158
        }
        if (ok) {   // cyclic over all ranks -> block: local i' = r'*cp + k is global (r'*cp + k)*P + r = block r', offset k*P + r
            ok = exchange_group(tr, 0, P, cur, B, cp, s);
            const size_t lp = ilog2(P);
            foreach_n(s, c, [=] __device__(size_t i) { out[i] = B[(i & (P - 1)) * cp + (i >> lp)]; });
        }
        ok = ok && hipGetLastError() == hipSuccess;
        temps_done();
        return ok;
    }
    // modular_reduce_impl (src/fftree.rs:277-281) = REDC, multiply by c, REDC (redc_impl :232-259 with moiety S0) of a length-m
    // vector spread over the Q = 2^lq ranks [base, base + Q): every length-m/2 vector has hc = m/2Q entries per rank, in BLOCK
    // order (entry j of the rank = position a*hc + j) or, with cyc, in CYCLIC order (position j*Q + a; the split EXTENDs then run
    // cyclic-in / cyclic-out: 2 exchanges each instead of 4).  (e0, e1) = the rank's de-interleaved input, (h0, h1) = its share
    // of the result; a0i[j*sa], a1[j*sa], zi[j*sz] = the rank's entries of 1/a on S0, a on S1 and 1/Z_0 on S1, cc0[j*sc], cc1[j*sc]
    // = its entries of c on S0 / S1.  t0, x0, x1, A, B: scratch of hc elements each.
    bool modred_split(Transport& tr, int base, unsigned lq, size_t m, const E* e0, const E* e1, E* h0, E* h1, const E* a0i, const E* a1, size_t sa,
                      const E* zi, size_t sz, const E* cc0, const E* cc1, size_t sc, E* t0, E* x0, E* x1, E* A, E* B, hipStream_t s, bool cyc = false) {
        const size_t e = m / 2, hc = e >> lq;
        auto redc = [&](const E* y0, const E* y1) -> bool {
            foreach_n(s, hc, [=] __device__(size_t j) { t0[j] = F::mul(a0i[j * sa], y0[j]); });                                         // :238
            if (!extend_split(tr, base, lq, t0, t0, e, 1, s, A, B, cyc, cyc)) return false;                                            // g1 (:239-245)
            foreach_n(s, hc, [=] __device__(size_t j) { h1[j] = F::mul(zi[j * sz], F::sub(y1[j], F::mul(a1[j * sa], t0[j]))); });      // :253-255
            return extend_split(tr, base, lq, h1, h0, e, 0, s, A, B, cyc, cyc);                                                        // :256
        };
        if (!redc(e0, e1)) return false;
        foreach_n(s, hc, [=] __device__(size_t j) { x0[j] = F::mul(cc0[j * sc], h0[j]); x1[j] = F::mul(cc1[j * sc], h1[j]); });
        return redc(x0, x1);
    }
    // FFTree::exit of n evaluations block-distributed over all ranks.  Level m = c*Q runs inside groups of Q ranks with every
    // length-m/2 vector spread over the whole group, c/2 entries per rank; REDC and the pointwise steps are
    // src/fftree.rs:206-219, 232-259, 277-281 restricted to the rank's positions.
    //   Every vector of a level stays CYCLIC (position j*Q + a on rank a): one all-to-all turns the user's block into (e0, e1)
    //   cyclic over all ranks, each level is 4 cyclic split EXTENDs (8 exchanges) and ONE exchange that re-distributes (u0 | v0)
    //   for the two half-groups of the next level — rank a's whole u0 share is exactly the even (a even) or odd (a odd) half of
    //   what sub-rank a/2 of the lower half-group needs next, its v0 share the same for the upper half-group: 9 exchanges per
    //   level (block form: 17).
    bool api_exit_split(Transport& tr, const E* in, E* out, size_t n, hipStream_t s) {
        const size_t P = (size_t)tr.world, c = n / P, hc = c / 2;
        if ((P & (P - 1)) || c < 2 * P || hc < P) return false;
        E *cur = nullptr, *e0 = nullptr, *e1 = nullptr, *t0 = nullptr, *h0 = nullptr, *h1 = nullptr, *A = nullptr, *B = nullptr, *x0 = nullptr, *x1 = nullptr, *Rb = nullptr;
        E *blk = nullptr, *Y = nullptr;
        const bool sh = shard_mode();
        // round 4, FULL contexts and transforms of at most 2^gather_max_log_: EVERY top level redundantly.  The projection
        // (tools/split_project.py) shows the split top levels latency bound at these sizes — a split level of an n = 2^20 EXIT costs a
        // rank ~0.45 ms in ~30 small launches plus nine exchanges, the same level on the WHOLE block ~0.2 ms at full-chip efficiency.
        // So: ONE all-gather of the input, then every rank walks its own path down the tree — level Q on the Q c block that contains
        // its chunk (single-GPU fused passes on the full tables), keep the half that contains the chunk, level Q / 2 on that, ...
        // — less than twice the top level's arithmetic in total, no exchange after the first.  Needs the whole chain on every GPU
        // (a full context; EXIT-shard contexts keep the split levels — their point is a transform whose tables exceed one GPU, and at
        // those sizes a split level is throughput bound and wins).
        if (!sh && ilog2(n) <= gather_max_log_ && trees_.size() > ilog2(n)) {
            E *ga = nullptr, *gb = nullptr;
            if (!collective_prepare(tr, 3, n, 2, s, [&] { ga = temp(n); gb = temp(n); if (!ensure_scratch(n)) throw DeviceAllocError(); })) return false;
            P2P snd[64], rcv[64];
            if (P > 64) { temps_done(); return false; }
            for (size_t q = 0; q < P; ++q) { snd[q] = {(int)q, const_cast<E*>(in), c * sizeof(E)}; rcv[q] = {(int)q, ga + q * c, c * sizeof(E)}; }
            bool ok = tr.exchange(snd, (int)P, rcv, (int)P, s);
            E *src = ga, *dst = gb; size_t off = 0;                         // the current block starts at src + off
            for (size_t Q = P; ok && Q >= 2; Q /= 2) {
                const size_t m = c * Q;
                const size_t blk0 = ((size_t)tr.rank / Q) * Q * c;           // global position of the block of Q c that contains this rank's chunk
                if (Q == P) off = blk0;                                      // (0: the whole vector)
                exit_levels(src + off, dst, m, 1, s, scratch_, ilog2(m), ilog2(m));       // dst[0, m) = [u0 | v0] of the block
                off = (((size_t)tr.rank / (Q / 2)) & 1) * (m / 2);            // the half that contains the chunk
                E* t = src; src = dst; dst = t;
            }
            if (ok) ok = exit(src + off, out, c, 1, s);
            ok = ok && hipGetLastError() == hipSuccess;
            temps_done();
            return ok;
        }
        // round 4: the level of the PAIRS (Q = 2, blocks of 2c) runs redundantly on both ranks of a pair: ONE exchange hands each rank
        // its partner's (e0, e1) share, the level itself is the single-GPU EXIT level of the 2c block on the full tree T_2c (fused
        // passes, no pack / unpack, no cyclic passes), and each rank keeps its own half of [u0 | v0] — which IS its chunk for the
        // local levels: 1 exchange instead of 9 (8 of the split EXTENDs + the re-blocking one) for twice the level's arithmetic.
        // EXIT-shard contexts carry T_2c for it (build_exit_shard); ECFFT_SPLIT_Q2_SPLIT=1 keeps the split form on a FULL context (A/B).
        const bool q2_local = sh ? have_pair_full_ : !q2_split_;      // shard contexts: the form the ranks agreed on at build time
        E* stg = nullptr;
        const bool stripe = P >= 4 && !stripe_off_;
        if (!collective_prepare(tr, 3, n, (q2_local ? 1 : 0) | (stripe ? 4 : 0), s, [&] { if (stripe) stg = temp(c); cur = temp(c); e0 = temp(hc); e1 = temp(hc); t0 = temp(hc); h0 = temp(hc); h1 = temp(hc); A = temp(hc); B = temp(hc);
                                                      x0 = temp(hc); x1 = temp(hc); Rb = temp(c);
                                                      if (q2_local) { blk = temp(2 * c); Y = temp(2 * c); if (!ensure_scratch(2 * c)) throw DeviceAllocError(); } })) return false;
        StripeScope stripe_scope(this, stg, c * sizeof(E));
        bool ok = true;
        {   // block -> (e0, e1) cyclic over all ranks: pair t = t'*P + r' of the chunk goes to rank r', slot t'
            const size_t cpp = hc / P; const unsigned lp = ilog2(P);
            E* S = cur;                                                      // [target r'][e0 piece | e1 piece]
            foreach_n(s, hc, [=] __device__(size_t t) {
                const size_t rp = t & (P - 1), tp = t >> lp;
                S[rp * 2 * cpp + tp] = in[2 * t]; S[rp * 2 * cpp + cpp + tp] = in[2 * t + 1];
            });
            ok = exchange_group(tr, 0, P, S, Rb, 2 * cpp, s);
            foreach_n(s, hc, [=] __device__(size_t j) { const size_t r = j / cpp, tp = j - r * cpp; e0[j] = Rb[r * 2 * cpp + tp]; e1[j] = Rb[r * 2 * cpp + cpp + tp]; });
        }
        for (size_t Q = P; ok && Q >= 2; Q /= 2) {
            const size_t half = Q / 2, m = c * Q;
            const int base = (int)((tr.rank / Q) * Q), a = tr.rank - base, ap = a % (int)half;
            const Tree& T = trees_[ilog2(m)];
            if (Q == 2 && q2_local) {
                // (e0, e1)[j] = evaluations 2i, 2i + 1 of the 2c block for i = 2j + a; the partner holds i = 2j + (1 - a)
                E *pe0 = h0, *pe1 = h1;
                P2P snd[2] = {{base + 1 - a, e0, hc * sizeof(E)}, {base + 1 - a, e1, hc * sizeof(E)}};
                P2P rcv[2] = {{base + 1 - a, pe0, hc * sizeof(E)}, {base + 1 - a, pe1, hc * sizeof(E)}};
                { const size_t hb = hc * sizeof(E);
                  ok = xchg(tr, [hb](int q, std::vector<Transport::MsgDesc>& m) { m.push_back({q ^ 1, hb}); m.push_back({q ^ 1, hb}); }, snd, 2, rcv, 2, s); }
                if (!ok) break;
                { const size_t aa = (size_t)a, bb = (size_t)(1 - a);
                  foreach_n(s, hc, [=] __device__(size_t j) {
                      blk[2 * (2 * j + aa)] = e0[j]; blk[2 * (2 * j + aa) + 1] = e1[j];
                      blk[2 * (2 * j + bb)] = pe0[j]; blk[2 * (2 * j + bb) + 1] = pe1[j];
                  }); }
                const unsigned lm = ilog2(m);
                if (sh) { if (!have_pair_full_) { ok = false; break; } ovr_tree_ = &pair_full_; ovr_set_ = nullptr; }     // the full T_2c an EXIT-shard context keeps aside
                exit_levels(blk, Y, m, 1, s, scratch_, lm, lm);            // src/fftree.rs:200-224 for the block: Y = [u0 | v0]
                ovr_tree_ = nullptr;
                ok = exit(Y + (size_t)a * c, out, c, 1, s);                   // this rank's half is its chunk of the local levels
                ok = ok && hipGetLastError() == hipSuccess;
                temps_done();
                return ok;
            }
            // the rank's entries of 1/xnn_s on S0, xnn_s on S1, 1/z0_s1 and z0z0_rem_xnn_s on S0 / S1 (positions j*Q + a of each
            // half): strided views of the full tables, or the compact arrays an EXIT-shard context holds in the same fields
            const size_t s2 = sh ? 1 : 2 * Q, s1 = sh ? 1 : Q;
            const E *xi = sh ? T.xnn_inv : T.xnn_inv + 2 * a, *xo = sh ? T.xnn : T.xnn + 2 * a + 1, *zi = sh ? T.z0_inv_s1 : T.z0_inv_s1 + a;
            const E *cc0 = sh ? T.z0z0 : T.z0z0 + 2 * a, *cc1 = sh ? T.z1z1 : T.z0z0 + 2 * a + 1;
            ok = modred_split(tr, base, ilog2(Q), m, e0, e1, h0, h1, xi, xo, s2, zi, s1, cc0, cc1, s2, t0, x0, x1, A, B, s, true);
            if (!ok) break;
            foreach_n(s, hc, [=] __device__(size_t j) { h1[j] = F::mul(xi[j * s2], F::sub(e0[j], h0[j])); });               // :215-219
            P2P snd[2] = {{base + a / 2, h0, hc * sizeof(E)}, {base + (int)half + a / 2, h1, hc * sizeof(E)}};
            P2P rcv[2] = {{base + 2 * ap, e0, hc * sizeof(E)}, {base + 2 * ap + 1, e1, hc * sizeof(E)}};
            { const size_t hb = hc * sizeof(E); const int Qi = (int)Q, hi = (int)half;
              ok = xchg(tr, [hb, Qi, hi](int q, std::vector<Transport::MsgDesc>& m) { const int b = (q / Qi) * Qi, aq = q - b; m.push_back({b + aq / 2, hb}); m.push_back({b + hi + aq / 2, hb}); },
                        snd, 2, rcv, 2, s); }
        }
        if (ok) foreach_n(s, hc, [=] __device__(size_t j) { cur[2 * j] = e0[j]; cur[2 * j + 1] = e1[j]; });
        if (ok) ok = exit(cur, out, c, 1, s);
        ok = ok && hipGetLastError() == hipSuccess;
        temps_done();
        return ok;
    }

    // FFTree::enter (src/fftree.rs:164-167): n coefficients -> n evaluations on the leaves of T_n.
    // in/out: device pointers, n elements, may alias.  Uses ctx scratch (caller holds lock()).
    // `count` independent polynomials of length n laid end to end share every launch (batched form; count = 1 is the
    // reference call): level l treats the buffer as count*n/m blocks.
    bool enter(const E* in, E* out, size_t n, size_t count, hipStream_t s) {
        const size_t nt = n * count;
        if (n == 1) { if (in != out) (void)hipMemcpyAsync(out, in, nt * sizeof(E), hipMemcpyDeviceToDevice, s); return true; }
        if (!ensure_scratch(nt)) return false;
        unsigned ln = ilog2(n);
        if (count == 1 && ln >= kSplitMinLog && nside_ > 0) {
            int next_side = 0;
            in_halves_ = true;
            enter_rec(in, out, n, s, scratch_, kSplitDepth, next_side);
            in_halves_ = false;
        } else if (const int ways = batch_ways(nt, count); ways > 1) {
            // round 6: a batch runs as `ways` parts of whole polynomials on as many streams — nothing to join but the end: one part's
            // load / store phases and launch fill / drain meet the others' multiplies, as in the two-halves schedule of one transform
            const size_t part = nt / (size_t)ways, cnt = count / (size_t)ways;
            run_ways(ways, s, [&](int i, hipStream_t si) { enter_levels(in + part * i, out + part * i, n, cnt, si, scratch_ + 3 * part * i, 1, ln); });
        } else {
            enter_levels(in, out, n, count, s, scratch_, 1, ln);
        }
        return true;
    }
    // fork `ways - 1` side streams off `s`, run part i on stream i (part 0 on `s` itself), join them back into `s`
    template <class Fn>
    void run_ways(int ways, hipStream_t s, Fn fn) {
        for (int i = 1; i < ways; ++i) { (void)hipEventRecord(ev_fork_[i - 1], s); (void)hipStreamWaitEvent(sides_[i - 1], ev_fork_[i - 1], 0); }
        in_halves_ = true;
        const double w = tblw_;
        for (int i = 0; i < ways; ++i) { if (i) tblw_ = 0.0; fn(i, i ? sides_[i - 1] : s); }     // (the tables are credited once in the byte model)
        tblw_ = w;
        in_halves_ = false;
        for (int i = 1; i < ways; ++i) { (void)hipEventRecord(ev_join_[i - 1], sides_[i - 1]); (void)hipStreamWaitEvent(s, ev_join_[i - 1], 0); }
    }
    // batched ENTER / EXIT / EXTEND as concurrent parts: the number of parts (1 = one stream) — a power of two <= ECFFT_BATCH_WAYS that
    // divides the count, every part at least 2^kSplitMinLog elements
    int batch_ways(size_t nt, size_t count) const {
        if (!ECFFT_BATCH_SPLIT || count < 2) return 1;
        for (int w = ECFFT_BATCH_WAYS; w >= 2; w >>= 1)
            if (count % (size_t)w == 0 && w - 1 <= nside_ && ((nt / (size_t)w) >> kSplitMinLog) != 0) return w;
        return 1;
    }
    // Concurrent halves, recursively: levels 1..L-1 never mix the two half-blocks, so they run as two independent ENTERs
    // of n/2 on two streams.  Their launches (each half as wide) interleave on the chip, so one half's load / store phases
    // overlap the other's compute; only the top level runs on the whole array.  Scratch: f(n) = n + 2 f(n/2), f = 3n at a leaf.
    void enter_rec(const E* in, E* out, size_t n, hipStream_t s, E* base, unsigned depth, int& next_side) {
        unsigned ln = ilog2(n);
        if (depth == 0 || ln < kSplitMinLog || next_side >= nside_) { enter_levels(in, out, n, 1, s, base, 1, ln); return; }
        int me = next_side++;
        hipStream_t s2 = sides_[me];
        E* X = base; E* sA = base + n; E* sB = sA + scratch_need(n / 2, depth - 1);
        (void)hipEventRecord(ev_fork_[me], s); (void)hipStreamWaitEvent(s2, ev_fork_[me], 0);
        enter_rec(in, X, n / 2, s, sA, depth - 1, next_side);
        double w = tblw_; tblw_ = 0.0; enter_rec(in + n / 2, X + n / 2, n / 2, s2, sB, depth - 1, next_side); tblw_ = w;
        (void)hipEventRecord(ev_join_[me], s2); (void)hipStreamWaitEvent(s, ev_join_[me], 0);
        enter_levels(X, out, n, 1, s, sA, ln, ln);
    }
    static size_t scratch_need(size_t n, unsigned depth) { return depth == 0 ? 3 * n : n + 2 * scratch_need(n / 2, depth - 1) > 4 * n ? n + 2 * scratch_need(n / 2, depth - 1) : 4 * n; }
    // levels l_begin..l_end of ENTER on count arrays of n elements; `in` = state before level l_begin; base = 3*n*count
    // elements of scratch (ping, pong, EXTEND work)
    void enter_levels(const E* in, E* out, size_t n, size_t count, hipStream_t s, E* base, unsigned l_begin, unsigned l_end) const {
        const size_t nt = n * count;
        E* bufA = base; E* bufB = base + nt; E* work = base + 2 * nt;
        const E* src = in;
        unsigned l0 = l_begin;
        const unsigned ll = log_low_for(nt);
        if (l_begin == 1 && l_end >= ll) {
            // levels 1..ll: one launch, one HBM round trip (k_enter_low)
            E* dst = (l_end == ll && out != in) ? out : bufA;
            double bytes = 0; for (unsigned l = 1; l <= ll; ++l) bytes += enter_level_alg_bytes(nt, l);
            if (ll == kLogLow)
                ECFFT_LAUNCH(KC_FUSED_ENTER, bytes, (k_enter_low<F, (int)kLogLow>), dim3((unsigned)(nt >> kLogLow)), dim3(kBlockLds),
                             2 * (sizeof(E) << kLogLow), s, dst, src, (const Tree*)d_trees_);
            else
                ECFFT_LAUNCH(KC_FUSED_ENTER, bytes, (k_enter_low<F, (int)kLogLowSmall, 2 * (int)kBlockLowSmall>), dim3((unsigned)(nt >> kLogLowSmall)), dim3(2 * kBlockLowSmall),
                             2 * (sizeof(E) << kLogLowSmall), s, dst, src, (const Tree*)d_trees_);
            src = dst; l0 = ll + 1;
        }
        for (unsigned l = l0; l <= l_end; ++l) {
            const Tree& T = trees_[l];
            size_t e = T.e;
            E* dst = (l == l_end && out != in) ? out : (src == bufA ? bufB : bufA);
            // EXTEND of every [u0 | v0] half onto S1 with the combine (:155-159) folded into its last pass
            IoDesc<F> io = io_plain(src, work); io.ld_mode = LD_SCALE; io.ld_tbl = T.winv[0];
            EnterFuse ef{src, dst, sizeof(E) * (3.0 * nt + 2.0 * e * tblw_)};
            extend_core(l, io, work, nt, 0, s, 0.0, 0.0, 0, nullptr, false, &ef);
            src = dst;
        }
        if (src != out) (void)hipMemcpyAsync(out, src, nt * sizeof(E), hipMemcpyDeviceToDevice, s);
    }
    // algorithmic bytes of one ENTER / EXIT level in the stage-streaming model
    // (tblw_ = 0 while the second of two concurrent halves is enqueued: each table is credited once)
    double enter_level_alg_bytes(size_t n, unsigned l) const {
        double e = (double)((size_t)1 << (l - 1));
        return sizeof(E) * (4.0 * (l - 1) * (double)n + 3.0 * (double)n + tblw_ * (8.0 * (e - 1) + 2.0 * e));
    }
    double exit_level_alg_bytes(size_t n, unsigned l) const {
        double e = (double)((size_t)1 << (l - 1));
        return sizeof(E) * (8.0 * (l - 1) * (double)n + 8.5 * (double)n + tblw_ * (32.0 * (e - 1) + 8.5 * e));
    }
#ifndef ECFFT_SPLIT_MIN_LOG
#define ECFFT_SPLIT_MIN_LOG 19
#endif
    static constexpr unsigned kSplitMinLog = ECFFT_SPLIT_MIN_LOG;   // single transforms of at least 2^this run as concurrent halves
#ifndef ECFFT_SPLIT_DEPTH
#define ECFFT_SPLIT_DEPTH 1
#endif
#ifndef ECFFT_NT_STORE_MAX_LOG
#define ECFFT_NT_STORE_MAX_LOG 0
#endif
    // launches of <= 2^this elements store their results non-temporally (IoDesc::nt_st, kernels.h data_st); 0 = never, the shipped value:
    // the non-temporal store itself measures neutral (<= 2^19) to negative (above) — profiles/r06/nt_data_ab.txt
    static constexpr unsigned kNtStoreMaxLog = ECFFT_NT_STORE_MAX_LOG;
    static constexpr unsigned kSplitDepth = ECFFT_SPLIT_DEPTH;      // recursion depth of the halving (2^depth concurrent streams)
    static constexpr int kMaxSides = 7;
    static constexpr unsigned kLogLow = (sizeof(E) == 32) ? 10 : 13;     // tile of the fused low-level kernels (2 x 32 KiB of LDS)
    // SMALL launches (fewer tiles than CUs) are latency bound: tools/ubench/sweep.hip shows that ONE 512-thread workgroup
    // already saturates its CU's integer VALUs (a sweep of 512 pairs = 4 multiplies per SIMD, ~3900 cycles), so the only way
    // to shorten a sweep is to spread a tile's pairs over more CUs.  For 32-byte fields launches with < kSmallTiles tiles of
    // the default size use 4x smaller tiles: 256 elements, 128-thread low-level kernels, one wave per SIMD.
    static constexpr unsigned kLogLowSmall = 8, kBlockLowSmall = 128, kSmallTiles = 256;
    // Round 4: the fused PASSES (row / column kernels) of a launch on ONE stream switch to the small tiles below 2 x kSmallTiles tiles
    // (a 2^18 launch is 256 big tiles = half the workgroup slots: ENTER at 2^18 0.77 -> 0.69 ms); inside the two-halves schedule the
    // other stream fills the chip and the rule stays at kSmallTiles (2^20 with 512: +14 %).  The low-level kernels keep kSmallTiles.
    // ECFFT_SMALL_TILES_MAX / ECFFT_SMALL_LOW_MAX: A/B knobs for both.
    unsigned small_tiles_max() const {
        static const int v = ab_env("ECFFT_SMALL_TILES_MAX") ? atoi(ab_env("ECFFT_SMALL_TILES_MAX")) : -1;
        return v >= 0 ? (unsigned)v : (in_halves_ ? kSmallTiles : 2 * kSmallTiles);
    }
    bool small_launch(size_t total) const { return sizeof(E) == 32 && (total >> kLogLow) < small_tiles_max() && total >= ((size_t)1 << kLogLowSmall); }
    static unsigned small_low_max() { static const unsigned v = ab_env("ECFFT_SMALL_LOW_MAX") ? (unsigned)atoi(ab_env("ECFFT_SMALL_LOW_MAX")) : kSmallTiles; return v; }
    unsigned log_low_for(size_t total) const { return sizeof(E) == 32 && (total >> kLogLow) < small_low_max() && total >= ((size_t)1 << kLogLowSmall) && !ef_small_off_ ? kLogLowSmall : kLogLow; }

    // FFTree::exit (src/fftree.rs:227-230): n evaluations -> n coefficients.
    bool exit(const E* in, E* out, size_t n1, size_t count, hipStream_t s) {
        const size_t n = n1 * count;
        if (n1 == 1) { if (in != out) (void)hipMemcpyAsync(out, in, n * sizeof(E), hipMemcpyDeviceToDevice, s); return true; }
        if (!ensure_scratch(n)) return false;
        unsigned ln = ilog2(n1);
        if (count == 1 && ln >= kSplitMinLog && nside_ > 0) {
            int next_side = 0;
            in_halves_ = true;
            exit_rec(in, out, n, s, scratch_, kSplitDepth, next_side);
            in_halves_ = false;
        } else if (const int ways = batch_ways(n, count); ways > 1) {
            const size_t part = n / (size_t)ways, cnt = count / (size_t)ways;
            run_ways(ways, s, [&](int i, hipStream_t si) { exit_levels(in + part * i, out + part * i, n1, cnt, si, scratch_ + 3 * part * i, ln, 1); });
        } else {
            exit_levels(in, out, n1, count, s, scratch_, ln, 1);
        }
        return true;
    }
    // top level on the whole array, then its two output blocks [u0 | v0] are independent EXITs of n/2: two streams, recursively
    void exit_rec(const E* in, E* out, size_t n, hipStream_t s, E* base, unsigned depth, int& next_side) {
        unsigned ln = ilog2(n);
        if (depth == 0 || ln < kSplitMinLog || next_side >= nside_) { exit_levels(in, out, n, 1, s, base, ln, 1); return; }
        int me = next_side++;
        hipStream_t s2 = sides_[me];
        E* Y = base; E* sA = base + n; E* sB = sA + scratch_need(n / 2, depth - 1);
        exit_levels(in, Y, n, 1, s, sA, ln, ln);
        (void)hipEventRecord(ev_fork_[me], s); (void)hipStreamWaitEvent(s2, ev_fork_[me], 0);
        exit_rec(Y, out, n / 2, s, sA, depth - 1, next_side);
        double w = tblw_; tblw_ = 0.0; exit_rec(Y + n / 2, out + n / 2, n / 2, s2, sB, depth - 1, next_side); tblw_ = w;
        (void)hipEventRecord(ev_join_[me], s2); (void)hipStreamWaitEvent(s, ev_join_[me], 0);
    }
    // levels l_from down to l_to of EXIT on count arrays of n1 evaluations; base = 3*n1*count elements of scratch
    void exit_levels(const E* in, E* out, size_t n1, size_t count, hipStream_t s, E* base, unsigned l_from, unsigned l_to) const {
        const size_t n = n1 * count;      // all sizes below are totals over the batch; the level count comes from n1
        E* bufA = base; E* bufB = base + n; E* G = base + 2 * n; E* H = G + n / 2;
        const E* cur = in;
        size_t nh = n / 2;
        const unsigned ll = log_low_for(n);
        unsigned l_stop = (l_to == 1 && l_from >= ll) ? ll : l_to - 1;      // levels l_stop..1 run fused in k_exit_low
        for (unsigned l = l_from; l > l_stop; --l) {
            const Tree& T = tree_at(l);
            E* dst = (l == l_to && out != in) ? out : (cur == bufA ? bufB : bufA);
            // The reference's pointwise steps of this level (8.5 n + 8.5 e algorithmic element moves)
            // are all folded into the first load / last store of the four EXTEND cores:
            //   core 1  load  t0 = e0 * (xinv_even / W0)                    [t0 = e0/a0        : n + e   ]
            //           store h1~ = e1 * (zinv/W1) - g1~ * (x_odd zinv)      [h1                : 1.5n + 2e], kept in H
            //   core 2  h1~ -> h0~ (S1 -> S0)
            //   core 3  load  t0' = h0~ * (c_even xinv_even)                 [h*c and t0'       : 3n + 3e ]
            //           store h1'~ = H * (c_odd zinv) - g1'~ * (x_odd zinv)  [h1'               : 1.5n + 2e]
            //   core 4  store u0 = W0 q0~ ; v0 = (e0 - u0) * xinv_even       [exit split        : 1.5n + 0.5e]
            double se = sizeof(E), ee = (double)T.e * tblw_;
            IoDesc<F> io1 = io_plain(cur, G);
            io1.src_stride = 2; io1.src_off = 0; io1.ld_mode = LD_SCALE; io1.ld_tbl = T.A1;
            io1.st_mode = ST_AXPBY; io1.st_a = T.NB2; io1.st_b = T.B1; io1.aux = cur; io1.aux_stride = 2; io1.aux_off = 1; io1.aux_out = H;
            // consecutive cores meet at a column pass on the same tiles: run those two passes as one launch (k_stages_col_mid)
            NextLoad nl2{LD_PLAIN, nullptr, 0.0}, nl3{LD_SCALE, T.C1, se * (3.0 * n + 3.0 * ee)}, nl4{LD_PLAIN, nullptr, 0.0};
            bool f1 = extend_core(l, io1, G, nh, 0, s, se * (1.0 * n + ee), se * (1.5 * n + 2.0 * ee), 0, &nl2, false);
            bool f2 = extend_core(l, io_plain(G, G), G, nh, 1, s, 0.0, 0.0, 0, &nl3, f1);
            IoDesc<F> io3 = io_plain(G, G);
            io3.ld_mode = LD_SCALE; io3.ld_tbl = T.C1;
            io3.st_mode = ST_AXPBY; io3.st_a = T.NB2; io3.st_b = T.D1; io3.aux = H; io3.aux_stride = 1; io3.aux_off = 0; io3.aux_out = nullptr;
            bool f3 = extend_core(l, io3, G, nh, 0, s, f2 ? 0.0 : se * (3.0 * n + 3.0 * ee), se * (1.5 * n + 2.0 * ee), 0, &nl4, f2);
            IoDesc<F> io4 = io_plain(G, dst);
            io4.st_mode = ST_EXIT_SPLIT; io4.st_a = T.w[0]; io4.st_b = T.xie; io4.aux = cur; io4.aux_stride = 2; io4.aux_off = 0;
            extend_core(l, io4, G, nh, 1, s, 0.0, se * (1.5 * n + 0.5 * ee), 0, nullptr, f3);
            cur = dst;
        }
        if (l_to == 1 && l_from >= ll) {
            E* dst = out != in ? out : (cur == bufA ? bufB : bufA);
            double bytes = 0; for (unsigned l = 1; l <= ll; ++l) bytes += exit_level_alg_bytes(n, l);
            if (ll == kLogLow)
                ECFFT_LAUNCH(KC_FUSED_EXIT, bytes, (k_exit_low<F, (int)kLogLow>), dim3((unsigned)(n >> kLogLow)), dim3(kBlockLds),
                             2 * (sizeof(E) << kLogLow), s, dst, cur, (const Tree*)d_trees_);
            else
                ECFFT_LAUNCH(KC_FUSED_EXIT, bytes, (k_exit_low<F, (int)kLogLowSmall, (int)kBlockLowSmall>), dim3((unsigned)(n >> kLogLowSmall)), dim3(kBlockLowSmall),
                             2 * (sizeof(E) << kLogLowSmall), s, dst, cur, (const Tree*)d_trees_);
            cur = dst;
        }
        if (cur != out) (void)hipMemcpyAsync(out, cur, n * sizeof(E), hipMemcpyDeviceToDevice, s);
    }

    // ------------------------------------------------------------------------------------------
    // Public wrappers of the remaining FFTree algorithms on USER data (crate representation),
    // composed from the same EXTEND kernels.  Device pointers; synchronous (they drain `s` before returning
    // because they use temporaries).  Caller holds lock().
    // ------------------------------------------------------------------------------------------
    bool api_mextend(const E* in, E* out, size_t e, size_t count, int target, hipStream_t s) {     // src/fftree.rs:138-141
        b_mextend(ilog2(e) + 1, in, out, count, target, s, true);
        return finish_api(s);
    }
    // redc_z0 / redc_z1 (src/fftree.rs:264-275): evals, a: n entries
    bool api_redc(const E* evals, const E* a, E* out, size_t n, int moiety, hipStream_t s) {
        unsigned l = ilog2(n); size_t e = n / 2;
        E* a0i = temp(e); E* a1 = temp(e);
        plain_halves(a, a0i, a1, e, s);
        batch_inv(a0i, a0i, e, s);                                          // the reference inverts a0 on every call (:235)
        b_redc(l, evals, a0i, a1, out, moiety, s);
        return finish_api(s);
    }
    // modular_reduce (src/fftree.rs:286-289)
    bool api_modular_reduce(const E* evals, const E* a, const E* c, E* out, size_t n, hipStream_t s) {
        unsigned l = ilog2(n); size_t e = n / 2;
        E* a0i = temp(e); E* a1 = temp(e); E* cp = temp(n);
        plain_halves(a, a0i, a1, e, s);
        batch_inv(a0i, a0i, e, s);
        { const E rinv = rinv_; foreach_n(s, n, [=] __device__(size_t i) { cp[i] = F::mul(c[i], rinv); }); }
        b_modular_reduce(l, evals, a0i, a1, cp, out, s);
        return finish_api(s);
    }
    // vanish (src/fftree.rs:313-316): nd domain points -> 2*nd evaluations on the leaves of T_{2 nd}
    bool api_vanish(const E* dom, E* out, size_t nd, hipStream_t s) {
        b_vanish(ilog2(nd) + 1, dom, out, s, true);
        return finish_api(s);
    }
    // degree (src/fftree.rs:169-198).  The reference's recursion is data dependent (g1 == e1 ? recurse on e0 : recurse on t0);
    // here every level computes BOTH candidates and a device-side flag selects between them, so the whole descent is
    // enqueued without a single host round trip (was: one stream synchronise per level) and the degree is read back once.
    bool api_degree(const E* evals, size_t n, hipStream_t s, size_t* degree) {
        unsigned long long hdeg = 0;
        if (n > 1) {
            E* cur = temp(n); E* e0 = temp(n / 2); E* e1 = temp(n / 2); E* g1 = temp(n / 2);
            unsigned long long* acc = reinterpret_cast<unsigned long long*>(temp((2 * sizeof(unsigned long long) + sizeof(E) - 1) / sizeof(E)));   // {degree, flag}
            (void)hipMemsetAsync(acc, 0, 2 * sizeof(unsigned long long), s);
            (void)hipMemcpyAsync(cur, evals, n * sizeof(E), hipMemcpyDeviceToDevice, s);
            for (size_t m = n; m >= 2; m >>= 1) {
                unsigned l = ilog2(m); size_t e = m / 2;
                const Tree& T = trees_[l];
                foreach_n(s, e, [=] __device__(size_t i) { e0[i] = cur[2 * i]; e1[i] = cur[2 * i + 1]; });
                b_extend(l, e0, g1, 1, 1, s);                               // :180
                const E* zi = T.z0_inv_s1;
                foreach_n(s, e, [=] __device__(size_t i) {                  // :181 (flag) and :187-189 (t1, kept in e1)
                    E d = F::sub(e1[i], g1[i]);
                    if (!F::is_zero(d)) atomicOr(acc + 1, 1ull);
                    e1[i] = F::mul(d, zi[i]);
                });
                b_extend(l, e1, g1, 1, 0, s);                               // t0 (:190)
                foreach_n(s, e, [=] __device__(size_t i) { cur[i] = acc[1] ? g1[i] : e0[i]; });   // :182 / :191
                foreach_n(s, 1, [=] __device__(size_t) { if (acc[1]) acc[0] += (unsigned long long)e; acc[1] = 0; });
            }
            if (hipMemcpyAsync(&hdeg, acc, sizeof(hdeg), hipMemcpyDeviceToHost, s) != hipSuccess) { temps_done(); return false; }
        }
        bool ok = finish_api(s);
        *degree = (size_t)hdeg;
        return ok;
    }

    // The reference's own (un-normalised) matrices of T_m, rebuilt on demand for export (src/fftree.rs:341-363):
    // R = [[v0, s0 v0], [v1, s1 v1]], v_j = v(s_j)^(d/2-1), D = R^-1; identity where d == 1.  out: 4*m elements on the
    // device, row-major Mat2x2 in BinaryTree heap order, crate representation.  Synchronous.
    // standard = true: plain (standard-form) residues instead of the crate's in-memory form — what ark-serialize writes
    bool export_matrices(unsigned log_m, bool decompose, E* out, hipStream_t s, bool standard = false) {
        size_t m = (size_t)1 << log_m, N = N_, stride = N_ / m;
        const E* f = f_; const E* den = den_;
        foreach_n(s, m, [=] __device__(size_t idx) {
            E one = standard ? F::one() : F::to_mont(F::one()), zero = F::zero();
            E r00 = one, r01 = zero, r10 = zero, r11 = one;
            size_t d = 1; unsigned k = 0;
            if (idx >= 2) {                                   // layer k occupies [d, 2d), d = m >> (k+1)
                unsigned lg = 63 - __clzll((unsigned long long)idx);
                d = (size_t)1 << lg; k = (unsigned)(__ffsll((unsigned long long)m) - 1) - 1 - lg;
            }
            if (idx >= 2 && d >= 2) {
                size_t j = idx - d, lay = N >> k;
                E s0 = f[lay + j * stride], s1 = f[lay + (j + d) * stride];
                uint64_t ex = d / 2 - 1;
                E v0 = F::pow_u64(F::mul_add(den[2 * k + 1], s0, den[2 * k]), ex);
                E v1 = F::pow_u64(F::mul_add(den[2 * k + 1], s1, den[2 * k]), ex);
                E a = v0, b = F::mul(s0, v0), c = v1, dd = F::mul(s1, v1);
                if (decompose) {
                    E di = F::inv(F::sub(F::mul(a, dd), F::mul(b, c)));
                    r00 = F::mul(dd, di); r01 = F::mul(F::neg(b), di); r10 = F::mul(F::neg(c), di); r11 = F::mul(a, di);
                } else { r00 = a; r01 = b; r10 = c; r11 = dd; }
                if (!standard) { r00 = F::to_mont(r00); r01 = F::to_mont(r01); r10 = F::to_mont(r10); r11 = F::to_mont(r11); }
            }
            out[4 * idx] = r00; out[4 * idx + 1] = r01; out[4 * idx + 2] = r10; out[4 * idx + 3] = r11;
        });
        return hipGetLastError() == hipSuccess && hipStreamSynchronize(s) == hipSuccess;
    }

    // f of T_m (BinaryTree<F>, 2m entries, heap order; entry 0 unused = 0): every (N/m)-th element of each layer of the top
    // tree's point set (src/fftree.rs:471-478), gathered on the device.  out: 2m elements on the device, plain.  Synchronous.
    bool gather_f(unsigned log_m, E* out, hipStream_t s) const {
        if (!f_) return false;
        const size_t m = (size_t)1 << log_m, N = N_, stride = N_ / m;
        const E* f = f_;
        foreach_n(s, 2 * m, [=] __device__(size_t idx) {
            if (idx == 0) { out[0] = F::zero(); return; }
            const unsigned lg = 63 - __clzll((unsigned long long)idx);     // layer with 2^lg points: idx in [2^lg, 2^(lg+1))
            const size_t sz = (size_t)1 << lg, j = idx - sz;
            out[idx] = f[(N / m) * sz + j * stride];                       // that layer of the top tree has N*sz/m points
        });
        return hipGetLastError() == hipSuccess && hipStreamSynchronize(s) == hipSuccess;
    }

    // out[i] = f(x[i], y[i], T[t_off + i*t_stride]) with T one of the reference's plain tables of T_m (data x
    // plain table needs no Montgomery correction).  mode 0: x*T   1: x*T + y   2: y - x*T   3: (y - x)*T.
    // The pointwise steps of the multi-GPU ENTER / EXIT are built from this (Python model: tests/split_model.py).
    bool table_fma(E* out, const E* x, const E* y, size_t cnt, unsigned log_m, int which, size_t t_off, size_t t_stride, int mode, hipStream_t s) const {
        const Tree& T = trees_[log_m];
        const E* tbl = nullptr; size_t len = 0;
        switch (which) {
            case 3: tbl = T.xnn; len = T.m; break;           // ECFFT_TBL_XNN_S
            case 4: tbl = T.xnn_inv; len = T.m; break;
            case 5: tbl = T.z0_s1; len = T.e; break;
            case 6: tbl = T.z1_s0; len = T.e; break;
            case 7: tbl = T.z0_inv_s1; len = T.e; break;
            case 8: tbl = T.z1_inv_s0; len = T.e; break;
            case 9: tbl = T.z0z0; len = T.m; break;
            case 10: tbl = T.z1z1; len = T.m; break;
            default: return false;
        }
        if (cnt && t_off + (cnt - 1) * t_stride >= len) return false;
        if (mode < 0 || mode > 3 || (mode != 0 && !y)) return false;
        foreach_n(s, cnt, [=] __device__(size_t i) {
            E t = tbl[t_off + i * t_stride];
            E r;
            if (mode == 0) r = F::mul(t, x[i]);
            else if (mode == 1) r = F::mul_add(t, x[i], y[i]);
            else if (mode == 2) r = F::sub(y[i], F::mul(t, x[i]));
            else r = F::mul(t, F::sub(y[i], x[i]));
            out[i] = r;
        });
        return hipGetLastError() == hipSuccess;
    }

    E* scratch() const { return scratch_; }

private:
    // ---- memory ----
    E* take(size_t n) {
        size_t a = (n + 7) & ~(size_t)7;
        if (arena_used_ + a > arena_cap_) { fprintf(stderr, "ecfft: internal error: table arena overflow\n"); throw DeviceAllocError(); }   // sized exactly in build(); unreachable
        E* p = arena_ + arena_used_; arena_used_ += a; return p;
    }
    static constexpr size_t kTeElems = sizeof(TE) / sizeof(E);
    static_assert(sizeof(TE) % sizeof(E) == 0, "table element must be a whole number of field elements");
    // n table constants in the arena, filled from the plain values src[0..n)
    TE* to_tables(const E* src, size_t n, hipStream_t s) {
        TE* d = reinterpret_cast<TE*>(take(n * kTeElems));
        foreach_n(s, n, [=] __device__(size_t i) { d[i] = F::to_table(src[i]); });
        return d;
    }
    // construction temporaries come from one slab (bump-allocated, reset after every tree); anything that does not fit,
    // and every temporary of the public algorithm wrappers, is an individual allocation released by finish_api()
    E* temp(size_t n) {
        size_t a = (n + 7) & ~(size_t)7;
        if (slab_ && slab_used_ + a <= slab_cap_) { E* q = slab_ + slab_used_; slab_used_ += a; return q; }
        // pooled individual allocations: finish_api() / build_tree() hand them back to the pool instead of freeing them,
        // so repeated algorithm calls (ecfft_redc, ecfft_degree, ...) do not hipMalloc / hipFree every time.  Best fit, but a
        // request never takes a block more than 4x its size (or 64 KiB): the 16-byte accumulator of ecfft_degree used to grab
        // a multi-GiB block and force a fresh allocation for the next large request.
        size_t bytes = (a ? a : 8) * sizeof(E);
        const size_t cap_fit = bytes * 4 > ((size_t)64 << 10) ? bytes * 4 : ((size_t)64 << 10);
        int best = -1;
        for (size_t i = 0; i < pool_.size(); ++i)
            if (!pool_[i].busy && pool_[i].bytes >= bytes && pool_[i].bytes <= cap_fit && (best < 0 || pool_[i].bytes < pool_[(size_t)best].bytes)) best = (int)i;
        if (best >= 0) { pool_[(size_t)best].busy = true; pool_[(size_t)best].idle_calls = 0; return (E*)pool_[(size_t)best].p; }
        void* p = nullptr;
        if (hipMalloc(&p, bytes) != hipSuccess) {
            (void)hipGetLastError();
            temps_trim(0);                                    // give back every idle pooled block and try once more
            if (hipMalloc(&p, bytes) != hipSuccess) { (void)hipGetLastError(); fprintf(stderr, "ecfft: temporary allocation of %zu bytes failed\n", bytes); throw DeviceAllocError(); }
        }
        pool_.push_back({p, bytes, true, false, 0});
        return (E*)p;
    }
    // End of a call: every pooled temporary becomes reusable.  The pool is bounded: idle blocks beyond `keep` bytes in total
    // (twice the transform scratch, at least 256 MiB) are returned to the device, largest first — one large ecfft_vanish /
    // ecfft_degree no longer pins 6-10 n elements of HBM for the life of the context (ecfft_ctx_trim returns all of it).  Pinned blocks (the temporaries of a sharded transform whose shape the ranks have agreed on) are never trimmed.
    void temps_done() {
        for (auto& b : pool_) { if (!b.busy) ++b.idle_calls; b.busy = false; }
        size_t keep = 2 * scratch_cap_ * sizeof(E);
        if (keep < ((size_t)256 << 20)) keep = (size_t)256 << 20;
        temps_trim(keep);                                    // rare: hipFree waits for the device
    }
    size_t pool_bytes() const { size_t t = 0; for (const auto& b : pool_) t += b.bytes; return t; }
    // frees idle, unpinned blocks: those idle for > max_idle calls, then the largest ones until the idle total is <= keep bytes
    void temps_trim(size_t keep, unsigned max_idle = ~0u) {
        for (size_t i = 0; i < pool_.size();) {
            if (!pool_[i].busy && !pool_[i].pinned && pool_[i].idle_calls > max_idle) { (void)hipFree(pool_[i].p); pool_[i] = pool_.back(); pool_.pop_back(); }
            else ++i;
        }
        for (;;) {
            size_t idle = 0; int big = -1;
            for (size_t i = 0; i < pool_.size(); ++i)
                if (!pool_[i].busy && !pool_[i].pinned) { idle += pool_[i].bytes; if (big < 0 || pool_[i].bytes > pool_[(size_t)big].bytes) big = (int)i; }
            if (idle <= keep || big < 0) break;
            (void)hipFree(pool_[(size_t)big].p); pool_[(size_t)big] = pool_.back(); pool_.pop_back();
        }
    }
    void temps_free() { for (auto& b : pool_) (void)hipFree(b.p); pool_.clear(); }
    void release() {
        temps_free();
        full_cyclic_free();
        if (arena_) (void)hipFree(arena_);
        if (slab_) { (void)hipFree(slab_); slab_ = nullptr; slab_cap_ = slab_used_ = 0; }
        if (scratch_) (void)hipFree(scratch_);
        if (d_trees_) (void)hipFree(d_trees_);
        d_trees_ = nullptr;
        for (int i = 0; i < nside_; ++i) { (void)hipStreamDestroy(sides_[i]); (void)hipEventDestroy(ev_fork_[i]); (void)hipEventDestroy(ev_join_[i]); }
        nside_ = 0;
        arena_ = nullptr; scratch_ = nullptr;
    }

    bool ensure_scratch(size_t nt) {
        const size_t need = scratch_need(nt, kSplitDepth) > 5 * nt ? scratch_need(nt, kSplitDepth) : 5 * nt;
        if (scratch_cap_ >= need) return true;
        if (scratch_) { (void)hipDeviceSynchronize(); (void)hipFree(scratch_); scratch_ = nullptr; scratch_cap_ = 0; }
        if (hipMalloc(&scratch_, need * sizeof(E)) != hipSuccess) { fprintf(stderr, "ecfft: scratch allocation of %zu elements failed\n", need); return false; }
        scratch_cap_ = need;
        return true;
    }
    bool finish_api(hipStream_t s) {
        bool ok = hipGetLastError() == hipSuccess && hipStreamSynchronize(s) == hipSuccess;
        temps_done();
        return ok;
    }
    // user table a (crate representation, 2e entries) -> plain even entries (to be inverted) and plain odd entries
    void plain_halves(const E* a, E* a_even, E* a_odd, size_t e, hipStream_t s) {
        const E rinv = rinv_;
        foreach_n(s, e, [=] __device__(size_t i) { a_even[i] = F::mul(a[2 * i], rinv); a_odd[i] = F::mul(a[2 * i + 1], rinv); });
    }

    // ---- construction-time device primitives (plain data) ----
    // out[i] = 1/in[i]; chunks share one Fermat inversion (Montgomery's trick).  Zero entries stay zero and do not disturb
    // their neighbours, like ark_ff::batch_inversion (used at src/fftree.rs:235, 331-333, 409-414).  The inversion is a
    // dependent chain of ~270 multiplies, i.e. its LATENCY is what a call costs; the chunk length grows with n so that at most
    // one wave per SIMD runs such a chain (more waves would only share the issue slots and stretch every chain), and the
    // prefix products live in `out` / a temporary instead of per-thread arrays (round 2: 528 B of scratch per thread).
    // Several independent arrays in ONE launch (the inversion chain's latency is paid once, not once per table).
    struct InvSeg { const E* in; E* out; size_t n; };
    struct InvSegs { const E* in[6]; E* out[6]; E* pre[6]; size_t n[6], first[7]; int k; };
    void batch_inv(const E* in, E* out, size_t n, hipStream_t s) { const InvSeg g{in, out, n}; batch_inv_multi(&g, 1, s); }
    void batch_inv_multi(const InvSeg* segs, int k, hipStream_t s) {
        size_t total = 0;
        for (int j = 0; j < k; ++j) total += segs[j].n;
        if (!total || k > 6) return;
        size_t CH = 8;
        while (CH < 64 && (total + CH - 1) / CH + (size_t)k > (size_t)65536) CH *= 2;
        InvSegs g{}; g.k = 0; g.first[0] = 0;
        for (int j = 0; j < k; ++j) {
            if (!segs[j].n) continue;
            const int q = g.k++;
            g.in[q] = segs[j].in; g.out[q] = segs[j].out; g.n[q] = segs[j].n;
            g.pre[q] = (segs[j].in == segs[j].out) ? temp(segs[j].n) : segs[j].out;   // in-place calls keep their inputs until the backward pass
            g.first[q + 1] = g.first[q] + (segs[j].n + CH - 1) / CH;                  // chunks never straddle two arrays
        }
        foreach_n(s, g.first[g.k], [=] __device__(size_t t) {
            int q = 0;
            while (q + 1 < g.k && t >= g.first[q + 1]) ++q;
            const E* in = g.in[q]; E* out = g.out[q]; E* pre = g.pre[q];
            const size_t n = g.n[q], b = (t - g.first[q]) * CH, cnt = (b + CH <= n) ? CH : n - b;
            E acc = F::one();
            for (size_t i = 0; i < cnt; ++i) { const E v = in[b + i]; pre[b + i] = acc; if (!F::is_zero(v)) acc = F::mul(acc, v); }
            acc = F::inv(acc);
            for (size_t i = cnt; i-- > 0;) {
                const E v = in[b + i];
                if (F::is_zero(v)) { out[b + i] = F::zero(); continue; }
                const E r = F::mul(acc, pre[b + i]);
                acc = F::mul(acc, v);
                out[b + i] = r;
            }
        });
    }
    void ew_mul(E* out, const E* a, const E* b, size_t n, hipStream_t s) {
        foreach_n(s, n, [=] __device__(size_t i) { out[i] = F::mul(a[i], b[i]); });
    }
    // FFTree::extend for construction (whole-vector form with pre/post scaling)
    void b_extend(unsigned log_m, const E* in, E* out, size_t count, int target, hipStream_t s) {
        extend(in, out, trees_[log_m].e, count, target, s);
    }
    // mextend (src/fftree.rs:128-141)
    // The compositions below run on PLAIN data during construction (mont = false) and on user data in the crate's
    // Montgomery form for the public wrappers (mont = true): tables that are ADDED to data are converted with to_mont,
    // data x data products get the extra factor R^-1 (rinv_), data x table products need nothing.
    void b_mextend(unsigned log_m, const E* in, E* out, size_t count, int target, hipStream_t s, bool mont = false) {
        const Tree& T = trees_[log_m];
        b_extend(log_m, in, out, count, target, s);
        const E* z = target == 1 ? T.z0_s1 : T.z1_s0;
        size_t mask = T.e - 1;
        foreach_n(s, T.e * count, [=] __device__(size_t i) { E zz = z[i & mask]; out[i] = F::add(out[i], mont ? F::to_mont(zz) : zz); });
    }
    // redc_impl (src/fftree.rs:232-259); a0inv/a1: e PLAIN entries; evals/out: m entries
    void b_redc(unsigned log_m, const E* evals, const E* a0inv, const E* a1, E* out, int moiety, hipStream_t s) {
        const Tree& T = trees_[log_m];
        size_t e = T.e;
        E* t0 = temp(e); E* h1 = temp(e);
        foreach_n(s, e, [=] __device__(size_t i) { t0[i] = F::mul(evals[2 * i], a0inv[i]); });
        b_extend(log_m, t0, t0, 1, 1 - moiety, s);                           // g1 = extend_impl(t0, opposite moiety)  (:239-245)
        const E* zinv = moiety == 0 ? T.z0_inv_s1 : T.z1_inv_s0;            // :247-250
        foreach_n(s, e, [=] __device__(size_t i) {
            h1[i] = F::mul(F::sub(evals[2 * i + 1], F::mul(t0[i], a1[i])), zinv[i]);
        });
        b_extend(log_m, h1, t0, 1, moiety, s);                               // h0 = extend_impl(h1, moiety)  (:256)
        foreach_n(s, e, [=] __device__(size_t i) { out[2 * i] = t0[i]; out[2 * i + 1] = h1[i]; });
    }
    void b_redc_s0(unsigned log_m, const E* evals, const E* a0inv, const E* a1, E* out, hipStream_t s) { b_redc(log_m, evals, a0inv, a1, out, 0, s); }
    // modular_reduce_impl (src/fftree.rs:277-281); c PLAIN
    void b_modular_reduce(unsigned log_m, const E* evals, const E* a0inv, const E* a1, const E* c, E* out, hipStream_t s) {
        size_t m = trees_[log_m].m;
        E* h = temp(m);
        b_redc_s0(log_m, evals, a0inv, a1, h, s);
        ew_mul(h, h, c, m, s);
        b_redc_s0(log_m, h, a0inv, a1, out, s);
    }
    // vanish_impl (src/fftree.rs:291-308), bottom-up: dom has e = m/2 entries, out m entries
    void b_vanish(unsigned log_m, const E* dom, E* out, hipStream_t s, bool mont = false) {
        const Tree& T = trees_[log_m];
        size_t e = T.e, m = T.m;
        E* Q = temp(m); E* Q2 = temp(m); E* q0 = temp(e); E* q1 = temp(e);
        const E* f = f_; size_t N = N_; const E rinv = rinv_;
        foreach_n(s, e, [=] __device__(size_t i) {        // T_2 leaves are the top tree's leaves 0 and N/2
            E l0 = f[N], l1 = f[N + N / 2];
            if (mont) { l0 = F::to_mont(l0); l1 = F::to_mont(l1); }
            Q[2 * i] = F::sub(dom[i], l0); Q[2 * i + 1] = F::sub(dom[i], l1);
        });
        unsigned le = ilog2(e);
        for (unsigned r = 1; r <= le; ++r) {
            size_t bs = (size_t)1 << r;                     // size of the blocks being merged
            foreach_n(s, e, [=] __device__(size_t g) {
                size_t b = g >> r, i = g & (bs - 1);
                E pr = F::mul(Q[(2 * b) * bs + i], Q[(2 * b + 1) * bs + i]);
                q0[g] = mont ? F::mul(pr, rinv) : pr;
            });
            b_mextend(r + 1, q0, q1, e >> r, 1, s, mont);
            E* dstQ = (r == le) ? out : Q2;
            foreach_n(s, e, [=] __device__(size_t g) { dstQ[2 * g] = q0[g]; dstQ[2 * g + 1] = q1[g]; });
            E* t = Q; Q = Q2; Q2 = t;
        }
        if (le == 0) (void)hipMemcpyAsync(out, Q, m * sizeof(E), hipMemcpyDeviceToDevice, s);
    }

    // matrix-core tables of the innermost 16-point map (mfma_blk16.h), per source parity: 32-byte fields, trees with e >= 16
    static size_t blk16_elems(unsigned l) { return (sizeof(E) == 32 && l >= 5) ? 2 * (Blk16::kArenaElems + 8) : 0; }
    void build_blk16(Tree& T, hipStream_t s) {
        T.blk16_A[0] = T.blk16_A[1] = nullptr; T.blk16_K[0] = T.blk16_K[1] = nullptr;
        if constexpr (sizeof(E) == 32) {
            if (T.e < 16 || mfma_off_) return;
            Blk16BuildArgs ba[2];
            for (int sg = 0; sg < 2; ++sg) {
                uint8_t* A = reinterpret_cast<uint8_t*>(take(Blk16::kArenaElems));
                unsigned long long* K = reinterpret_cast<unsigned long long*>(A + Blk16::kABytes);
                ba[sg] = {T.np0[sg], T.dinv[sg], T.p0[1 - sg], T.p1[1 - sg], T.inner[sg], A, K};
                T.blk16_A[sg] = A; T.blk16_K[sg] = K;
            }
            hipLaunchKernelGGL(k_blk16_build, dim3(2), dim3(256), 0, s, ba[0], ba[1], T.e);       // both source parities in one launch
        }
    }

    // The four lowest levels of ENTER and of EXIT as ONE 16 x 16 map each for the matrix cores (LevelTables::low16_A, read by the
    // 1024-element low-level kernels): every 16-block of a transform goes through the same levels on the same tables, so the maps
    // are the images of the 16 unit vectors under this context's own level code (16 transforms of 16 points in one batched call).
    static size_t low16_elems() { return sizeof(E) == 32 ? 2 * (Blk16::kArenaElems + 8) + 2 * (Blk16::kArenaElems32 + 8) : 0; }   // + the two low32 maps
    bool build_low16(unsigned l_top, hipStream_t s) {
        if constexpr (sizeof(E) == 32) {
            if (l_top < kLogLowSmall || mfma_off_ || low16_off_) return true;
            Tree& T = trees_[4];
            E* I = temp(256); E* O = temp(256);
            foreach_n(s, 256, [=] __device__(size_t j) { I[j] = ((j >> 4) == (j & 15)) ? F::one() : F::zero(); });
            for (int dir = 0; dir < 2; ++dir) {
                if (dir == 0) enter_levels(I, O, 16, 16, s, scratch_, 1, 4); else exit_levels(I, O, 16, 16, s, scratch_, 4, 1);
                uint8_t* A = reinterpret_cast<uint8_t*>(take(Blk16::kArenaElems));
                unsigned long long* K = reinterpret_cast<unsigned long long*>(A + Blk16::kABytes);
                hipLaunchKernelGGL(k_blk16_from_matrix, dim3(1), dim3(256), 0, s, O, A, K, true);
                T.low16_A[dir] = A; T.low16_K[dir] = K;
            }
            // round 4: the FIVE lowest levels as one 32 x 32 map per 32-block, for the 1024-element low-level kernels (low32_mask_:
            // bit 0 ENTER, bit 1 EXIT) — the images of the 32 unit vectors under the level code (32 transforms of 32 points)
            if (l_top >= kLogLow && low32_mask_) {
                Tree& T5 = trees_[5];
                E* I32 = temp(1024); E* O32 = temp(1024); E* CS = temp(1024);
                foreach_n(s, 1024, [=] __device__(size_t j) { I32[j] = ((j >> 5) == (j & 31)) ? F::one() : F::zero(); });
                for (int dir = 0; dir < 2; ++dir) {
                    if (!((low32_mask_ >> dir) & 1u)) continue;
                    if (dir == 0) enter_levels(I32, O32, 32, 32, s, scratch_, 1, 5); else exit_levels(I32, O32, 32, 32, s, scratch_, 5, 1);
                    uint8_t* A = reinterpret_cast<uint8_t*>(take(Blk16::kArenaElems32));
                    unsigned long long* K = reinterpret_cast<unsigned long long*>(A + Blk16::kABytes32);
                    hipLaunchKernelGGL(k_blk32_expand, dim3(4), dim3(256), 0, s, (const E*)O32, A, CS, true);
                    hipLaunchKernelGGL(k_blk32_seeds, dim3(1), dim3(32), 0, s, (const E*)CS, K);
                    T5.low32_A[dir] = A; T5.low32_K[dir] = K;
                }
            }
        }
        return hipGetLastError() == hipSuccess;
    }

    bool build_tree(unsigned l, hipStream_t s) {
        Tree& T = trees_[l];
        size_t m = (size_t)1 << l, e = m / 2;
        T.m = m; T.e = e; T.log_m = l;
        const E* f = f_; size_t N = N_; size_t stride = N_ / m;
        T.xnn = take(m); T.xnn_inv = take(m);
        {
            E* xnn = T.xnn; uint64_t ex = m / 2;
            foreach_n(s, m, [=] __device__(size_t j) { xnn[j] = F::pow_u64(f[N + j * stride], ex); });
            if (l == 0) batch_inv(T.xnn, T.xnn_inv, m, s);      // otherwise inverted together with the stage tables below
        }
        if (l == 0) return true;
        unsigned le = ilog2(e);
        size_t es = e > 1 ? e : 1;
        // plain values first (temporaries), then the kernels' table form in the arena
        E *hp0[2], *hp1[2], *hnp0[2], *hdinv[2], *hw[2], *hwinv[2];
        for (int sg = 0; sg < 2; ++sg) {
            hp0[sg] = temp(es); hp1[sg] = temp(es); hnp0[sg] = temp(es); hdinv[sg] = temp(es);
            hw[sg] = temp(es); hwinv[sg] = temp(es);
            E *p0 = hp0[sg], *p1 = hp1[sg], *np0 = hnp0[sg], *dinv = hdinv[sg];
            if (e > 1) {
                // entry g of the concatenated stage tables: stage k = number of leading ones ... computed by scan
                foreach_n(s, e - 1, [=] __device__(size_t g) {
                    // stage k occupies [e - 2h, e - h), h = e >> (k+1)
                    size_t rem = e - g;                       // in (h, 2h]
                    unsigned k = 0; size_t h = e >> 1;
                    while (rem <= h) { h >>= 1; ++k; }
                    size_t i = g - (e - 2 * h);
                    size_t lay = N >> k;                      // offset (= size) of layer k in the top tree
                    E a = f[lay + (2 * i + sg) * stride];
                    E b = f[lay + (2 * i + sg + 2 * h) * stride];
                    p0[g] = a; p1[g] = b; np0[g] = F::neg(a); dinv[g] = F::sub(b, a);
                });
            }
            // normalisation weights W(s) of the leaves of parity sg (DESIGN.md "Normalised butterflies")
            E* w = hw[sg]; const E* den = den_;
            foreach_n(s, e, [=] __device__(size_t i) {
                size_t j = 2 * i + sg;
                E U = F::one(), C = F::one();
                for (unsigned b = 0; b + 1 < le; ++b) {
                    size_t lsz = m >> b;                      // |L_b| of T_m
                    E sb = f[(N >> b) + (j & (lsz - 1)) * stride];
                    E V = F::mul_add(den[2 * b + 1], sb, den[2 * b]);
                    C = F::mul(C, V);
                    U = F::mul(F::sqr(U), C);
                }
                w[i] = U;
            });
        }
        E *xq = nullptr, *xqi = nullptr;                        // X^(m/4) on the leaves of T_m and its inverse (src/fftree.rs:328-330), used below
        if (l >= 2) { xq = temp(m); xqi = temp(m); E* q = xq; uint64_t ex = m / 4; foreach_n(s, m, [=] __device__(size_t j) { q[j] = F::pow_u64(f[N + j * stride], ex); }); }
        {   // 1/xnn_s, 1/(s1 - s0) and 1/W of both parities, 1/X^(m/4): six independent arrays, ONE inversion launch
            const InvSeg segs[6] = {{T.xnn, T.xnn_inv, m}, {hdinv[0], hdinv[0], e - 1}, {hdinv[1], hdinv[1], e - 1}, {hw[0], hwinv[0], e}, {hw[1], hwinv[1], e},
                                    {xq, xqi, l >= 2 ? m : 0}};
            batch_inv_multi(segs, 6, s);
        }
        // merged innermost stage pair (h = 1, stage k = le-1): out_j = a + c_j*(b - a) with
        // c_j = (p_j^target - p_0^source) / (p_1^source - p_0^source), table offset e-2 (kernels.h)
        for (int sg = 0; sg < 2; ++sg) {
            E* in = temp(2);
            if (e > 1) {
                const E *sp0 = hp0[sg] + (e - 2), *sdi = hdinv[sg] + (e - 2), *tp0 = hp0[1 - sg] + (e - 2), *tp1 = hp1[1 - sg] + (e - 2);
                foreach_n(s, 1, [=] __device__(size_t) {
                    in[0] = F::mul(F::sub(tp0[0], sp0[0]), sdi[0]);
                    in[1] = F::mul(F::sub(tp1[0], sp0[0]), sdi[0]);
                });
            } else {
                (void)hipMemsetAsync(in, 0, 2 * sizeof(E), s);
            }
            T.inner[sg] = to_tables(in, 2, s);
        }
        for (int sg = 0; sg < 2; ++sg) {   // c0t = np0 * dinv (pair-split decompose, kernels.h lds_extend_core)
            E* c0 = temp(es);
            if (e > 1) { const E *n0 = hnp0[sg], *di = hdinv[sg]; foreach_n(s, e - 1, [=] __device__(size_t g) { c0[g] = F::mul(n0[g], di[g]); }); }
            else (void)hipMemsetAsync(c0, 0, sizeof(E), s);
            if (e > 1) (void)hipMemsetAsync(c0 + (e - 1), 0, sizeof(E), s);
            T.c0t[sg] = to_tables(c0, es, s);
        }
        for (int sg = 0; sg < 2; ++sg) {
            if (e == 1) {   // no butterfly stage: the (never read) stage tables still get defined contents
                (void)hipMemsetAsync(hp0[sg], 0, sizeof(E), s); (void)hipMemsetAsync(hp1[sg], 0, sizeof(E), s);
                (void)hipMemsetAsync(hnp0[sg], 0, sizeof(E), s); (void)hipMemsetAsync(hdinv[sg], 0, sizeof(E), s);
            }
            T.p0[sg] = to_tables(hp0[sg], es, s); T.p1[sg] = to_tables(hp1[sg], es, s);
            T.np0[sg] = to_tables(hnp0[sg], es, s); T.dinv[sg] = to_tables(hdinv[sg], es, s);
            T.w[sg] = to_tables(hw[sg], es, s); T.winv[sg] = to_tables(hwinv[sg], es, s);
        }
        build_blk16(T, s);
        T.z0_s1 = take(es); T.z1_s0 = take(es); T.z0_inv_s1 = take(es); T.z1_inv_s0 = take(es);
        T.z0z0 = take(m); T.z1z1 = take(m);
        if (l == 1) {                                          // base cases src/fftree.rs:399-403, 454-458
            E *z0 = T.z0_s1, *z1 = T.z1_s0, *zz0 = T.z0z0, *zz1 = T.z1z1;
            foreach_n(s, 1, [=] __device__(size_t) {
                E s0 = f[N], s1 = f[N + stride];
                z0[0] = F::sub(s1, s0); z1[0] = F::sub(s0, s1);
                zz0[0] = zz0[1] = F::sqr(s0); zz1[0] = zz1[1] = F::sqr(s1);
            });
        } else {
            const Tree& S = trees_[l - 1];
            // z0_s1 (src/fftree.rs:386-393)
            E* a = temp(e); E* b = temp(e);
            {
                const E *sz0 = S.z0_s1, *sz1 = S.z1_s0;
                foreach_n(s, e / 2, [=] __device__(size_t i) {
                    a[2 * i] = F::zero(); a[2 * i + 1] = sz0[i];
                    b[2 * i] = sz1[i]; b[2 * i + 1] = F::zero();
                });
            }
            b_extend(l, a, a, 1, 1, s);
            b_extend(l, b, b, 1, 1, s);
            ew_mul(T.z0_s1, a, b, e, s);
            // z1_s0 = vanish(S1)[even]  (src/fftree.rs:396-397)
            E* s1 = temp(e); E* van = temp(m);
            foreach_n(s, e, [=] __device__(size_t i) { s1[i] = f[N + (2 * i + 1) * stride]; });
            b_vanish(l, s1, van, s);
            { E* z1 = T.z1_s0; foreach_n(s, e, [=] __device__(size_t i) { z1[i] = van[2 * i]; }); }
        }
        { const InvSeg segs[2] = {{T.z0_s1, T.z0_inv_s1, e}, {T.z1_s0, T.z1_inv_s0, e}}; batch_inv_multi(segs, 2, s); }
        if (l >= 2) {
            const Tree& S = trees_[l - 1];
            // a0inv / a1 views of xnn_s for REDC on this tree and on the subtree
            E* xa0i = temp(e); E* xa1 = temp(e);
            { const E *xi = T.xnn_inv, *x = T.xnn; foreach_n(s, e, [=] __device__(size_t i) { xa0i[i] = xi[2 * i]; xa1[i] = x[2 * i + 1]; }); }
            E* sxa0i = temp(e / 2 ? e / 2 : 1); E* sxa1 = temp(e / 2 ? e / 2 : 1);
            { const E *xi = S.xnn_inv, *x = S.xnn; foreach_n(s, e / 2, [=] __device__(size_t i) { sxa0i[i] = xi[2 * i]; sxa1[i] = x[2 * i + 1]; }); }
            // X^(m/4) tables on the leaves of T_m (src/fftree.rs:328-330)
            E* xqa0i = temp(e); E* xqa1 = temp(e);
            foreach_n(s, e, [=] __device__(size_t i) { xqa0i[i] = xqi[2 * i]; xqa1[i] = xq[2 * i + 1]; });
            // z0z0_rem_xnn_s (src/fftree.rs:418-446)
            E* sq = temp(e); E* zz0 = temp(e); E* zz1 = temp(e); E* zz = temp(m); E* tmp = temp(m); E* dr = temp(m);
            ew_mul(sq, S.z0z0, S.z1z1, e, s);                                       // :421-423
            b_modular_reduce(l - 1, sq, sxa0i, sxa1, S.z0z0, zz0, s);               // :424-425
            b_extend(l, zz0, zz1, 1, 1, s);                                         // :426
            foreach_n(s, e, [=] __device__(size_t i) { zz[2 * i] = zz0[i]; zz[2 * i + 1] = zz1[i]; });
            {
                const E *z0s1 = T.z0_s1, *xnn = T.xnn;
                foreach_n(s, m, [=] __device__(size_t i) {                           // :430-438
                    E z0 = (i & 1) ? z0s1[i / 2] : F::zero();
                    E y = F::sub(z0, xnn[i]);
                    tmp[i] = F::mul(F::sub(F::sqr(y), zz[i]), xqi[i]);
                });
            }
            b_modular_reduce(l, tmp, xqa0i, xqa1, zz, dr, s);                        // :439-440
            { E* o = T.z0z0; foreach_n(s, m, [=] __device__(size_t i) { o[i] = F::mul_add(xq[i], dr[i], zz[i]); }); } // :441-446
            // z1z1_rem_xnn_s (src/fftree.rs:449-452)
            {
                const E *z1s0 = T.z1_s0, *xnn = T.xnn;
                foreach_n(s, m, [=] __device__(size_t i) {
                    E z1 = (i & 1) ? F::zero() : z1s0[i / 2];
                    tmp[i] = F::sqr(F::sub(z1, xnn[i]));
                });
            }
            b_modular_reduce(l, tmp, xa0i, xa1, T.z0z0, T.z1z1, s);
        }
        // fused pointwise tables of the hot path
        {
            E *xe = temp(es), *w1x = temp(es), *A1 = temp(es), *B1 = temp(es), *NB2 = temp(es), *C1 = temp(es), *D1 = temp(es), *xie = temp(es);
            const E *xnn = T.xnn, *xi = T.xnn_inv, *w1 = hw[1], *wi0 = hwinv[0], *wi1 = hwinv[1], *zi = T.z0_inv_s1, *c = T.z0z0;
            foreach_n(s, e, [=] __device__(size_t i) {
                E xo = xnn[2 * i + 1], xiev = xi[2 * i], z = zi[i];
                xe[i] = xnn[2 * i];
                w1x[i] = F::mul(w1[i], xo);
                xie[i] = xiev;
                A1[i] = F::mul(xiev, wi0[i]);
                B1[i] = F::mul(z, wi1[i]);
                NB2[i] = F::neg(F::mul(xo, z));
                C1[i] = F::mul(c[2 * i], xiev);
                D1[i] = F::mul(c[2 * i + 1], z);
            });
            T.xe = to_tables(xe, es, s); T.w1x = to_tables(w1x, es, s); T.A1 = to_tables(A1, es, s); T.B1 = to_tables(B1, es, s);
            T.NB2 = to_tables(NB2, es, s); T.C1 = to_tables(C1, es, s); T.D1 = to_tables(D1, es, s); T.xie = to_tables(xie, es, s);
        }
        hipError_t err = hipGetLastError();
        if (err != hipSuccess) { fprintf(stderr, "ecfft: kernel launch failed: %s\n", hipGetErrorString(err)); return false; }
        // temporaries are only needed until the stream drains; free them per tree to bound memory
        if (hipStreamSynchronize(s) != hipSuccess) return false;
        temps_free();
        slab_used_ = 0;
        return true;
    }

    HostTree<F> host_;
    E rinv_ = F::inv(F::to_mont(F::one()));   // R^-1 as a plain residue (1 for M31)
    size_t N_ = 0; unsigned L_ = 0; int device_ = 0;
    E* arena_ = nullptr; size_t arena_cap_ = 0, arena_used_ = 0;
    E* f_ = nullptr; E* den_ = nullptr; E* scratch_ = nullptr; size_t scratch_cap_ = 0;
    E* slab_ = nullptr; size_t slab_cap_ = 0, slab_used_ = 0;
    std::vector<Tree> trees_;
    Tree* d_trees_ = nullptr;
    struct PoolBuf { void* p; size_t bytes; bool busy; bool pinned; unsigned idle_calls; };
    std::set<uint64_t> agreed_shapes_;                // (op, variant, world, len) of the collective calls the ranks have voted on
    bool fail_next_collective_ = false;
    std::vector<PoolBuf> pool_;
    std::mutex mu_;
    mutable Profiler prof_;
    hipStream_t sides_[kMaxSides] = {}; hipEvent_t ev_fork_[kMaxSides] = {}, ev_join_[kMaxSides] = {}; int nside_ = 0;
    bool bad_points_ = false;
    int shard_kind_ = kShardNone;                           // sharded EXTEND-only / ENTER-only context (build_*_shard)
    unsigned shard_log_p_ = 0, shard_rank_ = 0;
    std::vector<ShardSet> sets_;                            // per tree: the rank's share of its EXTEND tables (shard contexts)
    const Tree* ovr_tree_ = nullptr; const ShardSet* ovr_set_ = nullptr;   // temporary share of one tree (sharded EXIT build)
    mutable double tblw_ = 1.0;     // weight of table bytes in the algorithmic-byte accounting (see enter())
    mutable std::map<uint64_t, TE*> full_cyc_; mutable size_t full_cyc_bytes_ = 0;   // compact cyclic tables of a FULL context (full_cyclic_table)
    bool full_cyc_off_ = ab_env("ECFFT_NO_FULL_CYCLIC") != nullptr;      // A/B switch: one-stage launches with stride-P table reads
    // bit 0: ENTER levels 1..5, bit 1: EXIT levels 5..1 as one 32-point map (low32).  OFF by default: measured on MI355X (profiles/r04/low32_ab.txt)
    // the 32-point phase costs what it removes — level 5 is four 16-point phases + five pointwise steps (~35 us per tile), the phase
    // streams 1 MiB of matrices per tile and runs twice the MFMAs of low16 (k_exit_low 1.63 -> 1.65-1.69 ms, k_enter_low 0.66 -> 0.66)
    unsigned low32_mask_ = ab_env("ECFFT_LOW32") ? (unsigned)atoi(ab_env("ECFFT_LOW32")) : 0u;
    bool low16_off_ = ab_env("ECFFT_NO_LOW16") != nullptr;             // A/B switch: the four lowest ENTER / EXIT levels as VALU sweeps
    bool mfma_off_ = ab_env("ECFFT_NO_MFMA") != nullptr;                // A/B switch: innermost stages on the VALU instead of the matrix cores
    unsigned small_min_logc_ = ab_env("ECFFT_SMALL_MIN_LOGC") ? (unsigned)atoi(ab_env("ECFFT_SMALL_MIN_LOGC")) : 1u;   // log2 of the shortest column-tile row of a small launch (rows of 2 elements: 7 stages in one pass; A/B knob)
    bool in_halves_ = false;                                            // enqueueing the two-halves schedule of one transform (caller holds lock())
    Tree pair_full_{}; bool have_pair_full_ = false;                    // EXIT-shard contexts: the full tree T_2c (c = n / world) of the redundant pair level
    // full contexts: split EXITs of at most 2^this run every top level redundantly after one all-gather (0: never) — api_exit_split
    unsigned gather_max_log_ = ab_env("ECFFT_SPLIT_GATHER_MAX_LOG") ? (unsigned)atoi(ab_env("ECFFT_SPLIT_GATHER_MAX_LOG")) : 21u;
    // link striping of the big pairwise exchanges (Transport::exchange_striped): staging of the call in flight; A/B switches (test builds)
    mutable void* stripe_stage_ = nullptr; mutable size_t stripe_bytes_ = 0;
    struct StripeScope { const DeviceChain* ch; StripeScope(const DeviceChain* c, void* p, size_t b) : ch(c) { ch->stripe_stage_ = p; ch->stripe_bytes_ = p ? b : 0; }
                         ~StripeScope() { ch->stripe_stage_ = nullptr; ch->stripe_bytes_ = 0; } };
    bool stripe_off_ = ab_env("ECFFT_NO_STRIPE") != nullptr;
    size_t stripe_gain_ = ab_env("ECFFT_STRIPE_MIN_GAIN") ? (size_t)atoll(ab_env("ECFFT_STRIPE_MIN_GAIN")) : ~(size_t)0;
    bool q2_split_ = ab_env("ECFFT_SPLIT_Q2_SPLIT") != nullptr;        // A/B switch (full contexts): the pair level of a split EXIT as four split EXTENDs
    bool col256_off_ = ab_env("ECFFT_NO_COL256") != nullptr;            // A/B switch: small column passes on the generic kernels (pair-split LDS sweeps)
    bool row256_off_ = ab_env("ECFFT_NO_ROW256") != nullptr;            // A/B switch: small row passes on the generic kernel (pair-split LDS sweeps)
    bool ef_small_off_ = ab_env("ECFFT_NO_SMALL_TILES") != nullptr;   // A/B switch for the small-launch tile rule
};

}  // namespace ecfft
