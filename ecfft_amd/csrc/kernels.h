// HIP kernels of the ECFFT hot path (gfx950).  Field-generic: F = ecfft::Secp256k1 or ecfft::M31.
//
// All butterflies are the NORMALISED form derived in DESIGN.md ("Normalised butterflies"): the
// reference's 2x2 matrices (src/fftree.rs:355-362)
//      R = [[v0, s0*v0], [v1, s1*v1]],   D = R^-1,   v_j = v(s_j)^(d/2-1)
// factor as R = diag(v0, v1) * [[1, s0], [1, s1]].  Carrying the diagonal as a per-point weight W
// (W_k(s) = v_k(s)^(h_k-1) * W_{k+1}(psi_k(s))) turns every stage into
//      recombine:  (A, B) -> (A + s0*B, A + s1*B)                       2 field muls, was 4
//      decompose:  (a, b) -> q1 = (b - a)/(s1 - s0), q0 = a - s0*q1      2 field muls, was 4
// with one multiply by 1/W on the way in and by W on the way out of an EXTEND (both folded into the
// neighbouring pointwise tables).  Field elements are canonical residues, so the re-association is
// bit-exact against the reference.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>
#include <type_traits>
#include "mfma_blk16.h"

namespace ecfft {

constexpr int kBlock = 256;

// Workgroup barrier that publishes LDS writes only: __syncthreads() also drains the vector-memory counter (s_waitcnt vmcnt(0)),
// which would turn every table request issued ahead of the barrier into a wait AT the barrier.
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
// Round 6: sweeps whose pairs stay inside one wave's own 128-element span (one pair per thread, pair distance <= 64: the wave that
// wrote an element is the only one that reads it next) need no WORKGROUP barrier between them.  LDS operations of one wave execute
// in issue order, so a wavefront-scope fence (no instruction; it only stops the compiler from moving LDS accesses across it) is all
// the ordering there is to ask for.  ECFFT_WAVE_LOCAL: 0 = workgroup barrier everywhere (rounds 1-5), 1 = wavefront fence,
// 2 = s_waitcnt lgkmcnt(0) without the s_barrier (A/B forms; profiles/r06/wave_local_ab.txt).
#ifndef ECFFT_WAVE_LOCAL
#define ECFFT_WAVE_LOCAL 1
#endif
__device__ __forceinline__ void wave_local_sync() {
#if ECFFT_WAVE_LOCAL == 2
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#elif ECFFT_WAVE_LOCAL == 1
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    asm volatile("" ::: "memory");
#else
    lds_barrier();
#endif
}
// pair distance 2^lh of a one-pair-per-thread sweep (idx = ((tid >> lh) << (lh + 1)) + (tid & (h - 1))): wave-local iff h <= 64
__device__ __forceinline__ constexpr bool wave_local_lh(int lh) { return ECFFT_WAVE_LOCAL != 0 && lh <= 6; }
#ifndef ECFFT_RADIX_LDSBAR
#define ECFFT_RADIX_LDSBAR 1                 // 4-byte engine: the barrier behind the early table requests is LDS-only
#endif

// Value ranges: inside a kernel (registers, LDS) field elements may be in the field's LAZY range (F::tmul / F::tmul_add /
// F::sub keep them there: [0, p] for M31, canonical for secp256k1); every store to HBM goes through F::canon(), so all
// arrays between launches, and everything the caller sees, are canonical residues.
// ---------------------------------------------------------------------------------------------
// streaming butterfly stages (one launch per stage).  On the hot path only the cyclic shards of a
// multi-GPU split EXTEND use them (log2 P stages with strided tables); everything else is fused below.
// buf holds `npairs*2` elements = count vectors of length e laid end to end; h = pair distance.
// ---------------------------------------------------------------------------------------------
template <class F>
__global__ __launch_bounds__(kBlock) void k_decompose_stage(typename F::elem* __restrict__ buf,
                                                             const typename F::telem* __restrict__ np0,
                                                             const typename F::telem* __restrict__ dinv,
                                                             uint32_t log_h, size_t npairs, uint32_t tstride, uint32_t toff) {
    // tstride/toff: table entry of local pair index i is i*tstride + toff (cyclic shards of a split EXTEND; 1/0 otherwise)
    size_t g = (size_t)blockIdx.x * kBlock + threadIdx.x;
    if (g >= npairs) return;
    size_t h = (size_t)1 << log_h;
    size_t i = g & (h - 1);
    size_t idx = ((g >> log_h) << (log_h + 1)) + i;
    i = i * tstride + toff;
    typename F::elem a = buf[idx], b = buf[idx + h];
    typename F::elem q1 = F::tmul(dinv[i], F::sub(b, a));
    typename F::elem q0 = F::tmul_add(np0[i], q1, a);
    buf[idx] = F::canon(q0);
    buf[idx + h] = F::canon(q1);
}

template <class F>
__global__ __launch_bounds__(kBlock) void k_recombine_stage(typename F::elem* __restrict__ buf,
                                                             const typename F::telem* __restrict__ p0,
                                                             const typename F::telem* __restrict__ p1,
                                                             uint32_t log_h, size_t npairs, uint32_t tstride, uint32_t toff) {
    size_t g = (size_t)blockIdx.x * kBlock + threadIdx.x;
    if (g >= npairs) return;
    size_t h = (size_t)1 << log_h;
    size_t i = g & (h - 1);
    size_t idx = ((g >> log_h) << (log_h + 1)) + i;
    i = i * tstride + toff;
    typename F::elem a = buf[idx], b = buf[idx + h];
    buf[idx] = F::canon(F::tmul_add(p0[i], b, a));
    buf[idx + h] = F::canon(F::tmul_add(p1[i], b, a));
}

// ---------------------------------------------------------------------------------------------
// Fused load / store operators.  Every pointwise step of ENTER / EXIT that sits between two EXTEND
// cores is folded into the first load or the last store of the neighbouring fused-stage kernel, so it
// costs no HBM pass of its own.  `pos` is the element's position in the work buffer (count vectors of
// length e laid end to end), i = pos mod e.
// ---------------------------------------------------------------------------------------------
enum { LD_PLAIN = 0, LD_SCALE = 1 };
enum { ST_PLAIN = 0, ST_SCALE = 1, ST_AXPBY = 2, ST_EXIT_SPLIT = 3, ST_ENTER = 4 };

template <class F>
struct IoDesc {
    using E = typename F::elem;
    using TE = typename F::telem;
    // load:  x = [ld_tbl[i] *] src[src_stride*pos + src_off]
    //        ld_tr_logp > 0: src holds the P = 2^ld_tr_logp chunks of an all-to-all, source-rank major, and position pos of the
    //        block shard is element pos / P of chunk pos mod P:  src[(pos mod P)*tr_chunk + pos / P]   (split EXTEND)
    const E* src; uint32_t src_stride, src_off; int ld_mode; const TE* ld_tbl; uint32_t ld_tr_logp, st_tr_logp; size_t tr_chunk;
    // store: ST_PLAIN  dst[pos] = x
    //        ST_SCALE  dst[pos] = st_a[i]*x
    //        ST_AXPBY  r = st_a[i]*x + st_b[i]*aux[aux_stride*pos + aux_off]; dst[pos] = r; aux_out[pos] = r (if set)
    //        ST_EXIT_SPLIT  u0 = st_a[i]*x; v0 = st_b[i]*(aux[2*pos] - u0); dst[b*2e + i] = u0; dst[b*2e + e + i] = v0  (b = pos / e)
    //        ST_ENTER  (pair operator, ENTER loop C src/fftree.rs:155-159 in normalised form; x = U1~ at pos = b*2e + i, y = V1~ at
    //                  pos + e, both still in the tile; aux = the level's input blocks [u0 | v0])
    //                  dst[b*2e + 2i] = aux[b*2e + i] + st_a[i]*aux[b*2e + e + i];  dst[b*2e + 2i + 1] = st_c[i]*x + st_b[i]*y
    E* dst; int st_mode; const TE* st_a; const TE* st_b; const E* aux; uint32_t aux_stride, aux_off; E* aux_out; const TE* st_c;
    uint32_t nt_st;      // != 0: non-temporal stores of the pass's results (data_st; launches of the latency regime)
};

// Result stores of the fused passes (32-byte fields) go through data_st: two explicit 16-byte vector stores, non-temporal when the host
// sets IoDesc::nt_st for the launch (DeviceChain::kNtStoreMaxLog; wave-uniform branch).  Round 6, measured on MI355X
// (profiles/r06/nt_data_ab.txt): the NON-TEMPORAL store itself buys nothing — flag set for launches <= 2^17 .. 2^19: +-0.3 % against the flag
// never set; set everywhere: +1.5 .. +3 % from 2^20 on (the next pass wants the results in cache); non-temporal LOADS of the tile data
// lose everywhere — so the shipped library never sets the flag.  What DOES pay is this form of the store: with it hipcc allocates the
// latency-regime column kernels differently (k_stages_col256 77 -> 101 VGPRs, k_stages_col_mid256 102 -> 94; same instructions, more loads
// in flight) and single transforms of 2^12 .. 2^19 run 0.7 - 3.2 % faster than with a plain `dst[pos] = x` (2^16 -0.7 %, 2^17 -1.5 %,
// 2^18 -1.4 %, 2^19 -3.2 %; >= 2^20 and batches unchanged), interleaved on one box against the previous sources.
typedef uint32_t nt_v4u __attribute__((ext_vector_type(4)));
template <class E>
__device__ __forceinline__ void data_st(E* p, const E& x, uint32_t nt) {
    if constexpr (sizeof(E) == 32) {
        if (nt) {                                                        // wave-uniform
            nt_v4u* q = reinterpret_cast<nt_v4u*>(p);
            nt_v4u a = {x.l[0], x.l[1], x.l[2], x.l[3]}, b = {x.l[4], x.l[5], x.l[6], x.l[7]};
            __builtin_nontemporal_store(a, q); __builtin_nontemporal_store(b, q + 1);
            return;
        }
    }
    *p = x;
}
template <class F>
__device__ __forceinline__ typename F::elem io_load(const IoDesc<F>& io, size_t pos, size_t emask) {
    typename F::elem v = io.ld_tr_logp ? io.src[(pos & (((size_t)1 << io.ld_tr_logp) - 1)) * io.tr_chunk + (pos >> io.ld_tr_logp)]
                                       : io.src[(size_t)io.src_stride * pos + io.src_off];
    if (io.ld_mode == LD_SCALE) v = F::tmul(io.ld_tbl[pos & emask], v);
    return v;
}
template <class F>
__device__ __forceinline__ void io_store(const IoDesc<F>& io, size_t pos, uint32_t log_e, const typename F::elem& x) {
    using E = typename F::elem;
    const size_t emask = ((size_t)1 << log_e) - 1, i = pos & emask;
    switch (io.st_mode) {
        case ST_PLAIN:   // st_tr_logp > 0: the mirror image of the transposed load — scatter into the send chunks of an all-to-all
            if (io.st_tr_logp) io.dst[(pos & (((size_t)1 << io.st_tr_logp) - 1)) * io.tr_chunk + (pos >> io.st_tr_logp)] = F::canon(x);
            else data_st(&io.dst[pos], F::canon(x), io.nt_st);
            break;
        case ST_SCALE: data_st(&io.dst[pos], F::canon(F::tmul(io.st_a[i], x)), io.nt_st); break;
        case ST_AXPBY: {
            E r = F::canon(F::tmul_add(io.st_a[i], x, F::tmul(io.st_b[i], io.aux[(size_t)io.aux_stride * pos + io.aux_off])));
            data_st(&io.dst[pos], r, io.nt_st);
            if (io.aux_out) io.aux_out[pos] = r;
            break;
        }
        default: {  // ST_EXIT_SPLIT
            E u0 = F::canon(F::tmul(io.st_a[i], x));
            E v0 = F::canon(F::tmul(io.st_b[i], F::sub(io.aux[2 * pos], u0)));
            size_t base = (pos >> log_e) << (log_e + 1);
            data_st(&io.dst[base + i], u0, io.nt_st);
            data_st(&io.dst[base + ((size_t)1 << log_e) + i], v0, io.nt_st);
        }
    }
}

// the store operator of one EXTEND core followed by the load operator of the next one, applied to a value in flight
// (ST_PLAIN / ST_SCALE / ST_AXPBY only; side output aux_out is written): used where two cores are fused in one launch
template <class F>
__device__ __forceinline__ typename F::elem io_mid(const IoDesc<F>& io, size_t pos, size_t emask, const typename F::elem& x) {
    using E = typename F::elem;
    const size_t i = pos & emask;
    E r = x;
    if (io.st_mode == ST_SCALE) r = F::tmul(io.st_a[i], x);
    else if (io.st_mode == ST_AXPBY) {
        r = F::canon(F::tmul_add(io.st_a[i], x, F::tmul(io.st_b[i], io.aux[(size_t)io.aux_stride * pos + io.aux_off])));
        if (io.aux_out) io.aux_out[pos] = r;
    }
    if (io.ld_mode == LD_SCALE) r = F::tmul(io.ld_tbl[i], r);
    return r;
}

// table entry `idx` of a table whose base pointer is wave-uniform: 32-bit BYTE offset, so the load is "SGPR base + 32-bit lane
// offset" (global_load ... v_off, s[base:base+1]) instead of a 64-bit per-lane address built with v_lshl_add_u64
// The pointer is also cast to the GLOBAL address space: table pointers that were themselves loaded from memory (LevelTables) are
// generic, and generic (flat_load) accesses count against the LDS counter as well as the vector-memory one.
template <class TE>
__device__ __forceinline__ TE ldt(const TE* __restrict__ base, uint32_t idx) {
#ifdef ECFFT_EXP_NO_TABLE_LOADS      // experiment: constants made up in registers (no VMEM) — the bound of what free table loads could reach
    { TE r; uint32_t* w = reinterpret_cast<uint32_t*>(&r);
#pragma unroll
      for (int k = 0; k < (int)(sizeof(TE) / 4); ++k) w[k] = (idx + 0x01010101u * (uint32_t)k) >> (sizeof(TE) == 4 ? 2 : 1);
      return r; }
#endif
    typedef const __attribute__((address_space(1))) char* gchar;
    if constexpr (sizeof(TE) == 4) {
        typedef const __attribute__((address_space(1))) TE* gte;
        return *(gte)((gchar)(reinterpret_cast<const char*>(base)) + (size_t)(idx * (uint32_t)sizeof(TE)));
    } else {
        // multi-word constants (secp256k1: 64 bytes): 16-byte global loads; 64-bit offset (tables may exceed 4 GiB)
        static_assert(sizeof(TE) % 16 == 0, "table constants are whole 16-byte words");
        typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
        typedef const __attribute__((address_space(1))) u32x4* gq;
        const gq p = (gq)((gchar)(reinterpret_cast<const char*>(base)) + (size_t)idx * sizeof(TE));
        TE r;
        u32x4* rp = reinterpret_cast<u32x4*>(&r);
#pragma unroll
        for (int k = 0; k < (int)(sizeof(TE) / 16); ++k) rp[k] = p[k];
        return r;
    }
}
// ---------------------------------------------------------------------------------------------
// Tile load / store for 4-byte fields, four consecutive elements ("quad") at a time.  The element-at-a-time loops of the
// generic kernels compile to a loop that waits for every 4-byte load before issuing the next one — 16 serial HBM round
// trips per thread for an 8192-element tile.  Here a thread issues ALL its 16-byte loads (data, strided gathers, operator
// tables) first, then computes, then stores 16 bytes per instruction.  `pos_of(j)` maps the LDS index j (a multiple of 4,
// j..j+3 contiguous in LDS) to the position of the quad in the work buffer (contiguous there too).
// Preconditions checked by vio_ok(): e >= 4, 16-byte aligned pointers, strides as produced by the level drivers.
// ---------------------------------------------------------------------------------------------
struct Quad { uint32_t v[4]; };
#ifdef ECFFT_EXP_TILE_IO_L2       // experiment: every tile load / store of the 4-byte vector paths lands in one 64 KiB window (cache resident):
#define ECFFT_DPOS(p) ((p) & (size_t)0x3FFF)      // the work stays, the HBM latency and bandwidth go — what a perfect prefetch ring could reach
#else
#define ECFFT_DPOS(p) (p)
#endif
__device__ __forceinline__ Quad ldq(const uint32_t* p) { uint4 t = *reinterpret_cast<const uint4*>(p); return Quad{{t.x, t.y, t.z, t.w}}; }
__device__ __forceinline__ void stq(uint32_t* p, const Quad& q) { *reinterpret_cast<uint4*>(p) = make_uint4(q.v[0], q.v[1], q.v[2], q.v[3]); }
// four consecutive 4-byte table entries starting at entry idx (16-byte aligned), uniform base pointer in the GLOBAL address space + 32-bit byte offset (see ldt below)
__device__ __forceinline__ Quad ldq_tab(const uint32_t* __restrict__ base, uint32_t idx) {
    typedef const __attribute__((address_space(1))) char* gchar;
    typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
    typedef const __attribute__((address_space(1))) u32x4* gq;
    const u32x4 t = *(gq)((gchar)(reinterpret_cast<const char*>(base)) + (size_t)(idx * 4u));
    return Quad{{t.x, t.y, t.z, t.w}};
}
// four elements at p[stride*k + off], k = 0..3, stride 1 (off 0) or 2 (off 0 / 1)
__device__ __forceinline__ Quad ldq_strided(const uint32_t* p, uint32_t stride, uint32_t off) {
    if (stride == 1) return ldq(p);
    uint4 a = *reinterpret_cast<const uint4*>(p), b = *reinterpret_cast<const uint4*>(p + 4);
    return off ? Quad{{a.y, a.w, b.y, b.w}} : Quad{{a.x, a.z, b.x, b.z}};
}
template <class F>
__device__ __forceinline__ bool vio_ok(const IoDesc<F>& io, uint32_t log_e) {
    if constexpr (sizeof(typename F::elem) != 4) return false;
    auto al = [](const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; };
    bool ok = log_e >= 2 && io.ld_tr_logp == 0 && io.st_tr_logp == 0 && al(io.src) && al(io.dst) && ((io.src_stride == 1 && io.src_off == 0) || (io.src_stride == 2 && io.src_off < 2));
    if (io.ld_mode == LD_SCALE) ok = ok && al(io.ld_tbl);
    if (io.st_mode == ST_SCALE || io.st_mode == ST_AXPBY || io.st_mode == ST_EXIT_SPLIT) ok = ok && al(io.st_a);
    if (io.st_mode == ST_AXPBY || io.st_mode == ST_EXIT_SPLIT) ok = ok && al(io.st_b) && al(io.aux) && ((io.aux_stride == 1 && io.aux_off == 0) || (io.aux_stride == 2 && io.aux_off < 2));
    if (io.st_mode == ST_AXPBY && io.aux_out) ok = ok && al(io.aux_out);
    if (io.st_mode == ST_EXIT_SPLIT) ok = ok && io.aux_stride == 2 && io.aux_off == 0;
    return ok;
}
template <class F, int NQ, int BLK, class PosFn>
__device__ __forceinline__ void vio_load(const IoDesc<F>& io, size_t emask, typename F::elem* tile, PosFn pos_of, uint32_t tid0) {
    const uint32_t tid = tid0;   // callers that split a tile into batches pass tid + batch*NQ*BLK (quad c of the batch = index 4*(tid + c*BLK))
    Quad d[NQ], t[NQ];
#pragma unroll
    for (int c = 0; c < NQ; ++c) { const size_t pos = pos_of(4u * (tid + (uint32_t)c * BLK)); d[c] = ldq_strided(io.src + (size_t)io.src_stride * ECFFT_DPOS(pos), io.src_stride, io.src_off); }
    if (io.ld_mode == LD_SCALE) {
#pragma unroll
        for (int c = 0; c < NQ; ++c) { const size_t pos = pos_of(4u * (tid + (uint32_t)c * BLK)); t[c] = ldq_tab(io.ld_tbl, (uint32_t)(pos & emask)); }
#pragma unroll
        for (int c = 0; c < NQ; ++c)
#pragma unroll
            for (int k = 0; k < 4; ++k) d[c].v[k] = F::tmul(t[c].v[k], d[c].v[k]);
    }
#pragma unroll
    for (int c = 0; c < NQ; ++c) stq(tile + 4u * (tid + (uint32_t)c * BLK), d[c]);
}
template <class F, int NQ, int BLK, class PosFn>
__device__ __forceinline__ void vio_store(const IoDesc<F>& io, uint32_t log_e, const typename F::elem* tile, PosFn pos_of, uint32_t tid) {
    const size_t e = (size_t)1 << log_e, emask = e - 1;
    Quad x[NQ], a[NQ], b[NQ], y[NQ];
#pragma unroll
    for (int c = 0; c < NQ; ++c) x[c] = ldq(tile + 4u * (tid + (uint32_t)c * BLK));
    const int m = io.st_mode;
    if (m != ST_PLAIN) {
#pragma unroll
        for (int c = 0; c < NQ; ++c) { const size_t pos = pos_of(4u * (tid + (uint32_t)c * BLK)); a[c] = ldq_tab(io.st_a, (uint32_t)(pos & emask)); }
    }
    if (m == ST_AXPBY || m == ST_EXIT_SPLIT) {
#pragma unroll
        for (int c = 0; c < NQ; ++c) {
            const size_t pos = pos_of(4u * (tid + (uint32_t)c * BLK));
            b[c] = ldq_tab(io.st_b, (uint32_t)(pos & emask));
            y[c] = ldq_strided(io.aux + (size_t)io.aux_stride * ECFFT_DPOS(pos), io.aux_stride, io.aux_off);
        }
    }
#pragma unroll
    for (int c = 0; c < NQ; ++c) {
        const size_t pos = pos_of(4u * (tid + (uint32_t)c * BLK));
        Quad r, r2;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            if (m == ST_PLAIN) r.v[k] = F::canon(x[c].v[k]);
            else if (m == ST_SCALE) r.v[k] = F::canon(F::tmul(a[c].v[k], x[c].v[k]));
            else if (m == ST_AXPBY) r.v[k] = F::canon(F::tmul_add(a[c].v[k], x[c].v[k], F::tmul(b[c].v[k], y[c].v[k])));
            else { r.v[k] = F::canon(F::tmul(a[c].v[k], x[c].v[k])); r2.v[k] = F::canon(F::tmul(b[c].v[k], F::sub(y[c].v[k], r.v[k]))); }
        }
        if (m == ST_EXIT_SPLIT) {
            const size_t bs = (pos >> log_e) << (log_e + 1), i = pos & emask;
            stq(io.dst + ECFFT_DPOS(bs + i), r); stq(io.dst + ECFFT_DPOS(bs + e + i), r2);
        } else {
            stq(io.dst + ECFFT_DPOS(pos), r);
            if (m == ST_AXPBY && io.aux_out) stq(io.aux_out + ECFFT_DPOS(pos), r);
        }
    }
}

// io_mid on quads, in place in LDS: the store operator of one EXTEND core (ST_PLAIN / ST_SCALE / ST_AXPBY, side output
// aux_out written) followed by the load operator of the next (LD_PLAIN / LD_SCALE)
template <class F, int NQ, int BLK, class PosFn>
__device__ __forceinline__ void vio_mid(const IoDesc<F>& io, uint32_t log_e, typename F::elem* tile, PosFn pos_of, uint32_t tid) {
    const size_t emask = ((size_t)1 << log_e) - 1;
    Quad x[NQ], a[NQ], b[NQ], y[NQ], l[NQ];
    const int m = io.st_mode;
#pragma unroll
    for (int c = 0; c < NQ; ++c) {
        const size_t pos = pos_of(4u * (tid + (uint32_t)c * BLK)), i = pos & emask;
        x[c] = ldq(tile + 4u * (tid + (uint32_t)c * BLK));
        if (m != ST_PLAIN) a[c] = ldq_tab(io.st_a, (uint32_t)i);
        if (m == ST_AXPBY) { b[c] = ldq_tab(io.st_b, (uint32_t)i); y[c] = ldq_strided(io.aux + (size_t)io.aux_stride * pos, io.aux_stride, io.aux_off); }
        if (io.ld_mode == LD_SCALE) l[c] = ldq_tab(io.ld_tbl, (uint32_t)i);
    }
#pragma unroll
    for (int c = 0; c < NQ; ++c) {
        const size_t pos = pos_of(4u * (tid + (uint32_t)c * BLK));
        Quad r;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            uint32_t v = x[c].v[k];
            if (m == ST_SCALE) v = F::tmul(a[c].v[k], v);
            else if (m == ST_AXPBY) v = F::canon(F::tmul_add(a[c].v[k], v, F::tmul(b[c].v[k], y[c].v[k])));
            r.v[k] = v;
        }
        if (m == ST_AXPBY && io.aux_out) stq(io.aux_out + pos, r);
        if (io.ld_mode == LD_SCALE) {
#pragma unroll
            for (int k = 0; k < 4; ++k) r.v[k] = F::tmul(l[c].v[k], r.v[k]);
        }
        stq(tile + 4u * (tid + (uint32_t)c * BLK), r);
    }
}

// ENTER combine (ST_ENTER) on quads: NQ quads of pairs per thread; for quad c: ju / jv = LDS indices of U1~ / V1~, i = pair
// index inside the block (multiple of 4), bb = position of the block [u0 | v0] in the level's input / output
template <class F, int NQ, class IdxFn>
__device__ __forceinline__ void vio_enter_store(const typename F::elem* tile, const typename F::elem* __restrict__ src, typename F::elem* __restrict__ dst,
                                                const typename F::telem* __restrict__ xe, const typename F::telem* __restrict__ w1,
                                                const typename F::telem* __restrict__ w1x, size_t e, IdxFn idx, int cbase = 0) {
    // cbase: first quad of this batch (callers run several small batches: 7 live quads per pair-quad is 28 VGPRs)
    Quad u0[NQ], v0[NQ], tx[NQ], tw[NQ], twx[NQ], U[NQ], V[NQ];
#pragma unroll
    for (int c = 0; c < NQ; ++c) {
        uint32_t ju, jv; size_t i, bb; idx(cbase + c, ju, jv, i, bb);
        u0[c] = ldq(src + ECFFT_DPOS(bb + i)); v0[c] = ldq(src + ECFFT_DPOS(bb + e + i));
        tx[c] = ldq_tab(xe, (uint32_t)i); tw[c] = ldq_tab(w1, (uint32_t)i); twx[c] = ldq_tab(w1x, (uint32_t)i);
        U[c] = ldq(tile + ju); V[c] = ldq(tile + jv);
    }
#pragma unroll
    for (int c = 0; c < NQ; ++c) {
        uint32_t ju, jv; size_t i, bb; idx(cbase + c, ju, jv, i, bb);
        Quad lo, hi;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const uint32_t ev = F::canon(F::tmul_add(tx[c].v[k], v0[c].v[k], u0[c].v[k]));
            const uint32_t od = F::canon(F::tmul_add(twx[c].v[k], V[c].v[k], F::tmul(tw[c].v[k], U[c].v[k])));
            if (k < 2) { lo.v[2 * k] = ev; lo.v[2 * k + 1] = od; } else { hi.v[2 * (k - 2)] = ev; hi.v[2 * (k - 2) + 1] = od; }
        }
        stq(dst + ECFFT_DPOS(bb + 2 * i), lo); stq(dst + ECFFT_DPOS(bb + 2 * i + 4), hi);
    }
}

// ---------------------------------------------------------------------------------------------
// LDS-fused butterfly stages, "row kernel".  One workgroup owns a contiguous tile of 2^log_tile
// elements (<= 64 KiB of LDS: 2048 secp256k1 / 16384 M31 elements), loads it once, runs every
// decompose stage k in [k_first, log e) and then every recombine stage back down to k_first in LDS,
// and stores it once: 2*(log e - k_first) stages for one HBM round trip.  Stage k_first has pair
// distance h = e >> (k_first+1) with 2h <= tile and tiles are tile-aligned, so a butterfly's table
// index is its local pair index mod h: every tile of a level reads the SAME h-entry table prefix
// (L2-resident per XCD).
// ---------------------------------------------------------------------------------------------
#ifndef ECFFT_BLOCK_LDS
#define ECFFT_BLOCK_LDS 512
#endif
constexpr int kBlockLds = ECFFT_BLOCK_LDS;   // threads per workgroup of the LDS-fused kernels
#ifndef ECFFT_BLOCK_ROW
#define ECFFT_BLOCK_ROW ECFFT_BLOCK_LDS
#endif
constexpr int kBlockRow = ECFFT_BLOCK_ROW;   // ... of the row kernel (k_stages_lds)
#ifndef ECFFT_ROW_PIPE
#define ECFFT_ROW_PIPE 1                     // row-kernel sweeps request the next sweep's table constants before the barrier
#endif
#ifndef ECFFT_MIN_WAVES
#define ECFFT_MIN_WAVES 4                    // waves per SIMD the register allocator must leave room for
#endif

// ---------------------------------------------------------------------------------------------
// One butterfly stage over an LDS-resident array: `npairs` pairs at distance h = 2^lh, table entry = pair index mod h.
// DEC: (a, b) -> (a + ta*q1, q1) with q1 = tb*(b - a)   [ta = np0, tb = dinv];   else (a, b) -> (a + ta*b, a + tb*b) [p0, p1].
// 4-byte fields (M31) take 4 consecutive pairs per lane with 128-bit LDS and table accesses whenever h >= 4 (the pairs,
// their partners and their table entries are then contiguous and 16-byte aligned): 4x fewer memory instructions and
// index computations.  No trailing barrier.
// ---------------------------------------------------------------------------------------------
// PAIR-SPLIT form (32-byte fields, sweeps with at most BLK/2 pairs — the small tiles of latency-bound launches): threads
// [0, npairs) produce the LOW output of every pair, threads [npairs, 2*npairs) the HIGH one, ONE multiply each (decompose through
// tc = np0*dinv, which makes its two products independent), so a sweep costs one multiply of latency instead of two dependent
// ones and twice as many SIMDs work.  Both roles read (a, b); a barrier separates those reads from the in-place writes.
template <class F, bool DEC, int BLK = kBlockLds>
__device__ __forceinline__ void stage_sweep(typename F::elem* a_, const typename F::telem* __restrict__ ta, const typename F::telem* __restrict__ tb,
                                            uint32_t lh, uint32_t npairs, uint32_t tid, const typename F::telem* __restrict__ tc = nullptr) {
    using E = typename F::elem;
    const uint32_t h = 1u << lh;
    if constexpr (sizeof(E) == 32) {
        if (2 * npairs <= (uint32_t)BLK && (!DEC || tc)) {
            const bool act = tid < 2 * npairs, hi = tid >= npairs;
            const uint32_t g = hi ? tid - npairs : tid, i = g & (h - 1), idx = ((g >> lh) << (lh + 1)) + i;
            E x, y; typename F::telem t;
            if (act) { x = a_[idx]; y = a_[idx + h]; t = DEC ? (hi ? ldt(tb, i) : ldt(tc, i)) : (hi ? ldt(tb, i) : ldt(ta, i)); }
            __syncthreads();
            if (act) {
                if (DEC) { const E d = F::sub(y, x); if (hi) a_[idx + h] = F::tmul(t, d); else a_[idx] = F::tmul_add(t, d, x); }
                else a_[idx + (hi ? h : 0)] = F::tmul_add(t, y, x);
            }
            return;
        }
    }
    if constexpr (sizeof(E) == 4) {
        if (lh >= 2 && (npairs & 3u) == 0) {
            for (uint32_t g4 = tid; g4 < (npairs >> 2); g4 += BLK) {
                const uint32_t g = g4 << 2, i = g & (h - 1), idx = ((g >> lh) << (lh + 1)) + i;
                uint4 va = *reinterpret_cast<const uint4*>(a_ + idx), vb = *reinterpret_cast<const uint4*>(a_ + idx + h);
                const uint4 v0 = *reinterpret_cast<const uint4*>(ta + i), v1 = *reinterpret_cast<const uint4*>(tb + i);
                uint32_t xa[4] = {va.x, va.y, va.z, va.w}, xb[4] = {vb.x, vb.y, vb.z, vb.w};
                const uint32_t t0[4] = {v0.x, v0.y, v0.z, v0.w}, t1[4] = {v1.x, v1.y, v1.z, v1.w};
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    if (DEC) { E q1 = F::tmul(t1[c], F::sub(xb[c], xa[c])); xa[c] = F::tmul_add(t0[c], q1, xa[c]); xb[c] = q1; }
                    else { E o0 = F::tmul_add(t0[c], xb[c], xa[c]), o1 = F::tmul_add(t1[c], xb[c], xa[c]); xa[c] = o0; xb[c] = o1; }
                }
                *reinterpret_cast<uint4*>(a_ + idx) = make_uint4(xa[0], xa[1], xa[2], xa[3]);
                *reinterpret_cast<uint4*>(a_ + idx + h) = make_uint4(xb[0], xb[1], xb[2], xb[3]);
            }
            return;
        }
    }
    for (uint32_t g = tid; g < npairs; g += BLK) {
        const uint32_t i = g & (h - 1), idx = ((g >> lh) << (lh + 1)) + i;
        E a = a_[idx], b = a_[idx + h];
        if (DEC) { E q1 = F::tmul(ldt(tb, i), F::sub(b, a)); a_[idx] = F::tmul_add(ldt(ta, i), q1, a); a_[idx + h] = q1; }
        else { a_[idx] = F::tmul_add(ldt(ta, i), b, a); a_[idx + h] = F::tmul_add(ldt(tb, i), b, a); }
    }
}


// ---------------------------------------------------------------------------------------------
// Register-resident multi-stage engine for 4-byte fields (M31).  A 4-byte element leaves most of the register file
// idle in the one-stage-per-LDS-round-trip sweeps above, and rocprofv3 showed the M31 row kernel waiting, not issuing
// (51 % of wave-cycles in s_waitcnt / s_barrier, one VALU instruction per 5.4 cycles per SIMD): every sweep paid a
// barrier, an LDS round trip and a dependent L2 table load for 14-17 arithmetic instructions per pair.  Here a thread
// keeps EPT = 16 (or 8) elements in registers and runs NS = 1..3 consecutive stages on them per LDS round trip
// ("radix-2^NS step": groups of 2^NS elements spaced by the smallest pair distance of the step), the table constants of
// the whole step are requested BEFORE the barrier that publishes the previous step (they do not depend on the data), and
// the lowest log2(EPT) decompose stages, the merged innermost stage and the first log2(EPT) recombine stages run on EPT
// consecutive elements in one visit with wave-uniform (scalar) table constants.  25 sweeps of an 8192-element tile become
// 7 visits.
// ---------------------------------------------------------------------------------------------
template <class F, bool DEC>
__device__ __forceinline__ void bfly(typename F::elem& a, typename F::elem& b, const typename F::telem& t0, const typename F::telem& t1) {
    using E = typename F::elem;
    if (DEC) { E q1 = F::tmul(t1, F::sub(b, a)); a = F::tmul_add(t0, q1, a); b = q1; }
    else { E o0 = F::tmul_add(t0, b, a), o1 = F::tmul_add(t1, b, a); a = o0; b = o1; }
}

// NS stages on NG groups of 2^NS elements: group g = LDS positions pbase[g] + j*pstride (j < 2^NS).  Stage s' < NS pairs
// (j, j + 2^s') and reads table entry (e - 2*(tstride << s')) + tbase[g] + (j mod 2^s')*tstride of ta / tb.
// DEC runs s' = NS-1 .. 0 (large distance first), recombine 0 .. NS-1.  Starts with the barrier that makes the previous
// step's LDS writes visible (after the table loads have been issued); does NOT end with one.
template <class F, int NS, bool DEC, int NG>
__device__ __forceinline__ void radix_step(typename F::elem* a, const typename F::telem* __restrict__ ta, const typename F::telem* __restrict__ tb,
                                           uint32_t e, const uint32_t (&pbase)[NG], uint32_t pstride, const uint32_t (&tbase)[NG], uint32_t tstride,
                                           uint32_t halves = 1, uint32_t hstride = 0) {
    // all table offsets are 32-bit (tables of one tree have < 2^31 entries): uniform base pointer + 32-bit lane offset.
    // halves > 1: the same step on `halves` arrays `hstride` elements apart that use the SAME table entries (two vectors of a
    // batched EXTEND): the constants are loaded once
    using E = typename F::elem;
    using TE = typename F::telem;
    constexpr int G = 1 << NS;
    TE t0[NG][G - 1], t1[NG][G - 1];                        // stage s' occupies [2^s' - 1, 2^(s'+1) - 1)
#pragma unroll
    for (int g = 0; g < NG; ++g) {
#pragma unroll
        for (int sp = 0; sp < NS; ++sp) {
            const uint32_t off = e - 2 * (tstride << sp) + tbase[g];
#pragma unroll
            for (int m = 0; m < (1 << sp); ++m) { t0[g][(1 << sp) - 1 + m] = ldt(ta, off + (uint32_t)m * tstride); t1[g][(1 << sp) - 1 + m] = ldt(tb, off + (uint32_t)m * tstride); }
        }
    }
    if (ECFFT_RADIX_LDSBAR) lds_barrier(); else __syncthreads();          // the constants requested above stay in flight across it
#pragma unroll 1
    for (uint32_t hf = 0; hf < halves; ++hf, a += hstride) {
        E x[NG][G];
#pragma unroll
        for (int g = 0; g < NG; ++g)
#pragma unroll
            for (int j = 0; j < G; ++j) x[g][j] = a[pbase[g] + (uint32_t)j * pstride];
#pragma unroll
        for (int st = 0; st < NS; ++st) {
            const int sp = DEC ? NS - 1 - st : st;
#pragma unroll
            for (int g = 0; g < NG; ++g)
#pragma unroll
                for (int j = 0; j < G; ++j)
                    if (!(j & (1 << sp))) bfly<F, DEC>(x[g][j], x[g][j + (1 << sp)], t0[g][(1 << sp) - 1 + (j & ((1 << sp) - 1))], t1[g][(1 << sp) - 1 + (j & ((1 << sp) - 1))]);
        }
#pragma unroll
        for (int g = 0; g < NG; ++g)
#pragma unroll
            for (int j = 0; j < G; ++j) a[pbase[g] + (uint32_t)j * pstride] = x[g][j];
    }
}

// `cnt` consecutive stages whose SMALLEST pair distance is 2^lh_low elements of a flat array (row layout: table entry =
// position mod pair distance), in radix-8 steps plus a radix-4 / radix-2 remainder.  DEC: lh_low + cnt - 1 down to lh_low.
template <class F, bool DEC, int EPT, int BLK>
__device__ __forceinline__ void flat_stages(typename F::elem* a, const typename F::telem* __restrict__ ta, const typename F::telem* __restrict__ tb,
                                            uint32_t e, uint32_t lh_low, uint32_t cnt, uint32_t tid) {
    // chunk sizes: as many 3s as possible, remainder first for DEC / last for recombine does not matter for correctness as
    // long as the order of stages is monotone; the chunks are walked from the far end for DEC
    uint32_t done = 0;
    while (done < cnt) {
        uint32_t ns = cnt - done >= 3 ? 3 : cnt - done;
        // DEC walks distances downwards: this chunk covers lh in [top - ns + 1, top], top = lh_low + cnt - 1 - done
        const uint32_t lo = DEC ? lh_low + cnt - done - ns : lh_low + done;
        const uint32_t hl = 1u << lo;
        auto run = [&](auto NSc) {
            constexpr int NS = decltype(NSc)::value;
            constexpr int NG = EPT >> NS;
            uint32_t pb[NG], tb0[NG];
#pragma unroll
            for (int g = 0; g < NG; ++g) {
                const uint32_t q = tid + (uint32_t)BLK * g, i0 = q & (hl - 1);
                pb[g] = ((q >> lo) << (lo + NS)) | i0; tb0[g] = i0;
            }
            radix_step<F, NS, DEC, NG>(a, ta, tb, e, pb, hl, tb0, hl);
        };
        if (ns == 3) run(std::integral_constant<int, 3>{}); else if (ns == 2) run(std::integral_constant<int, 2>{}); else run(std::integral_constant<int, 1>{});
        done += ns;
    }
}

// the lowest stages (pair distance < EPT) on EPT consecutive elements per thread: decompose lh = min(le, LOG_EPT)-1 .. 1, the
// merged innermost pair (h = 1), recombine 1 .. min(le, LOG_EPT)-1.  All table constants are wave-uniform.  Begins with a barrier.
template <class F, int EPT, int BLK>
__device__ __forceinline__ void tail_stages(typename F::elem* a, const typename F::telem* __restrict__ np0, const typename F::telem* __restrict__ dinv,
                                            const typename F::telem* __restrict__ p0, const typename F::telem* __restrict__ p1,
                                            const typename F::telem* __restrict__ inner, uint32_t e, uint32_t le, uint32_t tid) {
    using E = typename F::elem;
    static_assert(sizeof(E) == 4 && (EPT == 16 || EPT == 8), "4-byte fields, 8 or 16 elements per thread");
    constexpr int LOG_EPT = EPT == 16 ? 4 : 3;
    __syncthreads();
    if (le == 0) return;
    E x[EPT];
    uint4* va = reinterpret_cast<uint4*>(a + (size_t)tid * EPT);
#pragma unroll
    for (int q = 0; q < EPT / 4; ++q) { uint4 v = va[q]; x[4 * q] = v.x; x[4 * q + 1] = v.y; x[4 * q + 2] = v.z; x[4 * q + 3] = v.w; }
#pragma unroll
    for (int lh = LOG_EPT - 1; lh >= 1; --lh) {
        if ((uint32_t)lh < le) {
            const uint32_t off = e - 2 * (1u << lh);
#pragma unroll
            for (int j = 0; j < EPT; ++j)
                if (!(j & (1 << lh))) bfly<F, true>(x[j], x[j + (1 << lh)], ldt(np0, off + (j & ((1 << lh) - 1))), ldt(dinv, off + (j & ((1 << lh) - 1))));
        }
    }
    {
        const typename F::telem c0 = ldt(inner, 0), c1 = ldt(inner, 1);
#pragma unroll
        for (int j = 0; j < EPT; j += 2) { E d = F::sub(x[j + 1], x[j]); E o0 = F::tmul_add(c0, d, x[j]); x[j + 1] = F::tmul_add(c1, d, x[j]); x[j] = o0; }
    }
#pragma unroll
    for (int lh = 1; lh < LOG_EPT; ++lh) {
        if ((uint32_t)lh < le) {
            const uint32_t off = e - 2 * (1u << lh);
#pragma unroll
            for (int j = 0; j < EPT; ++j)
                if (!(j & (1 << lh))) bfly<F, false>(x[j], x[j + (1 << lh)], ldt(p0, off + (j & ((1 << lh) - 1))), ldt(p1, off + (j & ((1 << lh) - 1))));
        }
    }
#pragma unroll
    for (int q = 0; q < EPT / 4; ++q) va[q] = make_uint4(x[4 * q], x[4 * q + 1], x[4 * q + 2], x[4 * q + 3]);
}

// every in-tile stage of an EXTEND core on BLK*EPT LDS elements = vectors of length e laid end to end (or, for e larger
// than the tile, a tile-aligned piece of one: then only the stages with pair distance below 2^log_span = tile run here).
// Decompose from distance 2^(log_span-1) down, merged innermost pair, recombine back up.  Starts with a barrier, ends with one.
template <class F, int EPT, int BLK>
__device__ __forceinline__ void lds_extend_fast(typename F::elem* a, const typename F::telem* __restrict__ np0, const typename F::telem* __restrict__ dinv,
                                                const typename F::telem* __restrict__ p0, const typename F::telem* __restrict__ p1,
                                                const typename F::telem* __restrict__ inner, size_t e64, uint32_t log_span, uint32_t tid) {
    const uint32_t e = (uint32_t)e64;
    constexpr uint32_t LOG_EPT = EPT == 16 ? 4 : 3;
    const uint32_t upper = log_span > LOG_EPT ? log_span - LOG_EPT : 0;      // stages with pair distance >= EPT
    if (upper) flat_stages<F, true, EPT, BLK>(a, np0, dinv, e, LOG_EPT, upper, tid);
    tail_stages<F, EPT, BLK>(a, np0, dinv, p0, p1, inner, e, log_span, tid);
    if (upper) flat_stages<F, false, EPT, BLK>(a, p0, p1, e, LOG_EPT, upper, tid);
    __syncthreads();
}

// Row-kernel sweeps of a 32-byte field, one pair per thread (tile = 2 * BLK elements), with the two table constants of the NEXT
// sweep requested before the barrier that ends the current one (see col_stages_pipe).  Pair distances 2^lh_from .. 2^lh_to
// (downwards for DEC, upwards otherwise), table entry e - 2h + (pair index mod h).  Ends with a barrier.
// lh_after: pair-distance log of a one-pair-per-thread consumer that follows the last sweep (then the closing barrier is relaxed
// like the ones between wave-local sweeps), or 99 = the consumer reads other waves' elements.
template <class F, bool DEC>
__device__ __forceinline__ void row_stages_pipe(typename F::elem* tile, const typename F::telem* __restrict__ ta, const typename F::telem* __restrict__ tb,
                                                uint32_t e, int lh_from, int lh_to, uint32_t tid, int lh_after = 99) {
    using E = typename F::elem;
    using TE = typename F::telem;
    if (DEC ? lh_from < lh_to : lh_from > lh_to) return;
    auto geom = [&](int lh, uint32_t& idx, uint32_t& ti) {
        const uint32_t h = 1u << lh, i = tid & (h - 1);
        idx = ((tid >> lh) << (lh + 1)) + i; ti = e - 2 * h + i;
    };
    uint32_t idx, ti;
    geom(lh_from, idx, ti);
    TE na = ldt(ta, ti), nb = ldt(tb, ti);
#pragma unroll 1
    for (int lh = lh_from; DEC ? lh >= lh_to : lh <= lh_to; lh += DEC ? -1 : 1) {
        const TE t0 = na, t1 = nb;
        const uint32_t lo = idx, up = idx + (1u << lh);
        const E a = tile[lo], b = tile[up];
        if (DEC) { const E q1 = F::tmul(t1, F::sub(b, a)); tile[lo] = F::tmul_add(t0, q1, a); tile[up] = q1; }
        else { const E o0 = F::tmul_add(t0, b, a), o1 = F::tmul_add(t1, b, a); tile[lo] = o0; tile[up] = o1; }
        if (lh != lh_to) { geom(lh + (DEC ? -1 : 1), idx, ti); na = ldt(ta, ti); nb = ldt(tb, ti); }
        const int nxt = lh != lh_to ? lh + (DEC ? -1 : 1) : lh_after;
        if (wave_local_lh(lh) && wave_local_lh(nxt)) wave_local_sync(); else lds_barrier();
    }
}

// Round 6, second form of the wave-local sweeps: the sweeps at pair distance 64 and 32 of a one-pair-per-thread
// row tile with the PAIR KEPT IN REGISTERS between them.  After the sweep at distance 64 lane l of a wave holds elements (l, l + 64)
// of the wave's 128-element span; the sweep at distance 32 wants (l, l + 32) in lanes < 32 and (l + 32, l + 64) in lanes >= 32: the
// upper half-wave's low elements trade places with the lower half-wave's high elements — one v_permlane32_swap per register (8 per
// pair); from distance 32 to 16 the same inside each half-wave: v_permlane16_swap.  No LDS write + read between the three sweeps.
// ECFFT_ROW_REGS: 0 = through LDS (row_stages_pipe), 1 = registers.  A/B: profiles/r06/row_regs_ab.txt.
#ifndef ECFFT_ROW_REGS
#define ECFFT_ROW_REGS 0
#endif
template <class E>
__device__ __forceinline__ void pair_swap32(E& a, E& b) {
#pragma unroll
    for (int w = 0; w < 8; ++w) { auto p = __builtin_amdgcn_permlane32_swap(a.l[w], b.l[w], false, false); a.l[w] = p[0]; b.l[w] = p[1]; }
}
template <class E>
__device__ __forceinline__ void pair_swap16(E& a, E& b) {
#pragma unroll
    for (int w = 0; w < 8; ++w) { auto p = __builtin_amdgcn_permlane16_swap(a.l[w], b.l[w], false, false); a.l[w] = p[0]; b.l[w] = p[1]; }
}
// decompose at distance 64, then 32: reads the pair (idx6, idx6 + 64) from the tile (published by a barrier), returns the pair
// (idx4, idx4 + 16), idx4 = ((tid >> 4) << 5) + (tid & 15), of the state after the distance-32 sweep — what the fused distance-16 sweep reads
template <class F>
__device__ __forceinline__ void row_regs_dec(const typename F::elem* tile, const typename F::telem* __restrict__ np0, const typename F::telem* __restrict__ dinv,
                                             uint32_t e, uint32_t tid, typename F::elem& a, typename F::elem& b) {
    const uint32_t l = tid & 63u, idx6 = ((tid >> 6) << 7) + l;
    typename F::telem t0 = ldt(np0, e - 128u + l), t1 = ldt(dinv, e - 128u + l);
    a = tile[idx6]; b = tile[idx6 + 64];
    bfly<F, true>(a, b, t0, t1);
    t0 = ldt(np0, e - 64u + (l & 31u)); t1 = ldt(dinv, e - 64u + (l & 31u));
    pair_swap32(a, b);
    bfly<F, true>(a, b, t0, t1);
    pair_swap16(a, b);
}
// recombine at distance 32, then 64, from the pair (idx4, idx4 + 16) after the distance-16 sweep; writes the pair (idx6, idx6 + 64) to the tile
template <class F>
__device__ __forceinline__ void row_regs_rec(typename F::elem* tile, const typename F::telem* __restrict__ p0, const typename F::telem* __restrict__ p1,
                                             uint32_t e, uint32_t tid, typename F::elem a, typename F::elem b) {
    const uint32_t l = tid & 63u, idx6 = ((tid >> 6) << 7) + l;
    typename F::telem t0 = ldt(p0, e - 64u + (l & 31u)), t1 = ldt(p1, e - 64u + (l & 31u));
    pair_swap16(a, b);
    bfly<F, false>(a, b, t0, t1);
    t0 = ldt(p0, e - 128u + l); t1 = ldt(p1, e - 128u + l);
    pair_swap32(a, b);
    bfly<F, false>(a, b, t0, t1);
    tile[idx6] = a; tile[idx6 + 64] = b;
}

template <class F, int LOG_TILE_CT>      // LOG_TILE_CT > 0: tile size known at compile time (loops unroll); 0: runtime log_tile
__global__ __launch_bounds__(kBlockRow, ECFFT_MIN_WAVES) void k_stages_lds(IoDesc<F> io,
                                                           const typename F::telem* __restrict__ np0,
                                                           const typename F::telem* __restrict__ dinv,
                                                           const typename F::telem* __restrict__ p0,
                                                           const typename F::telem* __restrict__ p1,
                                                           const typename F::telem* __restrict__ inner,
                                                           uint32_t log_e, uint32_t k_first, uint32_t log_tile,
                                                           const typename F::telem* __restrict__ c0t,
                                                           const uint8_t* __restrict__ blkA, const unsigned long long* __restrict__ blkK) {
    // blkA / blkK != nullptr (32-byte field, tiles of whole 1024-element sub-tiles, >= 4 in-tile stages): the stages with pair
    // distance <= 8 run on the int8 matrix cores as one 16 x 16 map per block (mfma_blk16.h) instead of 7 VALU sweeps
    using E = typename F::elem;
    extern __shared__ __attribute__((aligned(16))) unsigned char ecfft_smem[];
    E* tile = reinterpret_cast<E*>(ecfft_smem);
    if (LOG_TILE_CT > 0) log_tile = LOG_TILE_CT;
    const uint32_t T = 1u << log_tile, tid = threadIdx.x;
    const size_t base = (size_t)blockIdx.x << log_tile;
    const size_t e = (size_t)1 << log_e, emask = e - 1;
    constexpr bool kFast = sizeof(E) == 4 && LOG_TILE_CT > 0 && ((1u << (LOG_TILE_CT > 0 ? LOG_TILE_CT : 0)) % (kBlockRow * 16)) == 0;
    constexpr int kNQ = kFast ? (1 << (LOG_TILE_CT > 0 ? LOG_TILE_CT : 2)) / (4 * kBlockRow) : 1;
    bool vio = false;
    if constexpr (kFast) vio = vio_ok<F>(io, log_e);
    if constexpr (kFast) {
        if (vio) {
#ifndef ECFFT_EXP_NO_TILE_IO      // experiment (tools/experiments/README): compute only — the bound of what perfect tile prefetch could reach
#pragma unroll 1
            for (uint32_t b = 0; b < (uint32_t)kNQ / 4; ++b) vio_load<F, 4, kBlockRow>(io, emask, tile, [=](uint32_t j) { return base + j; }, tid + b * 4u * kBlockRow);
#endif
        }
    }
    if (!vio) {
#ifndef ECFFT_EXP_NO_TILE_IO
#pragma unroll
        for (uint32_t j = tid; j < T; j += kBlockRow) tile[j] = io_load<F>(io, base + j, emask);
#endif
    }
    __syncthreads();
    const uint32_t npairs = T >> 1;
    if constexpr (kFast) {
        // register-resident multi-stage engine, 8192 elements per call (a 64 KiB ST_ENTER tile is two independent halves)
#pragma unroll 1
        for (uint32_t o = 0; o < T; o += kBlockRow * 16)
            lds_extend_fast<F, 16, kBlockRow>(tile + o, np0, dinv, p0, p1, inner, e, log_e - k_first, tid);
    } else {
    // stages k_first .. log_e-2 (h >= 2); the two innermost stages (decompose h=1, recombine h=1) act on the
    // same pairs back to back and are merged into out_j = a + c_j*(b - a): 2 multiplies instead of 4
    const uint32_t k_inner = log_e ? log_e - 1 : 0;
    // the matrix-core form lives in the compile-time-tile instantiation only (1024-element tiles): its four 32 x 32 accumulators
    // next to the multiply's registers spill a few dwords in the run-time-tile instantiation, which the small tiles keep using
    constexpr bool kMfma = sizeof(E) == 32 && kBlockRow == 512 && LOG_TILE_CT == 10;
    bool mfma = false;
    if constexpr (kMfma) mfma = blkA != nullptr;
    const uint32_t k_dec_end = mfma ? log_e - 4 : k_inner;              // mfma: VALU sweeps only for pair distances >= 16
    // mfma with T == 1024 and a decompose sweep at distance 16 in this kernel: that sweep writes its results in operand form itself
    const bool fuse16 = mfma && T == (uint32_t)Blk16::kSub && k_first < k_dec_end;
    bool pipe = false;                                                  // one pair per thread: constants one sweep ahead
    if constexpr (sizeof(E) == 32 && ECFFT_ROW_PIPE) pipe = npairs == (uint32_t)kBlockRow && (e >> 31) == 0;
    // registers across the sweeps at distance 64 / 32 / 16 (ECFFT_ROW_REGS): the pipelined sweeps stop above distance 64
    const bool regs = ECFFT_ROW_REGS && sizeof(E) == 32 && pipe && fuse16 && log_e - k_first >= 7;
    if (pipe) {
        if constexpr (sizeof(E) == 32) row_stages_pipe<F, true>(tile, np0, dinv, (uint32_t)e, (int)(log_e - k_first) - 1, regs ? 7 : (int)(log_e - (k_dec_end - (fuse16 ? 1u : 0u))), tid, fuse16 && !regs ? 4 : 99);
    } else
    for (uint32_t k = k_first; k < k_dec_end - (fuse16 ? 1u : 0u); ++k) {
        const uint32_t lh = log_e - k - 1, h = 1u << lh;
        stage_sweep<F, true, kBlockRow>(tile, np0 + (e - 2 * (size_t)h), dinv + (e - 2 * (size_t)h), lh, npairs, tid, c0t ? c0t + (e - 2 * (size_t)h) : nullptr);
        __syncthreads();
    }
    if constexpr (kMfma) {
        if (mfma) {
            Blk16::APre pre;
            if (fuse16) {                                                // one pair per thread: (idx, idx + 16)
                const uint32_t i = tid & 15u, idx = ((tid >> 4) << 5) + i;
                E a, b;
                if constexpr (ECFFT_ROW_REGS && sizeof(E) == 32) { if (regs) row_regs_dec<F>(tile, np0, dinv, (uint32_t)e, tid, a, b); else { a = tile[idx]; b = tile[idx + 16]; } }
                else { a = tile[idx]; b = tile[idx + 16]; }
                const E q1 = F::tmul(ldt(dinv + (e - 32), i), F::sub(b, a));
                const E q0 = F::tmul_add(ldt(np0 + (e - 32), i), q1, a);
                __builtin_amdgcn_sched_barrier(0);
                pre = Blk16::prefetch(blkA, tid);                        // after the multiplies (they need the registers), before the barriers
                __builtin_amdgcn_sched_barrier(0);
                // operand form permutes chunks across threads' elements — inside 8-element groups (Blk16::phys), i.e. inside the wave's span
                if (ECFFT_WAVE_LOCAL) wave_local_sync(); else __syncthreads();
                Blk16::store_operand(tile, idx, q0); Blk16::store_operand(tile, idx + 16, q1);
                __syncthreads();
            } else {
                pre = Blk16::prefetch(blkA, tid);
                __builtin_amdgcn_sched_barrier(0);
                Blk16::to_operand_form<kBlockRow>(tile, T, tid);
            }
#ifndef ECFFT_EXP_SKIP_PHASE
#pragma unroll 1
            for (uint32_t o = 0; o < T; o += Blk16::kSub) {
                if (o) { pre = Blk16::prefetch(blkA, tid); __builtin_amdgcn_sched_barrier(0); }
                Blk16::phase(tile + o, blkA, blkK, tid, pre);
            }
#endif
            if (fuse16) {                                                // recombine sweep at distance 16, reading the swizzled results
                const uint32_t i = tid & 15u, idx = ((tid >> 4) << 5) + i;
                const E a = Blk16::load_swizzled(tile, idx), b = Blk16::load_swizzled(tile, idx + 16);
                const E o0 = F::tmul_add(ldt(p0 + (e - 32), i), b, a), o1 = F::tmul_add(ldt(p1 + (e - 32), i), b, a);
                // the swizzled positions read above and the plain ones written below lie in the wave's own span; so do the pairs of the
                // recombine sweep at distance 32 that follows in the pipelined form
                bool done_regs = false;
                if constexpr (ECFFT_ROW_REGS && sizeof(E) == 32) {
                    if (regs) {      // distance 32 and 64 in registers; the swizzled positions read above lie in the wave's own span, like the ones written
                        if (ECFFT_WAVE_LOCAL) wave_local_sync(); else __syncthreads();
                        row_regs_rec<F>(tile, p0, p1, (uint32_t)e, tid, o0, o1);
                        lds_barrier();
                        done_regs = true;
                    }
                }
                if (!done_regs) {
                if (ECFFT_WAVE_LOCAL) wave_local_sync(); else __syncthreads();
                tile[idx] = o0; tile[idx + 16] = o1;
                if (ECFFT_WAVE_LOCAL && pipe && k_first + 1 < k_dec_end) wave_local_sync(); else __syncthreads();
                }
            } else {
                Blk16::from_swizzled<kBlockRow>(tile, T, tid);
            }
        }
    }
    if (log_e > 0 && !mfma) {
        if (sizeof(E) == 32 && 2 * npairs <= (uint32_t)kBlockRow) {        // pair-split merged innermost stage
            const bool act = tid < 2 * npairs, hi = tid >= npairs;
            const uint32_t g = hi ? tid - npairs : tid;
            E a, b; typename F::telem c;
            if (act) { a = tile[2 * g]; b = tile[2 * g + 1]; c = inner[hi ? 1 : 0]; }
            __syncthreads();
            if (act) tile[2 * g + (hi ? 1 : 0)] = F::tmul_add(c, F::sub(b, a), a);
        } else {
        const typename F::telem c0 = inner[0], c1 = inner[1];
#pragma unroll
        for (uint32_t g = tid; g < npairs; g += kBlockRow) {
            E a = tile[2 * g], b = tile[2 * g + 1];
            E d = F::sub(b, a);
            tile[2 * g] = F::tmul_add(c0, d, a);
            tile[2 * g + 1] = F::tmul_add(c1, d, a);
        }
        }
        __syncthreads();
    }
    if (pipe) {
        if constexpr (sizeof(E) == 32) row_stages_pipe<F, false>(tile, p0, p1, (uint32_t)e, regs ? 7 : (int)(log_e - (k_dec_end - (fuse16 ? 1u : 0u))), (int)(log_e - k_first) - 1, tid);
    } else
    for (uint32_t k = k_dec_end - (fuse16 ? 1u : 0u); k-- > k_first;) {
        const uint32_t lh = log_e - k - 1, h = 1u << lh;
        stage_sweep<F, false, kBlockRow>(tile, p0 + (e - 2 * (size_t)h), p1 + (e - 2 * (size_t)h), lh, npairs, tid);
        __syncthreads();
    }
    }
    if (io.st_mode == ST_ENTER) {
        // the tile holds whole [U1~ | V1~] blocks (2e <= T): combine them with the level's input and store interleaved
        if constexpr (kFast) {
            auto al = [](const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; };
            if (log_e >= 2 && al(io.aux) && al(io.dst) && al(io.st_a) && al(io.st_b) && al(io.st_c)) {
#ifdef ECFFT_EXP_NO_TILE_IO
                return;
#endif
#pragma unroll 1
                for (int cb = 0; cb < kNQ / 2; cb += 2)
                    vio_enter_store<F, 2>(tile, io.aux, io.dst, io.st_a, io.st_c, io.st_b, e, [=](int c, uint32_t& ju, uint32_t& jv, size_t& i, size_t& bb) {
                        const uint32_t g = 4u * (tid + (uint32_t)c * kBlockRow), ii = g & (uint32_t)emask, lb = (g >> log_e) << (log_e + 1);
                        ju = lb + ii; jv = lb + (uint32_t)e + ii; i = ii; bb = base + lb;
                    }, cb);
                return;
            }
        }
#ifdef ECFFT_EXP_NO_TILE_IO
        return;
#endif
        for (uint32_t g = tid; g < npairs; g += kBlockRow) {
            const uint32_t i = g & (uint32_t)emask, lb = (g >> log_e) << (log_e + 1);
            const size_t bb = base + lb;
            const E u0 = io.aux[bb + i], v0 = io.aux[bb + e + i];
            const E ev = F::tmul_add(io.st_a[i], v0, u0);
            const E od = F::tmul_add(io.st_b[i], tile[lb + (uint32_t)e + i], F::tmul(io.st_c[i], tile[lb + i]));
            io.dst[bb + 2 * i] = F::canon(ev);
            io.dst[bb + 2 * i + 1] = F::canon(od);
        }
        return;
    }
    if constexpr (kFast) {
        if (vio) {
#ifndef ECFFT_EXP_NO_TILE_IO
#pragma unroll 1
            for (uint32_t b = 0; b < (uint32_t)kNQ / 4; ++b) vio_store<F, 4, kBlockRow>(io, log_e, tile, [=](uint32_t j) { return base + j; }, tid + b * 4u * kBlockRow);
#endif
            return;
        }
    }
#ifndef ECFFT_EXP_NO_TILE_IO
#pragma unroll
    for (uint32_t j = tid; j < T; j += kBlockRow) io_store<F>(io, base + j, log_e, tile[j]);
#endif
}

// forward declaration (defined with the low-level kernels below)
template <class F, int BLK, bool SWZ = false>
__device__ __forceinline__ void reg_extend32(typename F::elem* a, uint32_t len, uint32_t log_e, uint32_t k_first,
                                             const typename F::telem* __restrict__ c0t, const typename F::telem* __restrict__ dinv,
                                             const typename F::telem* __restrict__ p0, const typename F::telem* __restrict__ p1,
                                             const typename F::telem* __restrict__ inner, uint32_t tid,
                                             const uint8_t* __restrict__ blkA, const unsigned long long* __restrict__ blkK);

// ---------------------------------------------------------------------------------------------
// Row kernel of the LATENCY regime (32-byte fields, launches with fewer 1024-element tiles than CUs): 256-element
// tiles, 256 threads, ONE element per thread kept in registers through every in-tile stage (reg_extend32: cross-lane moves for
// pair distances < 64, LDS above), and — round 4 — the stages with pair distance <= 8 as one 16-point map per block on the
// matrix cores (v_mfma_i32_16x16x64_i8, Blk16::phase_n16_regs) when blkA != nullptr.  Same stages, same operators and the same
// results as k_stages_lds on a 256-element tile (ST_ENTER excepted: that operator needs whole [U | V] blocks and keeps the
// generic kernel).
// ---------------------------------------------------------------------------------------------
template <class F>
__global__ __launch_bounds__(256, 2) void k_stages_row256(IoDesc<F> io,
                                                          const typename F::telem* __restrict__ dinv,
                                                          const typename F::telem* __restrict__ p0,
                                                          const typename F::telem* __restrict__ p1,
                                                          const typename F::telem* __restrict__ inner,
                                                          uint32_t log_e, uint32_t k_first,
                                                          const typename F::telem* __restrict__ c0t,
                                                          const uint8_t* __restrict__ blkA, const unsigned long long* __restrict__ blkK) {
    using E = typename F::elem;
    if constexpr (sizeof(E) == 32) {
        __shared__ E tile[256];
        const uint32_t tid = threadIdx.x;
        const size_t pos = ((size_t)blockIdx.x << 8) + tid;
        const size_t emask = ((size_t)1 << log_e) - 1;
        tile[tid] = io_load<F>(io, pos, emask);
        __syncthreads();
        reg_extend32<F, 256>(tile, 256u, log_e, k_first, c0t, dinv, p0, p1, inner, tid, blkA, blkK);
        io_store<F>(io, pos, log_e, tile[tid]);
    }
}

#ifndef ECFFT_COL_PAD
#define ECFFT_COL_PAD 0
#endif
// LDS row stride of a column tile: C elements + optional padding.  Unpadded, the small-distance stages make consecutive lane
// groups hit the same half of the 256-byte bank row (rows are 128 B at C = 4), a 4-way conflict on ds_read_b128 — but an
// A/B on MI355X (pad 0 / 1 / 2 rows) showed no difference: the kernel is bound by the modular multiply, not by LDS.
template <class E>
__host__ __device__ constexpr uint32_t col_row_stride(uint32_t C) { return C + (sizeof(E) == 4 ? 4u * ECFFT_COL_PAD : 1u * ECFFT_COL_PAD); }

// column-tile variant of stage_sweep: pair (row r, column cc) with partner d rows below; table entry ((r mod d) << log_hs) + c0 + cc
template <class F, bool DEC>
__device__ __forceinline__ void col_stage_sweep(typename F::elem* tile, const typename F::telem* __restrict__ pa, const typename F::telem* __restrict__ pb,
                                                uint32_t sft, uint32_t log_c, uint32_t log_hs, size_t c0, uint32_t npairs, uint32_t tid,
                                                const typename F::telem* __restrict__ pc = nullptr) {
    using E = typename F::elem;
    const uint32_t C = 1u << log_c, d = 1u << sft;
    if constexpr (sizeof(E) == 32) {
        if (2 * npairs <= (uint32_t)kBlockLds && (!DEC || pc)) {          // pair-split form, see stage_sweep
            const bool act = tid < 2 * npairs, hi = tid >= npairs;
            const uint32_t g = hi ? tid - npairs : tid, cc = g & (C - 1), pr = g >> log_c;
            const uint32_t r = ((pr >> sft) << (sft + 1)) | (pr & (d - 1));
            const size_t i = ((size_t)(r & (d - 1)) << log_hs) + c0 + cc;
            const uint32_t RS = col_row_stride<E>(C), lo = r * RS + cc, up = lo + d * RS;
            E x, y; typename F::telem t;
            if (act) { x = tile[lo]; y = tile[up]; t = DEC ? (hi ? pb[i] : pc[i]) : (hi ? pb[i] : pa[i]); }
            __syncthreads();
            if (act) {
                if (DEC) { const E dd = F::sub(y, x); if (hi) tile[up] = F::tmul(t, dd); else tile[lo] = F::tmul_add(t, dd, x); }
                else tile[hi ? up : lo] = F::tmul_add(t, y, x);
            }
            return;
        }
    }
    if constexpr (sizeof(E) == 4) {
        if (log_c >= 2) {
            for (uint32_t g4 = tid; g4 < (npairs >> 2); g4 += kBlockLds) {
                const uint32_t g = g4 << 2, cc = g & (C - 1), pr = g >> log_c;
                const uint32_t r = ((pr >> sft) << (sft + 1)) | (pr & (d - 1));
                const size_t i = ((size_t)(r & (d - 1)) << log_hs) + c0 + cc;
                const uint32_t RS = col_row_stride<E>(C), lo = r * RS + cc, hi = lo + d * RS;
                uint4 va = *reinterpret_cast<const uint4*>(tile + lo), vb = *reinterpret_cast<const uint4*>(tile + hi);
                const uint4 v0 = *reinterpret_cast<const uint4*>(pa + i), v1 = *reinterpret_cast<const uint4*>(pb + i);
                uint32_t xa[4] = {va.x, va.y, va.z, va.w}, xb[4] = {vb.x, vb.y, vb.z, vb.w};
                const uint32_t t0[4] = {v0.x, v0.y, v0.z, v0.w}, t1[4] = {v1.x, v1.y, v1.z, v1.w};
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    if (DEC) { E q1 = F::tmul(t1[c], F::sub(xb[c], xa[c])); xa[c] = F::tmul_add(t0[c], q1, xa[c]); xb[c] = q1; }
                    else { E o0 = F::tmul_add(t0[c], xb[c], xa[c]), o1 = F::tmul_add(t1[c], xb[c], xa[c]); xa[c] = o0; xb[c] = o1; }
                }
                *reinterpret_cast<uint4*>(tile + lo) = make_uint4(xa[0], xa[1], xa[2], xa[3]);
                *reinterpret_cast<uint4*>(tile + hi) = make_uint4(xb[0], xb[1], xb[2], xb[3]);
            }
            return;
        }
    }
    for (uint32_t g = tid; g < npairs; g += kBlockLds) {
        const uint32_t cc = g & (C - 1), pr = g >> log_c;
        const uint32_t r = ((pr >> sft) << (sft + 1)) | (pr & (d - 1));
        const size_t i = ((size_t)(r & (d - 1)) << log_hs) + c0 + cc;
        const uint32_t RS = col_row_stride<E>(C), lo = r * RS + cc, hi = lo + d * RS;
        E a = tile[lo], b = tile[hi];
        if (DEC) { E q1 = F::tmul(pb[i], F::sub(b, a)); tile[lo] = F::tmul_add(pa[i], q1, a); tile[hi] = q1; }
        else { tile[lo] = F::tmul_add(pa[i], b, a); tile[hi] = F::tmul_add(pb[i], b, a); }
    }
}

#ifndef ECFFT_COL_PIPE
#define ECFFT_COL_PIPE 1
#endif
// Column stages of a 32-byte field with the table constants ONE SWEEP AHEAD (round 3).  A column pass of a single transform is one
// workgroup per CU, and every sweep used to be "request two 64-byte constants from HBM / L2 -> wait -> two multiplies ->
// barrier": the top sweeps' constants are read once per launch, so their latency was paid R times in a row (48 % VALU-busy solo,
// profiles/r03).  Here the constants of sweep s+1 (they depend only on the thread's index) are requested right after the
// multiplies of sweep s, when the registers are free again, and fly across the LDS-only barrier and the next sweep's LDS reads.
// One pair per thread is pipelined (pair index tid); further pairs of the thread (two vectors per workgroup) load theirs in
// place as before.  Same arithmetic, same order: bit-identical.  Ends with a barrier.
template <class F, bool DEC, int BLK>
__device__ __forceinline__ void col_stages_pipe(typename F::elem* tile, const typename F::telem* __restrict__ ta, const typename F::telem* __restrict__ tb,
                                                uint32_t R, uint32_t log_c, uint32_t log_hs, size_t c0, size_t e, uint32_t tid, uint32_t npairs) {
    using E = typename F::elem;
    using TE = typename F::telem;
    const uint32_t C = 1u << log_c, RS = col_row_stride<E>(C);
    auto geom = [&](uint32_t st, uint32_t g, uint32_t& lo, uint32_t& up, uint32_t& ti) {
        const uint32_t sft = DEC ? R - 1 - st : st, d = 1u << sft;
        const uint32_t cc = g & (C - 1), pr = g >> log_c;
        const uint32_t r = ((pr >> sft) << (sft + 1)) | (pr & (d - 1));
        ti = (uint32_t)(e - 2 * (((size_t)1 << log_hs) << sft)) + ((r & (d - 1)) << log_hs) + (uint32_t)c0 + cc;
        lo = r * RS + cc; up = lo + d * RS;
    };
    uint32_t lo, up, ti;
    geom(0, tid, lo, up, ti);
    TE na = ldt(ta, ti), nb = ldt(tb, ti);
#pragma unroll 1
    for (uint32_t st = 0; st < R; ++st) {
        const TE t0 = na, t1 = nb;
        const uint32_t lo0 = lo, up0 = up;
        {
            const E a = tile[lo0], b = tile[up0];
            if (DEC) { const E q1 = F::tmul(t1, F::sub(b, a)); tile[lo0] = F::tmul_add(t0, q1, a); tile[up0] = q1; }
            else { const E o0 = F::tmul_add(t0, b, a), o1 = F::tmul_add(t1, b, a); tile[lo0] = o0; tile[up0] = o1; }
        }
        for (uint32_t g = tid + BLK; g < npairs; g += BLK) {
            uint32_t l2, u2, i2; geom(st, g, l2, u2, i2);
            const E a = tile[l2], b = tile[u2];
            if (DEC) { const E q1 = F::tmul(ldt(tb, i2), F::sub(b, a)); tile[l2] = F::tmul_add(ldt(ta, i2), q1, a); tile[u2] = q1; }
            else { const E o0 = F::tmul_add(ldt(ta, i2), b, a), o1 = F::tmul_add(ldt(tb, i2), b, a); tile[l2] = o0; tile[u2] = o1; }
        }
        if (st + 1 < R) { geom(st + 1, tid, lo, up, ti); na = ldt(ta, ti); nb = ldt(tb, ti); }
        // pairs g in [64 q, 64 q + 64) of a stage with element distance (C << sft) <= 64 fill one aligned span of 128 elements, the same
        // span for every such stage: between two of them only the wave's own LDS traffic has to be ordered (wave_local_sync)
        const uint32_t sft = DEC ? R - 1 - st : st, sft_n = DEC ? sft - 1 : sft + 1;
        if (st + 1 < R && wave_local_lh((int)(sft + log_c)) && wave_local_lh((int)(sft_n + log_c))) wave_local_sync(); else lds_barrier();
    }
}

// The R stages of a column tile (2^R rows x C columns in LDS, row distance 2^s at stage k = kb - s; table entry
// ((row mod 2^s) << log_hs) + c0 + column, table base e - 2*(hs << s)).  DECOMPOSE runs s = R-1 .. 0, recombine 0 .. R-1.
// 4-byte fields with exactly 16 elements per thread use the register-resident radix steps (up to 3 stages per LDS round
// trip, table constants requested before the barrier); everything else sweeps one stage at a time.  Ends with a barrier.
template <class F, bool DEC>
__device__ __forceinline__ void col_stages(typename F::elem* tile, const typename F::telem* __restrict__ ta, const typename F::telem* __restrict__ tb,
                                           uint32_t R, uint32_t log_c, uint32_t log_hs, size_t c0, size_t e, uint32_t tid, uint32_t halves = 1,
                                           const typename F::telem* __restrict__ tc = nullptr) {
    // halves = 2: two such tiles back to back in LDS that use the same table entries (k_stages_col_enter)
    using E = typename F::elem;
    const uint32_t C = 1u << log_c, T = C << R;
    if constexpr (sizeof(E) == 4 && ECFFT_COL_PAD == 0) {
        if (T == kBlockLds * 16) {
          {
            uint32_t done = 0;
            while (done < R) {
                const uint32_t ns = R - done >= 3 ? 3 : R - done;
                const uint32_t slo = DEC ? R - done - ns : done;              // lowest stage index s of this chunk
                auto run = [&](auto NSc) {
                    constexpr int NS = decltype(NSc)::value;
                    constexpr int NG = 16 >> NS;
                    uint32_t pb[NG], tb0[NG];
#pragma unroll
                    for (int g = 0; g < NG; ++g) {
                        const uint32_t q = tid + (uint32_t)kBlockLds * g, cc = q & (C - 1), rq = q >> log_c;
                        const uint32_t rlow = rq & ((1u << slo) - 1), rb = ((rq >> slo) << (slo + NS)) | rlow;
                        pb[g] = (rb << log_c) + cc; tb0[g] = (rlow << log_hs) + (uint32_t)c0 + cc;
                    }
                    radix_step<F, NS, DEC, NG>(tile, ta, tb, (uint32_t)e, pb, C << slo, tb0, (1u << log_hs) << slo, halves, T);
                };
                if (ns == 3) run(std::integral_constant<int, 3>{}); else if (ns == 2) run(std::integral_constant<int, 2>{}); else run(std::integral_constant<int, 1>{});
                done += ns;
            }
          }
            __syncthreads();
            return;
        }
    }
    const size_t hs = (size_t)1 << log_hs;
    if constexpr (sizeof(E) == 32 && ECFFT_COL_PIPE) {
        const uint32_t npairs = (halves * T) >> 1;
        if (npairs >= (uint32_t)kBlockLds && npairs % kBlockLds == 0 && (e >> 32) == 0) {     // at least one pair per thread (not the pair-split small tiles)
            col_stages_pipe<F, DEC, kBlockLds>(tile, ta, tb, R, log_c, log_hs, c0, e, tid, npairs);
            return;
        }
    }
    for (uint32_t st = 0; st < R; ++st) {
        const uint32_t sft = DEC ? R - 1 - st : st;
        const size_t h = hs << sft;
        col_stage_sweep<F, DEC>(tile, ta + (e - 2 * h), tb + (e - 2 * h), sft, log_c, log_hs, c0, (halves * T) >> 1, tid, tc ? tc + (e - 2 * h) : nullptr);   // pairs never cross a tile (row distance < 2^R)
        __syncthreads();
    }
}

// ---------------------------------------------------------------------------------------------
// LDS-fused butterfly stages, "column kernel": the R = kb-ka+1 consecutive stages ka..kb whose pair
// distances (h_ka = hs*2^(R-1) ... h_kb = hs) are too large for a contiguous tile.  A workgroup
// gathers 2^R rows spaced hs apart, 2^log_c contiguous elements each (>= 4 KiB per row for R <= 4 on
// secp256k1: fully coalesced), runs the R stages in LDS and scatters the rows back.  DECOMPOSE runs
// ka -> kb (large distance first), RECOMBINE kb -> ka.  Table index of the pair (row r, column c) at
// stage k: (r mod d)*hs + c_global with d = 2^(kb-k).
// ---------------------------------------------------------------------------------------------
template <class F, bool DECOMPOSE, int LOG_TILE_CT>     // LOG_TILE_CT > 0: log2(tile elements) known at compile time
__global__ __launch_bounds__(kBlockLds, ECFFT_MIN_WAVES) void k_stages_col(IoDesc<F> io,
                                                           const typename F::telem* __restrict__ ta,   // np0 | p0
                                                           const typename F::telem* __restrict__ tb,   // dinv | p1
                                                           uint32_t log_e, uint32_t ka, uint32_t kb, uint32_t log_c,
                                                           const typename F::telem* __restrict__ tc,    // np0*dinv (pair-split decompose) | unused
                                                           uint32_t log_v) {  // 4-byte fast path only: 2^log_v consecutive spans per workgroup
    // log_v = 1: two consecutive 2h_ka-spans (same column chunk) share one workgroup — every span of a stage reads the SAME table
    // entries, so their constants are loaded once for both (halves the table traffic of the HBM-bound column passes)
    using E = typename F::elem;
    extern __shared__ __attribute__((aligned(16))) unsigned char ecfft_smem[];
    E* tile = reinterpret_cast<E*>(ecfft_smem);
    const uint32_t R = kb - ka + 1, tid = threadIdx.x;
    const uint32_t C = 1u << log_c, T = LOG_TILE_CT > 0 ? (1u << LOG_TILE_CT) : (C << R);
    const size_t e = (size_t)1 << log_e, emask = e - 1;
    const uint32_t log_hs = log_e - kb - 1;
    const size_t hs = (size_t)1 << log_hs;
    const uint32_t chunks_log = log_hs - log_c;                      // column chunks per 2h_ka block
    const size_t blk = ((size_t)blockIdx.x >> chunks_log) << log_v, chunk = (size_t)blockIdx.x & (((size_t)1 << chunks_log) - 1);
    const size_t B = (blk << (log_hs + R)) + (chunk << log_c);       // position of (row 0, col 0)
    const size_t c0 = (chunk << log_c);                              // column offset inside the hs-block
    constexpr bool kFast = sizeof(E) == 4 && ECFFT_COL_PAD == 0 && LOG_TILE_CT > 0 && (1u << (LOG_TILE_CT > 0 ? LOG_TILE_CT : 0)) == kBlockLds * 16;
    bool vio = false;
    if constexpr (kFast) vio = log_c >= 2 && vio_ok<F>(io, log_e);
    const uint32_t log_T = log_c + R;
    auto pos_of = [=](uint32_t j) { const uint32_t jj = j & (T - 1); return B + ((size_t)(j >> log_T) << (log_hs + R)) + ((size_t)(jj >> log_c) << log_hs) + (jj & (C - 1)); };
    if constexpr (kFast) {
        if (log_v && !vio) return;                                   // host only pairs spans on the vector path (never taken)
        if (vio) {
            // paired spans: two batches of four quads per thread (one batch of eight needs > 128 VGPRs and spills)
#ifndef ECFFT_EXP_NO_TILE_IO
#pragma unroll 1
            for (uint32_t b = 0; b < (1u << log_v); ++b) vio_load<F, 4, kBlockLds>(io, emask, tile, pos_of, tid + b * 4u * kBlockLds);
#endif
            __syncthreads();
            col_stages<F, DECOMPOSE>(tile, ta, tb, R, log_c, log_hs, c0, e, tid, 1u << log_v, nullptr);
#ifndef ECFFT_EXP_NO_TILE_IO
#pragma unroll 1
            for (uint32_t b = 0; b < (1u << log_v); ++b) vio_store<F, 4, kBlockLds>(io, log_e, tile, pos_of, tid + b * 4u * kBlockLds);
#endif
            return;
        }
    }
    if (!vio) {
#ifndef ECFFT_EXP_NO_TILE_IO
#pragma unroll
        for (uint32_t j = tid; j < T; j += kBlockLds) {
            uint32_t r = j >> log_c, cc = j & (C - 1);
            tile[r * col_row_stride<E>(C) + cc] = io_load<F>(io, B + ((size_t)r << log_hs) + cc, emask);
        }
#endif
    }
    __syncthreads();
    col_stages<F, DECOMPOSE>(tile, ta, tb, R, log_c, log_hs, c0, e, tid, 1, DECOMPOSE ? tc : nullptr);
#ifndef ECFFT_EXP_NO_TILE_IO
#pragma unroll
    for (uint32_t j = tid; j < T; j += kBlockLds) {
        uint32_t r = j >> log_c, cc = j & (C - 1);
        io_store<F>(io, B + ((size_t)r << log_hs) + cc, log_e, tile[r * col_row_stride<E>(C) + cc]);
    }
#endif
}

// ---------------------------------------------------------------------------------------------
// Two consecutive EXTEND cores of an EXIT level meet at a column pass: core A ends with the recombine stages kb..ka,
// core B (opposite direction, SAME point set: A's target parity = B's source parity) starts with the decompose stages
// ka..kb on the same tiles, with only a pointwise step in between.  This kernel runs both halves on one tile residency:
// load, R recombine stages, io_mid (A's store operator + B's load operator), R decompose stages, store — one launch and
// one HBM round trip instead of two.
// ---------------------------------------------------------------------------------------------
template <class F>
__global__ __launch_bounds__(kBlockLds, ECFFT_MIN_WAVES) void k_stages_col_mid(IoDesc<F> io,
                                                               const typename F::telem* __restrict__ p0, const typename F::telem* __restrict__ p1,
                                                               const typename F::telem* __restrict__ np0, const typename F::telem* __restrict__ dinv,
                                                               uint32_t log_e, uint32_t ka, uint32_t kb, uint32_t log_c,
                                                               const typename F::telem* __restrict__ c0t, uint32_t log_v) {
    using E = typename F::elem;
    extern __shared__ __attribute__((aligned(16))) unsigned char ecfft_smem[];
    E* tile = reinterpret_cast<E*>(ecfft_smem);
    const uint32_t R = kb - ka + 1, tid = threadIdx.x;
    const uint32_t C = 1u << log_c, T = C << R;
    const size_t e = (size_t)1 << log_e, emask = e - 1;
    const uint32_t log_hs = log_e - kb - 1;
    const size_t hs = (size_t)1 << log_hs;
    const uint32_t chunks_log = log_hs - log_c;
    const size_t blk = ((size_t)blockIdx.x >> chunks_log) << log_v, chunk = (size_t)blockIdx.x & (((size_t)1 << chunks_log) - 1);
    const size_t B = (blk << (log_hs + R)) + (chunk << log_c);
    const size_t c0 = (chunk << log_c);
    const uint32_t log_T = log_c + R;
    auto pos_of = [=](uint32_t j) { const uint32_t jj = j & (T - 1); return B + ((size_t)(j >> log_T) << (log_hs + R)) + ((size_t)(jj >> log_c) << log_hs) + (jj & (C - 1)); };
    bool vio = false;
    if constexpr (sizeof(E) == 4 && ECFFT_COL_PAD == 0) {
        IoDesc<F> chk = io; chk.src_stride = 1; chk.src_off = 0;           // this kernel reads src plainly and stores plainly
        const int m = io.st_mode;
        vio = T == kBlockLds * 16 && log_c >= 2 && (m == ST_PLAIN || m == ST_SCALE || m == ST_AXPBY) && vio_ok<F>(chk, log_e);
        if (log_v && !vio) return;                                   // host only pairs spans on the vector path (never taken)
        if (vio) {
            IoDesc<F> pl = io; pl.src_stride = 1; pl.src_off = 0; pl.ld_mode = LD_PLAIN; pl.st_mode = ST_PLAIN;
#pragma unroll 1
            for (uint32_t b = 0; b < (1u << log_v); ++b) vio_load<F, 4, kBlockLds>(pl, emask, tile, pos_of, tid + b * 4u * kBlockLds);
            __syncthreads();
            col_stages<F, false>(tile, p0, p1, R, log_c, log_hs, c0, e, tid, 1u << log_v);
#pragma unroll 1
            for (uint32_t b = 0; b < (2u << log_v); ++b) vio_mid<F, 2, kBlockLds>(io, log_e, tile, pos_of, tid + b * 2u * kBlockLds);   // 2 quads at a time: 4 spill
            __syncthreads();
            col_stages<F, true>(tile, np0, dinv, R, log_c, log_hs, c0, e, tid, 1u << log_v, nullptr);
#pragma unroll 1
            for (uint32_t b = 0; b < (1u << log_v); ++b) vio_store<F, 4, kBlockLds>(pl, log_e, tile, pos_of, tid + b * 4u * kBlockLds);
            return;
        }
    }
#ifndef ECFFT_EXP_NO_TILE_IO
    for (uint32_t j = tid; j < T; j += kBlockLds) {
        uint32_t r = j >> log_c, cc = j & (C - 1);
        tile[r * col_row_stride<E>(C) + cc] = io.src[B + ((size_t)r << log_hs) + cc];
    }
#endif
    __syncthreads();
    col_stages<F, false>(tile, p0, p1, R, log_c, log_hs, c0, e, tid);
#ifndef ECFFT_EXP_NO_MID
    for (uint32_t j = tid; j < T; j += kBlockLds) {
        uint32_t r = j >> log_c, cc = j & (C - 1);
        { const uint32_t q = r * col_row_stride<E>(C) + cc; tile[q] = io_mid<F>(io, B + ((size_t)r << log_hs) + cc, emask, tile[q]); }
    }
#endif
    __syncthreads();
    col_stages<F, true>(tile, np0, dinv, R, log_c, log_hs, c0, e, tid, 1, c0t);
#ifndef ECFFT_EXP_NO_TILE_IO
    for (uint32_t j = tid; j < T; j += kBlockLds) {
        uint32_t r = j >> log_c, cc = j & (C - 1);
        data_st(&io.dst[B + ((size_t)r << log_hs) + cc], F::canon(tile[r * col_row_stride<E>(C) + cc]), io.nt_st);
    }
#endif
}

// ---------------------------------------------------------------------------------------------
// The LAST column pass of an ENTER level's EXTEND (recombine stages kb..0) fused with the level's combine step
// (src/fftree.rs:155-159).  The combine needs U1~[i] and V1~[i] — the same column of two ADJACENT vectors of the batched
// EXTEND — so one workgroup takes both vectors' column tiles: rows 0..2^R-1 = vector 2b (U), rows 2^R..2^(R+1)-1 = vector
// 2b+1 (V), R = kb+1, C columns each (64 KiB of LDS).  The stage sweeps never pair across the two halves (row distance
// <= 2^(R-1)) and both halves read the same table entries.  Store: even = u0 + xe*v0, odd = w1*U1~ + w1x*V1~, interleaved,
// straight to the level's output — the separate combine launch (3n element moves for 1.5n multiplies, HBM-bound) and the
// EXTEND's write + re-read of its result are gone.
// ---------------------------------------------------------------------------------------------
template <class F>
__global__ __launch_bounds__(kBlockLds, ECFFT_MIN_WAVES) void k_stages_col_enter(const typename F::elem* __restrict__ work, const typename F::elem* __restrict__ src,
                                                                 typename F::elem* __restrict__ dst,
                                                                 const typename F::telem* __restrict__ p0, const typename F::telem* __restrict__ p1,
                                                                 const typename F::telem* __restrict__ xe, const typename F::telem* __restrict__ w1,
                                                                 const typename F::telem* __restrict__ w1x,
                                                                 uint32_t log_e, uint32_t kb, uint32_t log_c) {
    using E = typename F::elem;
    extern __shared__ __attribute__((aligned(16))) unsigned char ecfft_smem[];
    E* tile = reinterpret_cast<E*>(ecfft_smem);
    const uint32_t R = kb + 1, tid = threadIdx.x;
    const uint32_t C = 1u << log_c, T = C << (R + 1), RS = col_row_stride<E>(C);
    const size_t e = (size_t)1 << log_e;
    const uint32_t log_hs = log_e - R;                               // hs = e >> R: row spacing; 2^R rows span one vector
    const size_t hs = (size_t)1 << log_hs;
    const uint32_t chunks_log = log_hs - log_c;
    const size_t b = (size_t)blockIdx.x >> chunks_log, chunk = (size_t)blockIdx.x & (((size_t)1 << chunks_log) - 1);
    const size_t c0 = chunk << log_c;
    const size_t B = (b << (log_e + 1)) + c0;                        // position of (row 0, col 0) of the U vector
    bool vio = false;
    if constexpr (sizeof(E) == 4 && ECFFT_COL_PAD == 0) {
        auto al = [](const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; };
        vio = T == kBlockLds * 32 && log_c >= 2 && al(work) && al(src) && al(dst) && al(xe) && al(w1) && al(w1x);
        if (vio) {
#ifndef ECFFT_EXP_NO_TILE_IO
            Quad d[8];
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                const uint32_t j = 4u * (tid + (uint32_t)c * kBlockLds), r = j >> log_c, cc = j & (C - 1);
                d[c] = ldq(work + ECFFT_DPOS(B + ((size_t)(r >> R) << log_e) + ((size_t)(r & ((1u << R) - 1)) << log_hs) + cc));
            }
#pragma unroll
            for (int c = 0; c < 8; ++c) stq(tile + 4u * (tid + (uint32_t)c * kBlockLds), d[c]);
#endif
            __syncthreads();
            col_stages<F, false>(tile, p0, p1, R, log_c, log_hs, c0, e, tid, 2);
#ifdef ECFFT_EXP_NO_TILE_IO
            return;
#endif
            const size_t bb0 = b << (log_e + 1);
#pragma unroll 1
            for (int cb = 0; cb < 4; cb += 2)
                vio_enter_store<F, 2>(tile, src, dst, xe, w1, w1x, e, [=](int c, uint32_t& ju, uint32_t& jv, size_t& i, size_t& bb) {
                    const uint32_t j = 4u * (tid + (uint32_t)c * kBlockLds), r = j >> log_c, cc = j & (C - 1);
                    ju = j; jv = j + (C << R); i = ((size_t)r << log_hs) + c0 + cc; bb = bb0;
                }, cb);
            return;
        }
    }
#ifndef ECFFT_EXP_NO_TILE_IO
#pragma unroll
    for (uint32_t j = tid; j < T; j += kBlockLds) {
        const uint32_t r = j >> log_c, cc = j & (C - 1);            // r < 2^(R+1): r >> R selects the vector
        tile[r * RS + cc] = work[B + ((size_t)(r >> R) << log_e) + ((size_t)(r & ((1u << R) - 1)) << log_hs) + cc];
    }
#endif
    __syncthreads();
    // the U rows and the V rows are two independent column tiles that read the same table entries
    col_stages<F, false>(tile, p0, p1, R, log_c, log_hs, c0, e, tid, 2);
#ifdef ECFFT_EXP_NO_TILE_IO
    return;
#endif
#pragma unroll
    for (uint32_t j = tid; j < (T >> 1); j += kBlockLds) {
        const uint32_t r = j >> log_c, cc = j & (C - 1);
        const size_t i = ((size_t)r << log_hs) + c0 + cc, bb = b << (log_e + 1);
        const E u0 = src[bb + i], v0 = src[bb + e + i];
        const E ev = F::tmul_add(xe[i], v0, u0);
        const E od = F::tmul_add(w1x[i], tile[((1u << R) + r) * RS + cc], F::tmul(w1[i], tile[r * RS + cc]));
        dst[bb + 2 * i] = F::canon(ev);
        dst[bb + 2 * i + 1] = F::canon(od);
    }
}

// ---------------------------------------------------------------------------------------------
// Whole low levels of ENTER / EXIT in LDS.  Level m <= tile only touches data inside one tile-aligned
// block, so the first log(tile) levels of ENTER (bottom-up) and the last log(tile) levels of EXIT
// (top-down) run in ONE launch with a single HBM round trip: every pre-scale, butterfly stage and
// pointwise step of those levels happens in LDS.  `LevelTables` is the per-tree table set of
// device_tree.h (DeviceChain::Tree), indexed by log2(m).
// ---------------------------------------------------------------------------------------------
template <class F>
struct LevelTables {
    using E = typename F::elem;
    using TE = typename F::telem;
    size_t m, e; unsigned log_m;
    // butterfly / fused pointwise constants, in the form the kernels multiply by (F::telem)
    TE *p0[2], *p1[2], *np0[2], *dinv[2];
    TE *w[2], *winv[2];
    TE *xe, *w1x, *A1, *B1, *NB2, *C1, *D1, *xie;
    TE *inner[2];   // inner[srcpar] = {c0, c1}: the merged innermost (h = 1) decompose+recombine stage, out_j = a + c_j*(b - a)
    TE *c0t[2];     // c0t[s] = np0[s]*dinv[s]: decompose as two INDEPENDENT multiplies, q0 = a + c0t*(b - a), q1 = dinv*(b - a) (pair-split sweeps)
    E *xnn, *xnn_inv, *z0_s1, *z1_s0, *z0_inv_s1, *z1_inv_s0, *z0z0, *z1z1;
    // blk16_A[srcpar] / blk16_K[srcpar]: the innermost 16-point map of an EXTEND from parity srcpar as int8 matrices + accumulator
    // seeds for the matrix cores (mfma_blk16.h); nullptr: the VALU sweeps run (4-byte fields, trees with e < 16, shard contexts)
    const uint8_t* blk16_A[2]; const unsigned long long* blk16_K[2];
    // low16_A / _K (only in the entry of the tree with 16 leaves): levels 1..4 of ENTER [0] and levels 4..1 of EXIT [1] of a 16-block
    // as ONE 16 x 16 map in the same form - every 16-block of a transform goes through the same four levels on the same tables
    const uint8_t* low16_A[2]; const unsigned long long* low16_K[2];
    // low32_A / _K (only in the entry of the tree with 32 leaves; round 4): levels 1..5 of ENTER [0] / 5..1 of EXIT [1] of a 32-block as
    // ONE 32 x 32 map (1 MiB of matrices each, Blk16::phase32); nullptr: low16 (or the level code) runs
    const uint8_t* low32_A[2]; const unsigned long long* low32_K[2];
};

// ---------------------------------------------------------------------------------------------
// Register-resident, one-element-per-thread stage engine for 32-byte fields in the latency regime (tiles of at most one
// element per thread).  The pair-split sweeps above pay an LDS round trip and TWO workgroup barriers per stage for ONE
// multiply of work; here thread `tid` keeps element `tid` of the tile in registers through every in-tile stage and fetches its
// partner tid ^ h with eight cross-lane moves whenever h < 64 (same wave: no LDS traffic, no barrier), through LDS otherwise.
// Roles are per lane: bit lh of tid says whether the thread produces the low or the high output of its pair — one multiply
// each, decompose through c0t = np0*dinv as in the pair-split form, the hi role as t*d + 0 so divergent lanes share ONE
// instruction stream.  Stages k_first .. log_e-1 of vectors of length e = 2^log_e laid end to end in a[0, len); tables are
// the per-tree bases (entry e - 2h + i).  `a` must have been published with a barrier; ends with a barrier.
// ---------------------------------------------------------------------------------------------
#ifndef ECFFT_REG_ENGINE
#define ECFFT_REG_ENGINE 1
#endif
#ifndef ECFFT_N16_DEPTH2W
#define ECFFT_N16_DEPTH2W 2     // units of constant matrices in flight in the two-wave 16x16x64 phases of k_exit_low<8,128> (1: 252 instead of 284 VGPRs, measured 0.6-1.2 % slower)
#endif
#ifndef ECFFT_CORE_OPAQUE_TID
#define ECFFT_CORE_OPAQUE_TID 1  // lds_extend_core: per-call opaque thread index (no hoisting of its LDS addresses out of the callers' level loops)
#endif
#ifndef ECFFT_LDS_SWZ
#define ECFFT_LDS_SWZ 1      // k_exit_low<10,512>: XOR-swizzled LDS layout of the 32-byte elements (A/B: -DECFFT_LDS_SWZ=0)
#endif
// LDS layout of an array of 32-byte elements that ds_read_b128 / ds_write_b128 reach without bank conflicts (round 4): the 16-byte
// chunks 2j, 2j + 1 of element j are XOR-ed with (j >> 3) & 3 inside their 256-byte row (16 chunks = all 64 banks).  A wave-wide
// access takes the same half of each element: in the plain layout the 16 lanes of one of ds_read_b128's groups hit every second bank
// quad twice for consecutive elements (2-way: the 2.1x of SQ_LDS_BANK_CONFLICT over active LDS cycles in k_exit_low) and every fourth
// one four times for the stride-2 de-interleaving reads; with the XOR both patterns cover all sixteen quads.
template <class F, bool SWZ>
__device__ __forceinline__ typename F::elem lds_get(const typename F::elem* a, uint32_t j) {
    if constexpr (SWZ && sizeof(typename F::elem) == 32) {
        const uint4* p = reinterpret_cast<const uint4*>(a);
        const uint32_t sx = (j >> 3) & 3u;
        const uint4 lo = p[(2 * j) ^ sx], hi = p[(2 * j + 1) ^ sx];
        typename F::elem r; r.l[0] = lo.x; r.l[1] = lo.y; r.l[2] = lo.z; r.l[3] = lo.w; r.l[4] = hi.x; r.l[5] = hi.y; r.l[6] = hi.z; r.l[7] = hi.w;
        return r;
    } else return a[j];
}
template <class F, bool SWZ>
__device__ __forceinline__ void lds_put(typename F::elem* a, uint32_t j, const typename F::elem& x) {
    if constexpr (SWZ && sizeof(typename F::elem) == 32) {
        uint4* p = reinterpret_cast<uint4*>(a);
        const uint32_t sx = (j >> 3) & 3u;
        p[(2 * j) ^ sx] = make_uint4(x.l[0], x.l[1], x.l[2], x.l[3]);
        p[(2 * j + 1) ^ sx] = make_uint4(x.l[4], x.l[5], x.l[6], x.l[7]);
    } else a[j] = x;
}
template <class F>
__device__ __forceinline__ typename F::elem lane_xor(const typename F::elem& x, uint32_t h) {
    typename F::elem r;
#pragma unroll
    for (int w = 0; w < 8; ++w) r.l[w] = (uint32_t)__shfl_xor((int)x.l[w], (int)h);
    return r;
}
template <class F>
__device__ __forceinline__ typename F::elem lane_sel(bool c, const typename F::elem& p, const typename F::elem& q) {
    typename F::elem r;
#pragma unroll
    for (int w = 0; w < 8; ++w) r.l[w] = c ? p.l[w] : q.l[w];
    return r;
}
// ---------------------------------------------------------------------------------------------
// Column passes of the LATENCY regime (round 4): 256-element column tiles (2^R rows x C columns), ONE element per thread kept in
// registers through the R stages — the column-tile counterpart of reg_extend32 / k_stages_row256.  Thread tid holds (row
// tid >> log_c, column tid & (C - 1)); its partner at stage s' (row distance d = 2^s') is thread tid ^ (d C): eight cross-lane
// moves when d C < 64, an LDS exchange otherwise.  One multiply per thread per stage (decompose through c0t = np0 dinv), the roles
// sharing one instruction stream as in reg_extend32.  Same arithmetic, same order as col_stage_sweep: bit-identical.
// `a`: 256 (or, two vectors per workgroup, 512) elements of LDS for the exchanges.
// ---------------------------------------------------------------------------------------------
template <class F, bool DEC>
__device__ __forceinline__ void reg_col_stages(typename F::elem& x, typename F::elem* a, uint32_t R, uint32_t log_c, uint32_t log_hs, uint32_t c0, uint32_t e,
                                               const typename F::telem* __restrict__ ta,     // DEC: c0t = np0*dinv | p0
                                               const typename F::telem* __restrict__ tb,     // DEC: dinv | p1
                                               uint32_t tid) {
    using E = typename F::elem;
    using TE = typename F::telem;
    static_assert(sizeof(E) == 32, "32-byte fields");
    const uint32_t cc = tid & ((1u << log_c) - 1), r = tid >> log_c;        // r may carry a vector-select bit above bit R - 1
    for (uint32_t st = 0; st < R; ++st) {
        const uint32_t sft = DEC ? R - 1 - st : st, d = 1u << sft, h = d << log_c;
        const bool hi = (r >> sft) & 1u;
        const uint32_t ti = (e - 2u * ((1u << log_hs) << sft)) + ((r & (d - 1)) << log_hs) + c0 + cc;
        const TE t = hi ? ldt(tb, ti) : ldt(ta, ti);
        E xp;
        if (h < 64) xp = lane_xor<F>(x, h);
        else { __syncthreads(); a[tid] = x; __syncthreads(); xp = a[tid ^ h]; }
        const E A = lane_sel<F>(hi, xp, x), B = lane_sel<F>(hi, x, xp);
        if (DEC) x = F::tmul_add(t, F::sub(B, A), lane_sel<F>(hi, F::zero(), A));     // lo: a + c0t*(b - a)   hi: dinv*(b - a)
        else x = F::tmul_add(t, B, A);                                               // a + p*b
    }
}

// k_stages_col for 256-element tiles: see reg_col_stages
template <class F, bool DECOMPOSE>
__global__ __launch_bounds__(256, 2) void k_stages_col256(IoDesc<F> io, const typename F::telem* __restrict__ ta, const typename F::telem* __restrict__ tb,
                                                          uint32_t log_e, uint32_t ka, uint32_t kb, uint32_t log_c) {
    using E = typename F::elem;
    if constexpr (sizeof(E) == 32) {
        __shared__ E xch[256];
        const uint32_t R = kb - ka + 1, tid = threadIdx.x;
        const uint32_t log_hs = log_e - kb - 1, chunks_log = log_hs - log_c;
        const size_t blk = (size_t)blockIdx.x >> chunks_log, chunk = (size_t)blockIdx.x & (((size_t)1 << chunks_log) - 1);
        const uint32_t c0 = (uint32_t)(chunk << log_c);
        const size_t pos = (blk << (log_hs + R)) + c0 + ((size_t)(tid >> log_c) << log_hs) + (tid & ((1u << log_c) - 1));
        E x = io_load<F>(io, pos, ((size_t)1 << log_e) - 1);
        reg_col_stages<F, DECOMPOSE>(x, xch, R, log_c, log_hs, c0, 1u << log_e, ta, tb, tid);
        io_store<F>(io, pos, log_e, x);
    }
}
// k_stages_col_mid for 256-element tiles
template <class F>
__global__ __launch_bounds__(256, 2) void k_stages_col_mid256(IoDesc<F> io, const typename F::telem* __restrict__ p0, const typename F::telem* __restrict__ p1,
                                                              const typename F::telem* __restrict__ c0t, const typename F::telem* __restrict__ dinv,
                                                              uint32_t log_e, uint32_t ka, uint32_t kb, uint32_t log_c) {
    using E = typename F::elem;
    if constexpr (sizeof(E) == 32) {
        __shared__ E xch[256];
        const uint32_t R = kb - ka + 1, tid = threadIdx.x;
        const uint32_t log_hs = log_e - kb - 1, chunks_log = log_hs - log_c;
        const size_t blk = (size_t)blockIdx.x >> chunks_log, chunk = (size_t)blockIdx.x & (((size_t)1 << chunks_log) - 1);
        const uint32_t c0 = (uint32_t)(chunk << log_c);
        const size_t pos = (blk << (log_hs + R)) + c0 + ((size_t)(tid >> log_c) << log_hs) + (tid & ((1u << log_c) - 1));
        E x = io.src[pos];
        reg_col_stages<F, false>(x, xch, R, log_c, log_hs, c0, 1u << log_e, p0, p1, tid);
        x = io_mid<F>(io, pos, ((size_t)1 << log_e) - 1, x);
        reg_col_stages<F, true>(x, xch, R, log_c, log_hs, c0, 1u << log_e, c0t, dinv, tid);
        data_st(&io.dst[pos], F::canon(x), io.nt_st);
    }
}
// k_stages_col_enter for 256-element tiles: 512 threads, thread tid < 256 holds the U element, tid >= 256 the V element of the same
// column position; the combine's odd output w1 U + w1x V is one multiply per thread and an addition
template <class F>
__global__ __launch_bounds__(512, 2) void k_stages_col_enter256(const typename F::elem* __restrict__ work, const typename F::elem* __restrict__ src,
                                                                typename F::elem* __restrict__ dst,
                                                                const typename F::telem* __restrict__ p0, const typename F::telem* __restrict__ p1,
                                                                const typename F::telem* __restrict__ xe, const typename F::telem* __restrict__ w1,
                                                                const typename F::telem* __restrict__ w1x,
                                                                uint32_t log_e, uint32_t kb, uint32_t log_c) {
    using E = typename F::elem;
    if constexpr (sizeof(E) == 32) {
        __shared__ E xch[512];
        const uint32_t R = kb + 1, tid = threadIdx.x;
        const uint32_t log_hs = log_e - R, chunks_log = log_hs - log_c;
        const size_t b = (size_t)blockIdx.x >> chunks_log, chunk = (size_t)blockIdx.x & (((size_t)1 << chunks_log) - 1);
        const uint32_t c0 = (uint32_t)(chunk << log_c);
        const uint32_t rf = tid >> log_c, v = rf >> R, r = rf & ((1u << R) - 1), cc = tid & ((1u << log_c) - 1);
        const size_t i = ((size_t)r << log_hs) + c0 + cc, bb = b << (log_e + 1), e = (size_t)1 << log_e;
        E x = work[bb + ((size_t)v << log_e) + i];
        reg_col_stages<F, false>(x, xch, R, log_c, log_hs, c0, 1u << log_e, p0, p1, tid);
        const E y = F::tmul(v ? ldt(w1x, (uint32_t)i) : ldt(w1, (uint32_t)i), x);
        __syncthreads();
        xch[tid] = y;
        __syncthreads();
        if (v) dst[bb + 2 * i] = F::canon(F::tmul_add(ldt(xe, (uint32_t)i), src[bb + e + i], src[bb + i]));
        else dst[bb + 2 * i + 1] = F::canon(F::add(xch[tid], xch[tid + 256]));
    }
}

template <class F, int BLK, bool SWZ>
__device__ __forceinline__ void reg_extend32(typename F::elem* a, uint32_t len, uint32_t log_e, uint32_t k_first,
                                             const typename F::telem* __restrict__ c0t, const typename F::telem* __restrict__ dinv,
                                             const typename F::telem* __restrict__ p0, const typename F::telem* __restrict__ p1,
                                             const typename F::telem* __restrict__ inner, uint32_t tid,
                                             const uint8_t* __restrict__ blkA, const unsigned long long* __restrict__ blkK) {
    // blkA != nullptr (len == BLK == 512, 256 or 128, log_e - k_first >= 4): the stages with pair distance <= 8 run on the matrix cores
    using E = typename F::elem;
    using TE = typename F::telem;
    static_assert(sizeof(E) == 32, "32-byte fields");
    const bool act = tid < len;                                             // len is a multiple of 64: wave-uniform
    const size_t e = (size_t)1 << log_e;
    // SWZ: `a` is read and written in the XOR-swizzled layout (lds_get / lds_put); the matrix-core phase uses its own operand layout
    E x = act ? lds_get<F, SWZ>(a, tid) : F::zero();
    auto partner = [&](uint32_t h) -> E {
        if (h < 64) return act ? lane_xor<F>(x, h) : x;
        __syncthreads();                                                    // readers of the previous LDS exchange are done
        if (act) lds_put<F, SWZ>(a, tid, x);
        __syncthreads();
        return act ? lds_get<F, SWZ>(a, tid ^ h) : x;
    };
    const uint32_t k_inner = log_e ? log_e - 1 : 0;
    bool mfma = false;
    if constexpr (BLK == 512 || BLK == 256 || BLK == 128) mfma = blkA != nullptr;
    const uint32_t k_dec_end = mfma ? log_e - 4 : k_inner;
    for (uint32_t k = k_first; k < k_dec_end; ++k) {
        const uint32_t lh = log_e - k - 1, h = 1u << lh;
        const bool hi = (tid >> lh) & 1u;
        const uint32_t off = (uint32_t)(e - 2 * (size_t)h) + (tid & (h - 1));
        TE t; if (act) t = hi ? ldt(dinv, off) : ldt(c0t, off);
        const E xp = partner(h);
        if (act) {
            const E A = lane_sel<F>(hi, xp, x), B = lane_sel<F>(hi, x, xp);
            x = F::tmul_add(t, F::sub(B, A), lane_sel<F>(hi, F::zero(), A));   // lo: a + c0t*(b - a)   hi: dinv*(b - a)
        }
    }
    if constexpr (BLK == 512 || BLK == 256 || BLK == 128) {
        if (mfma) {                                                         // len == BLK: every thread holds an element
            if constexpr (BLK == 512) x = Blk16::phase512_regs(a, x, blkA, blkK, tid);
            else x = Blk16::phase_n16_regs<BLK / 64, (BLK == 128 ? ECFFT_N16_DEPTH2W : 2)>(a, x, blkA, blkK, tid);   // small launches: v_mfma_i32_16x16x64_i8
        }
    }
    if (log_e > 0 && k_first <= k_inner && !mfma) {                         // merged innermost stage pair (h = 1)
        const bool hi = tid & 1u;
        TE t; if (act) t = ldt(inner, hi ? 1u : 0u);
        const E xp = partner(1);
        if (act) {
            const E A = lane_sel<F>(hi, xp, x), B = lane_sel<F>(hi, x, xp);
            x = F::tmul_add(t, F::sub(B, A), A);
        }
    }
    for (uint32_t k = k_dec_end; k-- > k_first;) {
        const uint32_t lh = log_e - k - 1, h = 1u << lh;
        const bool hi = (tid >> lh) & 1u;
        const uint32_t off = (uint32_t)(e - 2 * (size_t)h) + (tid & (h - 1));
        TE t; if (act) t = hi ? ldt(p1, off) : ldt(p0, off);
        const E xp = partner(h);
        if (act) {
            const E A = lane_sel<F>(hi, xp, x), B = lane_sel<F>(hi, x, xp);
            x = F::tmul_add(t, B, A);                                        // a + p*b
        }
    }
    __syncthreads();
    if (act) lds_put<F, SWZ>(a, tid, x);
    __syncthreads();
}

// every stage (decompose then recombine) of EXTEND on `len` LDS elements = len/e vectors of length e;
// srcpar = parity of the source moiety.  Ends with a barrier.
template <class F, int BLK = kBlockLds, bool SWZ = false>      // SWZ: `a` in the XOR-swizzled layout — only the register engine (len <= BLK) reads it that way
__device__ __forceinline__ void lds_extend_core(typename F::elem* a, uint32_t len, uint32_t log_e, const LevelTables<F>& T, int srcpar) {
    using E = typename F::elem;
    // The thread index is made opaque per call (32-byte fields): the callers run this core once per LEVEL in a loop, and with a
    // loop-invariant tid the compiler hoists every LDS address of the matrix-core phase's swizzled layout (16 xor-ed offsets and
    // more) out of that loop and keeps them alive across the 110-VGPR multiplies — k_enter_low<10,512> spilled 60 B for it.
    // Recomputing them per call costs a few dozen integer instructions per level.
    uint32_t tid_ = threadIdx.x;
#if ECFFT_CORE_OPAQUE_TID
    if constexpr (sizeof(E) == 32) asm volatile("" : "+v"(tid_));
#endif
    const uint32_t tid = tid_, npairs = len >> 1;
    const size_t e = (size_t)1 << log_e;
    const int tgt = 1 - srcpar;
    // matrix-core form of the stages with pair distance <= 8 (mfma_blk16.h): arrays of 512 (one element per thread) or of whole
    // 1024-element sub-tiles, vectors of >= 16 elements
    const uint8_t* bA = nullptr; const unsigned long long* bK = nullptr;
    if constexpr (sizeof(E) == 32 && BLK == 512) {
        if (log_e >= 4 && (len == 512u || (len & 1023u) == 0)) { bA = T.blk16_A[srcpar]; bK = T.blk16_K[srcpar]; }
    }
    if constexpr (sizeof(E) == 32 && (BLK == 256 || BLK == 128)) {          // latency variants: one element per thread
        if (log_e >= 4 && len == (uint32_t)BLK) { bA = T.blk16_A[srcpar]; bK = T.blk16_K[srcpar]; }
    }
    if constexpr (sizeof(E) == 32 && ECFFT_REG_ENGINE) {
        if (len <= (uint32_t)BLK && (len & 63u) == 0) {
            reg_extend32<F, BLK, SWZ>(a, len, log_e, 0, T.c0t[srcpar], T.dinv[srcpar], T.p0[tgt], T.p1[tgt], T.inner[srcpar], tid, len == (uint32_t)BLK ? bA : nullptr, bK);
            return;
        }
    }
    if constexpr (SWZ) __builtin_trap();      // unreachable: the callers that swizzle pass len == BLK
    if constexpr (sizeof(E) == 32) {
        if (len == BLK) {
            // PAIR-SPLIT sweeps (k_exit_low: half as many pairs as threads).  One butterfly = two multiplies; with one pair per
            // thread half the workgroup idles and every sweep costs two dependent 169-instruction multiplies.  Here waves 0..3
            // produce the LOW output of every pair and waves 4..7 the HIGH one, one multiply each (decompose through the extra
            // table c0t = np0*dinv, which makes its two products independent): a sweep costs one multiply of latency and all
            // eight waves issue.  Both roles read (a, b); a barrier separates the reads from the in-place writes.
            const uint32_t g = tid & (npairs - 1);
            const bool hi = tid >= npairs;                                  // wave-uniform: npairs is a multiple of 64
            const uint32_t k_inner = log_e ? log_e - 1 : 0;
            for (uint32_t k = 0; k < k_inner; ++k) {
                const uint32_t lh = log_e - k - 1, h = 1u << lh, i = g & (h - 1), idx = ((g >> lh) << (lh + 1)) + i;
                const size_t off = e - 2 * (size_t)h + i;
                const typename F::telem t = hi ? ldt(T.dinv[srcpar], (uint32_t)off) : ldt(T.c0t[srcpar], (uint32_t)off);
                const E x = a[idx], y = a[idx + h];
                __syncthreads();
                const E d = F::sub(y, x);
                if (hi) a[idx + h] = F::tmul(t, d); else a[idx] = F::tmul_add(t, d, x);
                __syncthreads();
            }
            if (log_e > 0) {
                const typename F::telem t = ldt(T.inner[srcpar], hi ? 1u : 0u);
                const E x = a[2 * g], y = a[2 * g + 1];
                __syncthreads();
                a[2 * g + (hi ? 1 : 0)] = F::tmul_add(t, F::sub(y, x), x);
                __syncthreads();
            }
            for (uint32_t k = k_inner; k-- > 0;) {
                const uint32_t lh = log_e - k - 1, h = 1u << lh, i = g & (h - 1), idx = ((g >> lh) << (lh + 1)) + i;
                const size_t off = e - 2 * (size_t)h + i;
                const typename F::telem t = hi ? ldt(T.p1[tgt], (uint32_t)off) : ldt(T.p0[tgt], (uint32_t)off);
                const E x = a[idx], y = a[idx + h];
                __syncthreads();
                a[idx + (hi ? h : 0)] = F::tmul_add(t, y, x);
                __syncthreads();
            }
            return;
        }
    }
    if constexpr (sizeof(E) == 4) {
        // callers have published `a` with a barrier; the engine starts with one of its own and ends with one
        if (len == BLK * 16) { lds_extend_fast<F, 16, BLK>(a, T.np0[srcpar], T.dinv[srcpar], T.p0[tgt], T.p1[tgt], T.inner[srcpar], e, log_e, tid); return; }
        if (len == BLK * 8) { lds_extend_fast<F, 8, BLK>(a, T.np0[srcpar], T.dinv[srcpar], T.p0[tgt], T.p1[tgt], T.inner[srcpar], e, log_e, tid); return; }
    }
    const uint32_t k_inner = log_e ? log_e - 1 : 0;
    bool mfma = false;
    if constexpr (sizeof(E) == 32 && BLK == 512) mfma = bA != nullptr && (len & 1023u) == 0;
    const uint32_t k_dec_end = mfma ? log_e - 4 : k_inner;
    // one-pair-per-thread sweeps (32-byte fields, at least one pair per thread): the sweeps at pair distance <= 64 stay inside the
    // wave's own span, and so does the merged innermost stage — no workgroup barrier between them (wave_local_sync)
    const bool wl = sizeof(E) == 32 && 2 * npairs > (uint32_t)BLK;
    for (uint32_t k = 0; k < k_dec_end; ++k) {
        const uint32_t lh = log_e - k - 1, h = 1u << lh;
        stage_sweep<F, true, BLK>(a, T.np0[srcpar] + (e - 2 * (size_t)h), T.dinv[srcpar] + (e - 2 * (size_t)h), lh, npairs, tid, T.c0t[srcpar] + (e - 2 * (size_t)h));
        if (wl && wave_local_lh((int)lh) && (k + 1 < k_dec_end || !mfma)) wave_local_sync(); else __syncthreads();
    }
    if constexpr (sizeof(E) == 32 && BLK == 512) {
        if (mfma) {
            Blk16::APre pre = Blk16::prefetch(bA, tid);
            __builtin_amdgcn_sched_barrier(0);
            Blk16::to_operand_form<BLK>(a, len, tid);
#pragma unroll 1
            for (uint32_t o = 0; o < len; o += Blk16::kSub) {
                if (o) { pre = Blk16::prefetch(bA, tid); __builtin_amdgcn_sched_barrier(0); }
                Blk16::phase(a + o, bA, bK, tid, pre);
            }
            Blk16::from_swizzled<BLK>(a, len, tid);
        }
    }
    if (log_e > 0 && !mfma) {                           // merged innermost stage pair (h = 1)
        const typename F::telem c0 = ldt(T.inner[srcpar], 0u), c1 = ldt(T.inner[srcpar], 1u);
        for (uint32_t g = tid; g < npairs; g += BLK) {
            E x = a[2 * g], y = a[2 * g + 1];
            E d = F::sub(y, x);
            a[2 * g] = F::tmul_add(c0, d, x);
            a[2 * g + 1] = F::tmul_add(c1, d, x);
        }
        if (wl && ECFFT_WAVE_LOCAL && k_dec_end > 0) wave_local_sync(); else __syncthreads();
    }
    for (uint32_t k = k_dec_end; k-- > 0;) {
        const uint32_t lh = log_e - k - 1, h = 1u << lh;
        stage_sweep<F, false, BLK>(a, T.p0[tgt] + (e - 2 * (size_t)h), T.p1[tgt] + (e - 2 * (size_t)h), lh, npairs, tid);
        if (wl && k > 0 && wave_local_lh((int)lh + 1)) wave_local_sync(); else __syncthreads();
    }
}

// ENTER levels 1 .. log_tile (src/fftree.rs:143-161 for every block of size <= tile).  LDS: 2*tile elements.
template <class F, int LOG_TILE, int BLK = kBlockLds>
__global__ __launch_bounds__(BLK, (BLK >= 512 ? ECFFT_MIN_WAVES : 2)) void k_enter_low(typename F::elem* __restrict__ dst, const typename F::elem* __restrict__ src,
                                                          const LevelTables<F>* __restrict__ trees) {
    using E = typename F::elem;
    extern __shared__ __attribute__((aligned(16))) unsigned char ecfft_smem[];
    constexpr uint32_t log_tile = LOG_TILE, T = 1u << LOG_TILE, npairs = T >> 1;
    constexpr bool kRoles = npairs * 2 == (uint32_t)BLK;    // latency variant: two threads per pair (one output each)
    constexpr int PAIRS = kRoles ? 1 : (int)(npairs / BLK);     // compile-time trip counts keep ev/od in registers
    static_assert(kRoles || (npairs >= (uint32_t)BLK && npairs % BLK == 0), "tile too small for the workgroup");
    const uint32_t tid = threadIdx.x;
    E* cur = reinterpret_cast<E*>(ecfft_smem);
    E* work = cur + T;
    const size_t base = (size_t)blockIdx.x << log_tile;
    constexpr bool kQuad = sizeof(E) == 4 && T % (4 * BLK) == 0;     // 4-byte fields: quad-vectorised, loads-first pointwise steps
    bool qio = false;                                                      // user pointers may be only element-aligned
    if constexpr (kQuad) qio = ((reinterpret_cast<uintptr_t>(src) | reinterpret_cast<uintptr_t>(dst)) & 15) == 0;
    if constexpr (kQuad) {
        if (qio) {
            constexpr int NQ = (int)(T / (4 * BLK));
            Quad d[NQ];
#pragma unroll
            for (int c = 0; c < NQ; ++c) d[c] = ldq(src + base + 4u * (tid + (uint32_t)c * BLK));
#pragma unroll
            for (int c = 0; c < NQ; ++c) stq(cur + 4u * (tid + (uint32_t)c * BLK), d[c]);
        }
    }
    if (!qio) {
        for (uint32_t j = tid; j < T; j += BLK) cur[j] = src[base + j];
    }
    __syncthreads();
    uint32_t l_first = 1;
    if constexpr (sizeof(E) == 32 && LOG_TILE == 10 && BLK == 512) {
        // levels 1..5 on the matrix cores as one 32 x 32 map per 32-block (LevelTables::low32_A), else
        // levels 1..4 as one 16 x 16 map per 16-block (LevelTables::low16_A)
        const uint8_t* lA = trees[4].low16_A[0];
        const uint8_t* lA32 = trees[5].low32_A[0];
        if (lA32) {
            Blk16::to_operand_form32(cur, tid);
            Blk16::phase32(cur, lA32, trees[5].low32_K[0], tid);
            Blk16::from_swizzled32(cur, tid);
            l_first = 6;
        } else if (lA) {
            Blk16::APre pre = Blk16::prefetch(lA, tid);
            __builtin_amdgcn_sched_barrier(0);
            Blk16::to_operand_form<BLK>(cur, T, tid);
            Blk16::phase(cur, lA, trees[4].low16_K[0], tid, pre);
            Blk16::from_swizzled<BLK>(cur, T, tid);
            l_first = 5;
        }
    }
    if constexpr (sizeof(E) == 32 && LOG_TILE == 8 && BLK == 256) {
        // latency variant: the same map on the 16 blocks of the 256-element tile (v_mfma_i32_16x16x64_i8, mfma_blk16.h)
        const uint8_t* lA = trees[4].low16_A[0];
        if (lA) {
            Blk16::phase_n16<4, false>(cur, lA, trees[4].low16_K[0], tid, [&] { Blk16::to_operand_form<BLK>(cur, T, tid); });
            Blk16::from_swizzled<BLK>(cur, T, tid);
            l_first = 5;
        }
    }
    for (uint32_t l = l_first; l <= log_tile; ++l) {
        const LevelTables<F>& L = trees[l];
        const uint32_t le = l - 1, e = 1u << le;
        if constexpr (kQuad) {
            if (le >= 2) {
                constexpr int NQ = (int)(T / (4 * BLK)), NP = NQ / 2;
                {
                    Quad x[NQ], t[NQ];
                    const typename F::telem* wi = L.winv[0];
#pragma unroll
                    for (int c = 0; c < NQ; ++c) { const uint32_t j = 4u * (tid + (uint32_t)c * BLK); t[c] = ldq_tab(wi, j & (e - 1)); x[c] = ldq(cur + j); }
#pragma unroll
                    for (int c = 0; c < NQ; ++c) {
#pragma unroll
                        for (int k = 0; k < 4; ++k) x[c].v[k] = F::tmul(t[c].v[k], x[c].v[k]);
                        stq(work + 4u * (tid + (uint32_t)c * BLK), x[c]);
                    }
                }
                __syncthreads();
                lds_extend_core<F, BLK>(work, T, le, L, 0);
                Quad lo[NP], hi[NP];
                {
                    Quad u0[NP], v0[NP], U1[NP], V1[NP], tx[NP], tw[NP], twx[NP];
                    const typename F::telem *xe = L.xe, *w1 = L.w[1], *w1x = L.w1x;
#pragma unroll
                    for (int c = 0; c < NP; ++c) {
                        const uint32_t g = 4u * (tid + (uint32_t)c * BLK), i = g & (e - 1), bb = (g >> le) << l;
                        tx[c] = ldq_tab(xe, (uint32_t)i); tw[c] = ldq_tab(w1, (uint32_t)i); twx[c] = ldq_tab(w1x, (uint32_t)i);
                        u0[c] = ldq(cur + bb + i); v0[c] = ldq(cur + bb + e + i); U1[c] = ldq(work + bb + i); V1[c] = ldq(work + bb + e + i);
                    }
#pragma unroll
                    for (int c = 0; c < NP; ++c)
#pragma unroll
                        for (int k = 0; k < 4; ++k) {
                            const E ev = F::tmul_add(tx[c].v[k], v0[c].v[k], u0[c].v[k]);
                            const E od = F::tmul_add(twx[c].v[k], V1[c].v[k], F::tmul(tw[c].v[k], U1[c].v[k]));
                            if (k < 2) { lo[c].v[2 * k] = ev; lo[c].v[2 * k + 1] = od; } else { hi[c].v[2 * (k - 2)] = ev; hi[c].v[2 * (k - 2) + 1] = od; }
                        }
                }
                __syncthreads();
#pragma unroll
                for (int c = 0; c < NP; ++c) {
                    const uint32_t g = 4u * (tid + (uint32_t)c * BLK), i = g & (e - 1), bb = (g >> le) << l;
                    stq(cur + bb + 2 * i, lo[c]); stq(cur + bb + 2 * i + 4, hi[c]);
                }
                __syncthreads();
                continue;
            }
        }
        for (uint32_t j = tid; j < T; j += BLK) work[j] = F::tmul(ldt(L.winv[0], j & (e - 1)), cur[j]);
        __syncthreads();
        lds_extend_core<F, BLK>(work, T, le, L, 0);
        // combine (:155-159): block [u0|v0] + extended [U1|V1] -> interleaved evaluations; results are held in
        // registers across the barrier because the interleaving store overwrites other threads' inputs
        if constexpr (kRoles) {
            const bool hi = tid >= npairs;
            const uint32_t g = hi ? tid - npairs : tid, i = g & (e - 1), bb = (g >> le) << l;
            E r;
            if (!hi) r = F::tmul_add(ldt(L.xe, i), cur[bb + e + i], cur[bb + i]);
            else r = F::tmul_add(ldt(L.w1x, i), work[bb + e + i], F::tmul(ldt(L.w[1], i), work[bb + i]));
            __syncthreads();
            cur[bb + 2 * i + (hi ? 1 : 0)] = r;
            __syncthreads();
            continue;
        }
        E ev[PAIRS], od[PAIRS];
#pragma unroll
        for (int c = 0; c < PAIRS; ++c) {
            uint32_t g = tid + (uint32_t)c * BLK, i = g & (e - 1), bb = (g >> le) << l;
            E u0 = cur[bb + i], v0 = cur[bb + e + i], U1 = work[bb + i], V1 = work[bb + e + i];
            ev[c] = F::tmul_add(ldt(L.xe, i), v0, u0);
            od[c] = F::tmul_add(ldt(L.w1x, i), V1, F::tmul(ldt(L.w[1], i), U1));
        }
        __syncthreads();
#pragma unroll
        for (int c = 0; c < PAIRS; ++c) {
            uint32_t g = tid + (uint32_t)c * BLK, i = g & (e - 1), bb = (g >> le) << l;
            cur[bb + 2 * i] = ev[c]; cur[bb + 2 * i + 1] = od[c];
        }
        __syncthreads();
    }
    if constexpr (kQuad) {
        if (qio) {
            constexpr int NQ = (int)(T / (4 * BLK));
#pragma unroll
            for (int c = 0; c < NQ; ++c) {
                Quad q = ldq(cur + 4u * (tid + (uint32_t)c * BLK));
#pragma unroll
                for (int k = 0; k < 4; ++k) q.v[k] = F::canon(q.v[k]);
                stq(dst + base + 4u * (tid + (uint32_t)c * BLK), q);
            }
            return;
        }
    }
    for (uint32_t j = tid; j < T; j += BLK) dst[base + j] = F::canon(cur[j]);
}

// EXIT levels log_tile .. 1 (src/fftree.rs:200-224 with redc_impl :232-259 inlined, normalised form, see
// DeviceChain::exit).  LDS: cur (tile) + G (tile/2) + H (tile/2).
template <class F, int LOG_TILE, int BLK = kBlockLds>
__global__ __launch_bounds__(BLK, (BLK >= 512 ? ECFFT_MIN_WAVES : 1)) void k_exit_low(typename F::elem* __restrict__ dst, const typename F::elem* __restrict__ src,
                                                         const LevelTables<F>* __restrict__ trees) {
    using E = typename F::elem;
    extern __shared__ __attribute__((aligned(16))) unsigned char ecfft_smem[];
    constexpr uint32_t log_tile = LOG_TILE, T = 1u << LOG_TILE, nh = T >> 1;
    constexpr int PAIRS = (int)(nh / BLK);
    static_assert(PAIRS >= 1 && nh % BLK == 0, "tile too small for the workgroup");
    const uint32_t tid = threadIdx.x;
    E* cur = reinterpret_cast<E*>(ecfft_smem);
    E* G = cur + T;
    E* H = G + nh;
    const size_t base = (size_t)blockIdx.x << log_tile;
    constexpr bool kQuad = sizeof(E) == 4 && nh % (4 * BLK) == 0;
    // cur / G / H in the XOR-swizzled layout (lds_get / lds_put: no bank conflicts for consecutive elements) — the 1024-element
    // variant of 32-byte fields, whose EXTEND cores run on the register engine
    constexpr bool kSwz = ECFFT_LDS_SWZ && sizeof(E) == 32 && LOG_TILE == 10 && BLK == 512;
    bool qio = false;                                                      // user pointers may be only element-aligned
    if constexpr (kQuad) qio = ((reinterpret_cast<uintptr_t>(src) | reinterpret_cast<uintptr_t>(dst)) & 15) == 0;
    if constexpr (kQuad) {
        if (qio) {
            constexpr int NQ = (int)(T / (4 * BLK));
            Quad d[NQ];
#pragma unroll
            for (int c = 0; c < NQ; ++c) d[c] = ldq(src + base + 4u * (tid + (uint32_t)c * BLK));
#pragma unroll
            for (int c = 0; c < NQ; ++c) stq(cur + 4u * (tid + (uint32_t)c * BLK), d[c]);
        }
    }
    if (!qio) {
        for (uint32_t j = tid; j < T; j += BLK) lds_put<F, kSwz>(cur, j, src[base + j]);
    }
    __syncthreads();
    bool cur_swz = kSwz;                                                    // the final matrix-core phase leaves `cur` in the plain layout
    uint32_t l_last = 1;
    const uint8_t* lA = nullptr;
    const uint8_t* lA32 = nullptr;
    if constexpr (sizeof(E) == 32 && ((LOG_TILE == 10 && BLK == 512) || (LOG_TILE == 8 && BLK == 128))) { lA = trees[4].low16_A[1]; if (lA) l_last = 5; }
    if constexpr (sizeof(E) == 32 && LOG_TILE == 10 && BLK == 512) { lA32 = trees[5].low32_A[1]; if (lA32) l_last = 6; }
    for (uint32_t l = log_tile; l >= l_last; --l) {
        const LevelTables<F>& L = trees[l];
        const uint32_t le = l - 1, e = 1u << le;
        if constexpr (kQuad) {
            if (le >= 2) {
                // the level's five pointwise steps on quads of the pair index g (loads first), the four EXTEND cores between them
                constexpr int NP = (int)(nh / (4 * BLK));
                auto gq = [=](int c) { return 4u * (tid + (uint32_t)c * BLK); };
                auto tq = [=](const typename F::telem* t, int c) { return ldq_tab(t, gq(c) & (e - 1)); };
                auto evenq = [=](int c) { uint4 a = *reinterpret_cast<const uint4*>(cur + 2 * gq(c)), b = *reinterpret_cast<const uint4*>(cur + 2 * gq(c) + 4); return Quad{{a.x, a.z, b.x, b.z}}; };
                auto oddq = [=](int c) { uint4 a = *reinterpret_cast<const uint4*>(cur + 2 * gq(c)), b = *reinterpret_cast<const uint4*>(cur + 2 * gq(c) + 4); return Quad{{a.y, a.w, b.y, b.w}}; };
                {
                    Quad t[NP], x[NP];
#pragma unroll
                    for (int c = 0; c < NP; ++c) { t[c] = tq(L.A1, c); x[c] = evenq(c); }
#pragma unroll
                    for (int c = 0; c < NP; ++c) {
#pragma unroll
                        for (int k = 0; k < 4; ++k) x[c].v[k] = F::tmul(t[c].v[k], x[c].v[k]);
                        stq(G + gq(c), x[c]);
                    }
                }
                __syncthreads();
                lds_extend_core<F, BLK>(G, nh, le, L, 0);
                {
                    Quad ta[NP], tb[NP], x[NP], y[NP];
#pragma unroll
                    for (int c = 0; c < NP; ++c) { ta[c] = tq(L.NB2, c); tb[c] = tq(L.B1, c); x[c] = ldq(G + gq(c)); y[c] = oddq(c); }
#pragma unroll
                    for (int c = 0; c < NP; ++c) {
#pragma unroll
                        for (int k = 0; k < 4; ++k) x[c].v[k] = F::tmul_add(ta[c].v[k], x[c].v[k], F::tmul(tb[c].v[k], y[c].v[k]));
                        stq(G + gq(c), x[c]); stq(H + gq(c), x[c]);
                    }
                }
                __syncthreads();
                lds_extend_core<F, BLK>(G, nh, le, L, 1);
                {
                    Quad t[NP], x[NP];
#pragma unroll
                    for (int c = 0; c < NP; ++c) { t[c] = tq(L.C1, c); x[c] = ldq(G + gq(c)); }
#pragma unroll
                    for (int c = 0; c < NP; ++c) {
#pragma unroll
                        for (int k = 0; k < 4; ++k) x[c].v[k] = F::tmul(t[c].v[k], x[c].v[k]);
                        stq(G + gq(c), x[c]);
                    }
                }
                __syncthreads();
                lds_extend_core<F, BLK>(G, nh, le, L, 0);
                {
                    Quad ta[NP], tb[NP], x[NP], y[NP];
#pragma unroll
                    for (int c = 0; c < NP; ++c) { ta[c] = tq(L.NB2, c); tb[c] = tq(L.D1, c); x[c] = ldq(G + gq(c)); y[c] = ldq(H + gq(c)); }
#pragma unroll
                    for (int c = 0; c < NP; ++c) {
#pragma unroll
                        for (int k = 0; k < 4; ++k) x[c].v[k] = F::tmul_add(ta[c].v[k], x[c].v[k], F::tmul(tb[c].v[k], y[c].v[k]));
                        stq(G + gq(c), x[c]);
                    }
                }
                __syncthreads();
                lds_extend_core<F, BLK>(G, nh, le, L, 1);
                Quad uq[NP], vq[NP];
                {
                    Quad ta[NP], tb[NP], x[NP], y[NP];
#pragma unroll
                    for (int c = 0; c < NP; ++c) { ta[c] = tq(L.w[0], c); tb[c] = tq(L.xie, c); x[c] = ldq(G + gq(c)); y[c] = evenq(c); }
#pragma unroll
                    for (int c = 0; c < NP; ++c)
#pragma unroll
                        for (int k = 0; k < 4; ++k) { uq[c].v[k] = F::tmul(ta[c].v[k], x[c].v[k]); vq[c].v[k] = F::tmul(tb[c].v[k], F::sub(y[c].v[k], uq[c].v[k])); }
                }
                __syncthreads();
#pragma unroll
                for (int c = 0; c < NP; ++c) {
                    const uint32_t g = gq(c), i = g & (e - 1), bb = (g >> le) << l;
                    stq(cur + bb + i, uq[c]); stq(cur + bb + e + i, vq[c]);
                }
                __syncthreads();
                continue;
            }
        }
        for (uint32_t g = tid; g < nh; g += BLK) lds_put<F, kSwz>(G, g, F::tmul(ldt(L.A1, g & (e - 1)), lds_get<F, kSwz>(cur, 2 * g)));
        __syncthreads();
        lds_extend_core<F, BLK, kSwz>(G, nh, le, L, 0);
        for (uint32_t g = tid; g < nh; g += BLK) {
            uint32_t i = g & (e - 1);
            E r = F::tmul_add(ldt(L.NB2, i), lds_get<F, kSwz>(G, g), F::tmul(ldt(L.B1, i), lds_get<F, kSwz>(cur, 2 * g + 1)));
            lds_put<F, kSwz>(G, g, r); lds_put<F, kSwz>(H, g, r);
        }
        __syncthreads();
        lds_extend_core<F, BLK, kSwz>(G, nh, le, L, 1);
        for (uint32_t g = tid; g < nh; g += BLK) lds_put<F, kSwz>(G, g, F::tmul(ldt(L.C1, g & (e - 1)), lds_get<F, kSwz>(G, g)));
        __syncthreads();
        lds_extend_core<F, BLK, kSwz>(G, nh, le, L, 0);
        for (uint32_t g = tid; g < nh; g += BLK) {
            uint32_t i = g & (e - 1);
            lds_put<F, kSwz>(G, g, F::tmul_add(ldt(L.NB2, i), lds_get<F, kSwz>(G, g), F::tmul(ldt(L.D1, i), lds_get<F, kSwz>(H, g))));
        }
        __syncthreads();
        lds_extend_core<F, BLK, kSwz>(G, nh, le, L, 1);
        E u[PAIRS], v[PAIRS];
#pragma unroll
        for (int c = 0; c < PAIRS; ++c) {
            uint32_t g = tid + (uint32_t)c * BLK, i = g & (e - 1);
            u[c] = F::tmul(ldt(L.w[0], i), lds_get<F, kSwz>(G, g));
            v[c] = F::tmul(ldt(L.xie, i), F::sub(lds_get<F, kSwz>(cur, 2 * g), u[c]));
        }
        __syncthreads();
#pragma unroll
        for (int c = 0; c < PAIRS; ++c) {
            uint32_t g = tid + (uint32_t)c * BLK, i = g & (e - 1), bb = (g >> le) << l;
            lds_put<F, kSwz>(cur, bb + i, u[c]); lds_put<F, kSwz>(cur, bb + e + i, v[c]);
        }
        __syncthreads();
    }
    if constexpr (sizeof(E) == 32 && LOG_TILE == 10 && BLK == 512) {
        if (lA32) {             // levels 5..1 on the matrix cores: one 32 x 32 map per 32-block
            if constexpr (kSwz) {                                          // (off by default: the plain layout first)
                const E x0 = lds_get<F, true>(cur, tid), x1 = lds_get<F, true>(cur, tid + 512);
                __syncthreads();
                cur[tid] = x0; cur[tid + 512] = x1;
                __syncthreads();
            }
            Blk16::to_operand_form32(cur, tid);
            Blk16::phase32(cur, lA32, trees[5].low32_K[1], tid);
            Blk16::from_swizzled32(cur, tid);
            cur_swz = false;
        } else
        if (lA) {               // levels 4..1 on the matrix cores: one 16 x 16 map per 16-block
            Blk16::APre pre = Blk16::prefetch(lA, tid);
            __builtin_amdgcn_sched_barrier(0);
            Blk16::to_operand_form<BLK, kSwz>(cur, T, tid);
            Blk16::phase(cur, lA, trees[4].low16_K[1], tid, pre);
            Blk16::from_swizzled<BLK>(cur, T, tid);
            cur_swz = false;
        }
    }
    if constexpr (sizeof(E) == 32 && LOG_TILE == 8 && BLK == 128) {
        if (lA) {               // latency variant: 16 blocks on two waves (v_mfma_i32_16x16x64_i8), two results per lane
            Blk16::phase_n16<2, false, ECFFT_N16_DEPTH2W>(cur, lA, trees[4].low16_K[1], tid, [&] { Blk16::to_operand_form<BLK>(cur, T, tid); });
            Blk16::from_swizzled<BLK>(cur, T, tid);
        }
    }
    if constexpr (kQuad) {
        if (qio) {
            constexpr int NQ = (int)(T / (4 * BLK));
#pragma unroll
            for (int c = 0; c < NQ; ++c) {
                Quad q = ldq(cur + 4u * (tid + (uint32_t)c * BLK));
#pragma unroll
                for (int k = 0; k < 4; ++k) q.v[k] = F::canon(q.v[k]);
                stq(dst + base + 4u * (tid + (uint32_t)c * BLK), q);
            }
            return;
        }
    }
    if (cur_swz) { for (uint32_t j = tid; j < T; j += BLK) dst[base + j] = F::canon(lds_get<F, kSwz>(cur, j)); }
    else for (uint32_t j = tid; j < T; j += BLK) dst[base + j] = F::canon(cur[j]);
}

// ---------------------------------------------------------------------------------------------
// pointwise scaling by a (strided) table: cyclic shards of a multi-GPU split EXTEND.  (The combine step of ENTER,
// src/fftree.rs:155-159, has no kernel of its own: it is the store operator of the level's last EXTEND pass — ST_ENTER in
// k_stages_lds, k_stages_col_enter.)
// ---------------------------------------------------------------------------------------------
// work[j] = src[j] * winv0[j mod e]
template <class F>
__global__ __launch_bounds__(kBlock) void k_scale_by_table(typename F::elem* dst,  // may alias src
                                                            const typename F::elem* src,
                                                            const typename F::telem* __restrict__ tbl,
                                                            size_t tbl_mask, size_t n, uint32_t tstride, uint32_t toff) {
    size_t g = (size_t)blockIdx.x * kBlock + threadIdx.x;
    if (g >= n) return;
    dst[g] = F::canon(F::tmul(tbl[(g & tbl_mask) * tstride + toff], src[g]));
}

// ---------------------------------------------------------------------------------------------
// generic element-wise helper for tree construction: functor(i) for i < n
// ---------------------------------------------------------------------------------------------
template <class Fn>
__global__ __launch_bounds__(kBlock) void k_foreach(Fn fn, size_t n) {
    size_t g = (size_t)blockIdx.x * kBlock + threadIdx.x;
    if (g < n) fn(g);
}

}  // namespace ecfft
