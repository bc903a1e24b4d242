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

namespace ecfft {

constexpr int kBlock = 256;

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
    const E* src; uint32_t src_stride, src_off; int ld_mode; const TE* ld_tbl;
    // store: ST_PLAIN  dst[pos] = x
    //        ST_SCALE  dst[pos] = st_a[i]*x
    //        ST_AXPBY  r = st_a[i]*x + st_b[i]*aux[aux_stride*pos + aux_off]; dst[pos] = r; aux_out[pos] = r (if set)
    //        ST_EXIT_SPLIT  u0 = st_a[i]*x; v0 = st_b[i]*(aux[2*pos] - u0); dst[b*2e + i] = u0; dst[b*2e + e + i] = v0  (b = pos / e)
    //        ST_ENTER  (pair operator, ENTER loop C src/fftree.rs:155-159 in normalised form; x = U1~ at pos = b*2e + i, y = V1~ at
    //                  pos + e, both still in the tile; aux = the level's input blocks [u0 | v0])
    //                  dst[b*2e + 2i] = aux[b*2e + i] + st_a[i]*aux[b*2e + e + i];  dst[b*2e + 2i + 1] = st_c[i]*x + st_b[i]*y
    E* dst; int st_mode; const TE* st_a; const TE* st_b; const E* aux; uint32_t aux_stride, aux_off; E* aux_out; const TE* st_c;
};

template <class F>
__device__ __forceinline__ typename F::elem io_load(const IoDesc<F>& io, size_t pos, size_t emask) {
    typename F::elem v = io.src[(size_t)io.src_stride * pos + io.src_off];
    if (io.ld_mode == LD_SCALE) v = F::tmul(io.ld_tbl[pos & emask], v);
    return v;
}
template <class F>
__device__ __forceinline__ void io_store(const IoDesc<F>& io, size_t pos, uint32_t log_e, const typename F::elem& x) {
    using E = typename F::elem;
    const size_t emask = ((size_t)1 << log_e) - 1, i = pos & emask;
    switch (io.st_mode) {
        case ST_PLAIN: io.dst[pos] = F::canon(x); break;
        case ST_SCALE: io.dst[pos] = F::canon(F::tmul(io.st_a[i], x)); break;
        case ST_AXPBY: {
            E r = F::canon(F::tmul_add(io.st_a[i], x, F::tmul(io.st_b[i], io.aux[(size_t)io.aux_stride * pos + io.aux_off])));
            io.dst[pos] = r;
            if (io.aux_out) io.aux_out[pos] = r;
            break;
        }
        default: {  // ST_EXIT_SPLIT
            E u0 = F::canon(F::tmul(io.st_a[i], x));
            E v0 = F::canon(F::tmul(io.st_b[i], F::sub(io.aux[2 * pos], u0)));
            size_t base = (pos >> log_e) << (log_e + 1);
            io.dst[base + i] = u0;
            io.dst[base + ((size_t)1 << log_e) + i] = v0;
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
template <class F, bool DEC, int BLK = kBlockLds>
__device__ __forceinline__ void stage_sweep(typename F::elem* a_, const typename F::telem* __restrict__ ta, const typename F::telem* __restrict__ tb,
                                            uint32_t lh, uint32_t npairs, uint32_t tid) {
    using E = typename F::elem;
    const uint32_t h = 1u << lh;
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
        if (DEC) { E q1 = F::tmul(tb[i], F::sub(b, a)); a_[idx] = F::tmul_add(ta[i], q1, a); a_[idx + h] = q1; }
        else { a_[idx] = F::tmul_add(ta[i], b, a); a_[idx + h] = F::tmul_add(tb[i], b, a); }
    }
}


template <class F, int LOG_TILE_CT>      // LOG_TILE_CT > 0: tile size known at compile time (loops unroll); 0: runtime log_tile
__global__ __launch_bounds__(kBlockRow, ECFFT_MIN_WAVES) void k_stages_lds(IoDesc<F> io,
                                                           const typename F::telem* __restrict__ np0,
                                                           const typename F::telem* __restrict__ dinv,
                                                           const typename F::telem* __restrict__ p0,
                                                           const typename F::telem* __restrict__ p1,
                                                           const typename F::telem* __restrict__ inner,
                                                           uint32_t log_e, uint32_t k_first, uint32_t log_tile) {
    using E = typename F::elem;
    extern __shared__ __attribute__((aligned(16))) unsigned char ecfft_smem[];
    E* tile = reinterpret_cast<E*>(ecfft_smem);
    if (LOG_TILE_CT > 0) log_tile = LOG_TILE_CT;
    const uint32_t T = 1u << log_tile, tid = threadIdx.x;
    const size_t base = (size_t)blockIdx.x << log_tile;
    const size_t e = (size_t)1 << log_e, emask = e - 1;
#pragma unroll
    for (uint32_t j = tid; j < T; j += kBlockRow) tile[j] = io_load<F>(io, base + j, emask);
    __syncthreads();
    const uint32_t npairs = T >> 1;
    // stages k_first .. log_e-2 (h >= 2); the two innermost stages (decompose h=1, recombine h=1) act on the
    // same pairs back to back and are merged into out_j = a + c_j*(b - a): 2 multiplies instead of 4
    const uint32_t k_inner = log_e ? log_e - 1 : 0;
    for (uint32_t k = k_first; k < k_inner; ++k) {
        const uint32_t lh = log_e - k - 1, h = 1u << lh;
        stage_sweep<F, true, kBlockRow>(tile, np0 + (e - 2 * (size_t)h), dinv + (e - 2 * (size_t)h), lh, npairs, tid);
        __syncthreads();
    }
    if (log_e > 0) {
        const typename F::telem c0 = inner[0], c1 = inner[1];
#pragma unroll
        for (uint32_t g = tid; g < npairs; g += kBlockRow) {
            E a = tile[2 * g], b = tile[2 * g + 1];
            E d = F::sub(b, a);
            tile[2 * g] = F::tmul_add(c0, d, a);
            tile[2 * g + 1] = F::tmul_add(c1, d, a);
        }
        __syncthreads();
    }
    for (uint32_t k = k_inner; k-- > k_first;) {
        const uint32_t lh = log_e - k - 1, h = 1u << lh;
        stage_sweep<F, false, kBlockRow>(tile, p0 + (e - 2 * (size_t)h), p1 + (e - 2 * (size_t)h), lh, npairs, tid);
        __syncthreads();
    }
    if (io.st_mode == ST_ENTER) {
        // the tile holds whole [U1~ | V1~] blocks (2e <= T): combine them with the level's input and store interleaved
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
#pragma unroll
    for (uint32_t j = tid; j < T; j += kBlockRow) io_store<F>(io, base + j, log_e, tile[j]);
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
                                                uint32_t sft, uint32_t log_c, uint32_t log_hs, size_t c0, uint32_t npairs, uint32_t tid) {
    using E = typename F::elem;
    const uint32_t C = 1u << log_c, d = 1u << sft;
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
                                                           uint32_t log_e, uint32_t ka, uint32_t kb, uint32_t log_c) {
    using E = typename F::elem;
    extern __shared__ __attribute__((aligned(16))) unsigned char ecfft_smem[];
    E* tile = reinterpret_cast<E*>(ecfft_smem);
    const uint32_t R = kb - ka + 1, tid = threadIdx.x;
    const uint32_t C = 1u << log_c, T = LOG_TILE_CT > 0 ? (1u << LOG_TILE_CT) : (C << R);
    const size_t e = (size_t)1 << log_e, emask = e - 1;
    const uint32_t log_hs = log_e - kb - 1;
    const size_t hs = (size_t)1 << log_hs;
    const uint32_t chunks_log = log_hs - log_c;                      // column chunks per 2h_ka block
    const size_t blk = (size_t)blockIdx.x >> chunks_log, chunk = (size_t)blockIdx.x & (((size_t)1 << chunks_log) - 1);
    const size_t B = (blk << (log_hs + R)) + (chunk << log_c);       // position of (row 0, col 0)
    const size_t c0 = (chunk << log_c);                              // column offset inside the hs-block
#pragma unroll
    for (uint32_t j = tid; j < T; j += kBlockLds) {
        uint32_t r = j >> log_c, cc = j & (C - 1);
        tile[r * col_row_stride<E>(C) + cc] = io_load<F>(io, B + ((size_t)r << log_hs) + cc, emask);
    }
    __syncthreads();
    const uint32_t npairs = T >> 1;
    for (uint32_t st = 0; st < R; ++st) {
        const uint32_t k = DECOMPOSE ? ka + st : kb - st;
        const uint32_t s = kb - k, d = 1u << s;                      // row distance of the pair
        const size_t h = hs << s;
        col_stage_sweep<F, DECOMPOSE>(tile, ta + (e - 2 * h), tb + (e - 2 * h), s, log_c, log_hs, c0, npairs, tid);
        __syncthreads();
    }
#pragma unroll
    for (uint32_t j = tid; j < T; j += kBlockLds) {
        uint32_t r = j >> log_c, cc = j & (C - 1);
        io_store<F>(io, B + ((size_t)r << log_hs) + cc, log_e, tile[r * col_row_stride<E>(C) + cc]);
    }
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
                                                               uint32_t log_e, uint32_t ka, uint32_t kb, uint32_t log_c) {
    using E = typename F::elem;
    extern __shared__ __attribute__((aligned(16))) unsigned char ecfft_smem[];
    E* tile = reinterpret_cast<E*>(ecfft_smem);
    const uint32_t R = kb - ka + 1, tid = threadIdx.x;
    const uint32_t C = 1u << log_c, T = C << R;
    const size_t e = (size_t)1 << log_e, emask = e - 1;
    const uint32_t log_hs = log_e - kb - 1;
    const size_t hs = (size_t)1 << log_hs;
    const uint32_t chunks_log = log_hs - log_c;
    const size_t blk = (size_t)blockIdx.x >> chunks_log, chunk = (size_t)blockIdx.x & (((size_t)1 << chunks_log) - 1);
    const size_t B = (blk << (log_hs + R)) + (chunk << log_c);
    const size_t c0 = (chunk << log_c);
    for (uint32_t j = tid; j < T; j += kBlockLds) {
        uint32_t r = j >> log_c, cc = j & (C - 1);
        tile[r * col_row_stride<E>(C) + cc] = io.src[B + ((size_t)r << log_hs) + cc];
    }
    __syncthreads();
    const uint32_t npairs = T >> 1;
    for (uint32_t half = 0; half < 2; ++half) {
        const bool dec = half == 1;
        for (uint32_t st = 0; st < R; ++st) {
            const uint32_t k = dec ? ka + st : kb - st;
            const uint32_t sft = kb - k, d = 1u << sft;
            const size_t h = hs << sft;
            if (dec) col_stage_sweep<F, true>(tile, np0 + (e - 2 * h), dinv + (e - 2 * h), sft, log_c, log_hs, c0, npairs, tid);
            else col_stage_sweep<F, false>(tile, p0 + (e - 2 * h), p1 + (e - 2 * h), sft, log_c, log_hs, c0, npairs, tid);
            __syncthreads();
        }
        if (!dec) {
            for (uint32_t j = tid; j < T; j += kBlockLds) {
                uint32_t r = j >> log_c, cc = j & (C - 1);
                { const uint32_t q = r * col_row_stride<E>(C) + cc; tile[q] = io_mid<F>(io, B + ((size_t)r << log_hs) + cc, emask, tile[q]); }
            }
            __syncthreads();
        }
    }
    for (uint32_t j = tid; j < T; j += kBlockLds) {
        uint32_t r = j >> log_c, cc = j & (C - 1);
        io.dst[B + ((size_t)r << log_hs) + cc] = F::canon(tile[r * col_row_stride<E>(C) + cc]);
    }
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
#pragma unroll
    for (uint32_t j = tid; j < T; j += kBlockLds) {
        const uint32_t r = j >> log_c, cc = j & (C - 1);            // r < 2^(R+1): r >> R selects the vector
        tile[r * RS + cc] = work[B + ((size_t)(r >> R) << log_e) + ((size_t)(r & ((1u << R) - 1)) << log_hs) + cc];
    }
    __syncthreads();
    const uint32_t npairs = T >> 1;
    for (uint32_t st = 0; st < R; ++st) {                            // stage k = kb - st, row distance 2^st
        const size_t h = hs << st;
        col_stage_sweep<F, false>(tile, p0 + (e - 2 * h), p1 + (e - 2 * h), st, log_c, log_hs, c0, npairs, tid);
        __syncthreads();
    }
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
    E *xnn, *xnn_inv, *z0_s1, *z1_s0, *z0_inv_s1, *z1_inv_s0, *z0z0, *z1z1;
};

// every stage (decompose then recombine) of EXTEND on `len` LDS elements = len/e vectors of length e;
// srcpar = parity of the source moiety.  Ends with a barrier.
template <class F>
__device__ __forceinline__ void lds_extend_core(typename F::elem* a, uint32_t len, uint32_t log_e, const LevelTables<F>& T, int srcpar) {
    using E = typename F::elem;
    const uint32_t tid = threadIdx.x, npairs = len >> 1;
    const size_t e = (size_t)1 << log_e;
    const int tgt = 1 - srcpar;
    const uint32_t k_inner = log_e ? log_e - 1 : 0;
    for (uint32_t k = 0; k < k_inner; ++k) {
        const uint32_t lh = log_e - k - 1, h = 1u << lh;
        stage_sweep<F, true>(a, T.np0[srcpar] + (e - 2 * (size_t)h), T.dinv[srcpar] + (e - 2 * (size_t)h), lh, npairs, tid);
        __syncthreads();
    }
    if (log_e > 0) {                                    // merged innermost stage pair (h = 1)
        const typename F::telem c0 = T.inner[srcpar][0], c1 = T.inner[srcpar][1];
        for (uint32_t g = tid; g < npairs; g += kBlockLds) {
            E x = a[2 * g], y = a[2 * g + 1];
            E d = F::sub(y, x);
            a[2 * g] = F::tmul_add(c0, d, x);
            a[2 * g + 1] = F::tmul_add(c1, d, x);
        }
        __syncthreads();
    }
    for (uint32_t k = k_inner; k-- > 0;) {
        const uint32_t lh = log_e - k - 1, h = 1u << lh;
        stage_sweep<F, false>(a, T.p0[tgt] + (e - 2 * (size_t)h), T.p1[tgt] + (e - 2 * (size_t)h), lh, npairs, tid);
        __syncthreads();
    }
}

// ENTER levels 1 .. log_tile (src/fftree.rs:143-161 for every block of size <= tile).  LDS: 2*tile elements.
template <class F, int LOG_TILE>
__global__ __launch_bounds__(kBlockLds, ECFFT_MIN_WAVES) void k_enter_low(typename F::elem* __restrict__ dst, const typename F::elem* __restrict__ src,
                                                          const LevelTables<F>* __restrict__ trees) {
    using E = typename F::elem;
    extern __shared__ __attribute__((aligned(16))) unsigned char ecfft_smem[];
    constexpr uint32_t log_tile = LOG_TILE, T = 1u << LOG_TILE, npairs = T >> 1;
    constexpr int PAIRS = (int)(npairs / kBlockLds);     // compile-time trip counts keep ev/od in registers
    static_assert(PAIRS >= 1 && npairs % kBlockLds == 0, "tile too small for the workgroup");
    const uint32_t tid = threadIdx.x;
    E* cur = reinterpret_cast<E*>(ecfft_smem);
    E* work = cur + T;
    const size_t base = (size_t)blockIdx.x << log_tile;
    for (uint32_t j = tid; j < T; j += kBlockLds) cur[j] = src[base + j];
    __syncthreads();
    for (uint32_t l = 1; l <= log_tile; ++l) {
        const LevelTables<F>& L = trees[l];
        const uint32_t le = l - 1, e = 1u << le;
        for (uint32_t j = tid; j < T; j += kBlockLds) work[j] = F::tmul(L.winv[0][j & (e - 1)], cur[j]);
        __syncthreads();
        lds_extend_core<F>(work, T, le, L, 0);
        // combine (:155-159): block [u0|v0] + extended [U1|V1] -> interleaved evaluations; results are held in
        // registers across the barrier because the interleaving store overwrites other threads' inputs
        E ev[PAIRS], od[PAIRS];
#pragma unroll
        for (int c = 0; c < PAIRS; ++c) {
            uint32_t g = tid + (uint32_t)c * kBlockLds, i = g & (e - 1), bb = (g >> le) << l;
            E u0 = cur[bb + i], v0 = cur[bb + e + i], U1 = work[bb + i], V1 = work[bb + e + i];
            ev[c] = F::tmul_add(L.xe[i], v0, u0);
            od[c] = F::tmul_add(L.w1x[i], V1, F::tmul(L.w[1][i], U1));
        }
        __syncthreads();
#pragma unroll
        for (int c = 0; c < PAIRS; ++c) {
            uint32_t g = tid + (uint32_t)c * kBlockLds, i = g & (e - 1), bb = (g >> le) << l;
            cur[bb + 2 * i] = ev[c]; cur[bb + 2 * i + 1] = od[c];
        }
        __syncthreads();
    }
    for (uint32_t j = tid; j < T; j += kBlockLds) dst[base + j] = F::canon(cur[j]);
}

// EXIT levels log_tile .. 1 (src/fftree.rs:200-224 with redc_impl :232-259 inlined, normalised form, see
// DeviceChain::exit).  LDS: cur (tile) + G (tile/2) + H (tile/2).
template <class F, int LOG_TILE>
__global__ __launch_bounds__(kBlockLds, ECFFT_MIN_WAVES) void k_exit_low(typename F::elem* __restrict__ dst, const typename F::elem* __restrict__ src,
                                                         const LevelTables<F>* __restrict__ trees) {
    using E = typename F::elem;
    extern __shared__ __attribute__((aligned(16))) unsigned char ecfft_smem[];
    constexpr uint32_t log_tile = LOG_TILE, T = 1u << LOG_TILE, nh = T >> 1;
    constexpr int PAIRS = (int)(nh / kBlockLds);
    static_assert(PAIRS >= 1 && nh % kBlockLds == 0, "tile too small for the workgroup");
    const uint32_t tid = threadIdx.x;
    E* cur = reinterpret_cast<E*>(ecfft_smem);
    E* G = cur + T;
    E* H = G + nh;
    const size_t base = (size_t)blockIdx.x << log_tile;
    for (uint32_t j = tid; j < T; j += kBlockLds) cur[j] = src[base + j];
    __syncthreads();
    for (uint32_t l = log_tile; l >= 1; --l) {
        const LevelTables<F>& L = trees[l];
        const uint32_t le = l - 1, e = 1u << le;
        for (uint32_t g = tid; g < nh; g += kBlockLds) G[g] = F::tmul(L.A1[g & (e - 1)], cur[2 * g]);
        __syncthreads();
        lds_extend_core<F>(G, nh, le, L, 0);
        for (uint32_t g = tid; g < nh; g += kBlockLds) {
            uint32_t i = g & (e - 1);
            E r = F::tmul_add(L.NB2[i], G[g], F::tmul(L.B1[i], cur[2 * g + 1]));
            G[g] = r; H[g] = r;
        }
        __syncthreads();
        lds_extend_core<F>(G, nh, le, L, 1);
        for (uint32_t g = tid; g < nh; g += kBlockLds) G[g] = F::tmul(L.C1[g & (e - 1)], G[g]);
        __syncthreads();
        lds_extend_core<F>(G, nh, le, L, 0);
        for (uint32_t g = tid; g < nh; g += kBlockLds) {
            uint32_t i = g & (e - 1);
            G[g] = F::tmul_add(L.NB2[i], G[g], F::tmul(L.D1[i], H[g]));
        }
        __syncthreads();
        lds_extend_core<F>(G, nh, le, L, 1);
        E u[PAIRS], v[PAIRS];
#pragma unroll
        for (int c = 0; c < PAIRS; ++c) {
            uint32_t g = tid + (uint32_t)c * kBlockLds, i = g & (e - 1);
            u[c] = F::tmul(L.w[0][i], G[g]);
            v[c] = F::tmul(L.xie[i], F::sub(cur[2 * g], u[c]));
        }
        __syncthreads();
#pragma unroll
        for (int c = 0; c < PAIRS; ++c) {
            uint32_t g = tid + (uint32_t)c * kBlockLds, i = g & (e - 1), bb = (g >> le) << l;
            cur[bb + i] = u[c]; cur[bb + e + i] = v[c];
        }
        __syncthreads();
    }
    for (uint32_t j = tid; j < T; j += kBlockLds) dst[base + j] = F::canon(cur[j]);
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
