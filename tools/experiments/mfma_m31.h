// EXPERIMENT (round 3), NOT part of the product: see DESIGN.md section 4.3 — bit-exact, 10 % slower than the radix engine it replaced.
// The innermost butterfly stages of the Mersenne-31 EXTEND on the int8 matrix cores (gfx950 v_mfma_i32_32x32x32_i8).
//
// Same idea as mfma_blk16.h (secp256k1), and a much better fit: an M31 element is 4 bytes, so a constant c is a 4 x 4 int8 matrix
// (digit b of c * 2^(8j) mod p — and 2^31 = 1 mod p makes c * 2^(8j) a 31-bit ROTATION of c) and a whole B-point linear map is a
// 4B x 4B int8 matrix.  For B = 64 the stages with pair distance h = 32 .. 2 (decompose), the merged innermost pair and h = 2 .. 32
// (recombine) — 11 of the 25 sweeps of an 8192-element tile (src/fftree.rs:83-118 for the last six levels of the recursion;
// arithmetic of src/utils.rs:338-347) — are ONE 256 x 256 int8 matrix per (tree, source parity): 64 KiB, shared by every
// 64-block of the level.  A tile holds 128 blocks = 4 batches of 32 (N = 32 columns of an MFMA); wave w owns the 32 rows
// 32w .. 32w+31 of the matrix (8 outputs x 4 digits), keeps its 8 k-slices (8 KiB) in registers for all four batches, and
// accumulates 8 MFMAs per batch: 256 MFMAs per tile instead of 11 x 8192 six-instruction multiplies.  No cross-lane step at all:
//   B operand: lane (n, h) holds 16 bytes = input elements 8ks + 4h .. +3 of block n — one 16-byte LDS read of 4 consecutive elements;
//   D:         lane (n, h) holds rows (r & 3) + 8 (r >> 2) + 4h; row m carries digit m & 3 of output 4 ((m >> 2) & 1) + (m >> 3),
//              so the lane's 16 accumulators are the 4 digits of the 4 CONSECUTIVE outputs 8w + 4h .. +3 of block n — one 16-byte
//              LDS write after a ~10-instruction normalisation per element (y0 + y1 2^8 + y2 2^16 + y3 2^24 + seed, two folds).
// Data bytes 0..2 are unsigned (x - 128 = xor 0x80 goes into the MFMA, 128 * 0x010101 * sum_i T[o][i] is part of the output's seed),
// byte 3 is < 128 already; constants are recoded to signed digits (value c or c - p).  Inputs may be in the kernels' lazy range
// [0, p]; results are in [0, p] too (the engine's contract, field_m31.h), so the map is bit-identical after canon().
// LDS: the 16-byte chunks (quads) are swizzled by the block index while the tile is in operand form — the operand reads take the
// same quad of 32 blocks 256 B apart (one bank) — q' = q ^ ((q >> 4) & 15).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "field_m31.h"

namespace ecfft {

struct M31Blk64 {
    using F = M31;
    static constexpr int NB = 64;                                   // points of the composite map
    static constexpr size_t kABytes = (size_t)(4 * NB) * (4 * NB);  // 256 x 256 int8
    static constexpr size_t kKWords = NB;                           // one 64-bit seed per output
    static constexpr size_t kArenaElems = (kABytes + kKWords * 8) / sizeof(uint32_t);
    static constexpr uint32_t kXor = 0x00808080u;
    static constexpr unsigned long long kOff = (unsigned long long)0x7FFFFFFFull << 17;   // multiple of p above every negative column sum

    typedef int v4i __attribute__((ext_vector_type(4)));
    typedef int v16i __attribute__((ext_vector_type(16)));

    __host__ __device__ static inline uint32_t phys_quad(uint32_t q) { return q ^ ((q >> 4) & 15u); }   // q = element index / 4

    // plain tile of `len` elements (thread tid owns elements [tid*EPT, +EPT) for the reads) -> operand form.  Ends with a barrier.
    template <int EPT>
    __device__ static __forceinline__ void to_operand_form(uint32_t* a, uint32_t tid) {
        uint4* q = reinterpret_cast<uint4*>(a);
        uint4 v[EPT / 4];
#pragma unroll
        for (int c = 0; c < EPT / 4; ++c) v[c] = q[tid * (EPT / 4) + c];
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");          // LDS-only barriers: the matrix requests stay in flight
#pragma unroll
        for (int c = 0; c < EPT / 4; ++c) q[phys_quad(tid * (EPT / 4) + c)] = make_uint4(v[c].x ^ kXor, v[c].y ^ kXor, v[c].z ^ kXor, v[c].w ^ kXor);
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    }
    // results (plain bytes, lazy range) at swizzled quad positions -> ordinary layout.  Ends with a barrier.
    template <int EPT>
    __device__ static __forceinline__ void from_swizzled(uint32_t* a, uint32_t tid) {
        uint4* q = reinterpret_cast<uint4*>(a);
        uint4 v[EPT / 4];
#pragma unroll
        for (int c = 0; c < EPT / 4; ++c) v[c] = q[phys_quad(tid * (EPT / 4) + c)];
        __syncthreads();
#pragma unroll
        for (int c = 0; c < EPT / 4; ++c) q[tid * (EPT / 4) + c] = v[c];
        __syncthreads();
    }

    __device__ static __forceinline__ uint32_t normalise(int y0, int y1, int y2, int y3, unsigned long long seed) {
        const int a = y0 + (y1 << 8), b = y2 + (y3 << 8);                  // |.| < 2^31: column sums are < 2^22 in magnitude
        const long long v = (long long)seed + (long long)a + (long long)b * 65536ll;      // in [0, 2^49)
        const uint32_t lo = (uint32_t)v & F::P, hi = (uint32_t)((unsigned long long)v >> 31);
        const uint32_t r = lo + hi;                                        // < 2^31 + 2^18
        return (r & F::P) + (r >> 31);                                     // in [0, p]
    }

    // the map on every 64-block of `len` = BLK * EPT elements held in operand form (8 waves); results at swizzled positions.
    // Ends with a barrier.
    struct ARegs { v4i a[8]; };
    // this wave's 32 rows of the matrix, all 256 columns (8 KiB): requested BEFORE the conversion to operand form (two barriers and an
    // LDS round trip for the L2 latency to hide behind); follow the call with __builtin_amdgcn_sched_barrier(0)
    __device__ static __forceinline__ ARegs load_a(const uint8_t* __restrict__ Amat, uint32_t tid) {
        typedef const __attribute__((address_space(1))) char* gchar;
        typedef const __attribute__((address_space(1))) v4i* gv4;
        const uint32_t L = tid & 63, w = (uint32_t)__builtin_amdgcn_readfirstlane((int)(tid >> 6));
        const gchar ap = (gchar)(reinterpret_cast<const char*>(Amat)) + ((size_t)w * 8) * 1024 + L * 16;
        ARegs r;
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) r.a[ks] = *(gv4)(ap + (size_t)ks * 1024);
        return r;
    }
    template <int NBATCH>
    __device__ static __forceinline__ void phase(uint32_t* a, const ARegs& AR, const unsigned long long* __restrict__ Kc, uint32_t tid) {
        uint4* lds = reinterpret_cast<uint4*>(a);
        const uint32_t L = tid & 63, w = (uint32_t)__builtin_amdgcn_readfirstlane((int)(tid >> 6)), n = L & 31, h = L >> 5;
        const v4i* A = AR.a;
        unsigned long long seed[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) seed[k] = Kc[8 * w + 4 * h + k];
        uint4 res[NBATCH];
#pragma unroll
        for (int bt = 0; bt < NBATCH; ++bt) {
            const uint32_t blk = 32u * bt + n;
            v16i acc = {0};
#pragma unroll
            for (int ks = 0; ks < 8; ++ks) {
                const uint4 b = lds[phys_quad(blk * 16u + 2u * ks + h)];
                const v4i B = {(int)b.x, (int)b.y, (int)b.z, (int)b.w};
                acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(A[ks], B, acc, 0, 0, 0);
            }
            res[bt] = make_uint4(normalise(acc[0], acc[1], acc[2], acc[3], seed[0]), normalise(acc[4], acc[5], acc[6], acc[7], seed[1]),
                                 normalise(acc[8], acc[9], acc[10], acc[11], seed[2]), normalise(acc[12], acc[13], acc[14], acc[15], seed[3]));
        }
        __syncthreads();                       // every operand read is done: the tile can be overwritten
#pragma unroll
        for (int bt = 0; bt < NBATCH; ++bt) lds[phys_quad((32u * bt + n) * 16u + 2u * w + h)] = res[bt];   // outputs 8w + 4h .. +3 of the block
        __syncthreads();
    }
};

// Construction: the 64 x 64 matrix of the tree (the engine's own stage arithmetic applied to the unit vectors), its int8 expansion
// and the seeds.  np0 / dinv: decompose tables of the source parity, p0 / p1: recombine tables of the target parity, inner: merged
// innermost pair (doubled-residue table form, as the kernels read them); e = vector length of the tree (>= 64).  One workgroup of 256.
__global__ __launch_bounds__(256) void k_m31_blk64_build(const uint32_t* __restrict__ np0, const uint32_t* __restrict__ dinv, const uint32_t* __restrict__ p0,
                                                         const uint32_t* __restrict__ p1, const uint32_t* __restrict__ inner, size_t e,
                                                         uint8_t* __restrict__ Amat, unsigned long long* __restrict__ Kc) {
    using F = M31;
    __shared__ uint32_t Tm[64 * 64];                                 // [output][input]
    const uint32_t tid = threadIdx.x;
    if (tid < 64) {
        uint32_t x[64];
        for (int k = 0; k < 64; ++k) x[k] = (k == (int)tid) ? 1u : 0u;
        for (int lh = 5; lh >= 1; --lh) {
            const uint32_t hh = 1u << lh; const size_t off = e - 2 * (size_t)hh;
            for (uint32_t g = 0; g < 32; ++g) {
                const uint32_t i = g & (hh - 1), idx = ((g >> lh) << (lh + 1)) + i;
                const uint32_t a = x[idx], b = x[idx + hh];
                const uint32_t q1 = F::tmul(dinv[off + i], F::sub(b, a));
                x[idx] = F::tmul_add(np0[off + i], q1, a); x[idx + hh] = q1;
            }
        }
        {
            const uint32_t c0 = inner[0], c1 = inner[1];
            for (uint32_t g = 0; g < 32; ++g) { const uint32_t a = x[2 * g], d = F::sub(x[2 * g + 1], a); x[2 * g] = F::tmul_add(c0, d, a); x[2 * g + 1] = F::tmul_add(c1, d, a); }
        }
        for (int lh = 1; lh <= 5; ++lh) {
            const uint32_t hh = 1u << lh; const size_t off = e - 2 * (size_t)hh;
            for (uint32_t g = 0; g < 32; ++g) {
                const uint32_t i = g & (hh - 1), idx = ((g >> lh) << (lh + 1)) + i;
                const uint32_t a = x[idx], b = x[idx + hh];
                x[idx] = F::tmul_add(p0[off + i], b, a); x[idx + hh] = F::tmul_add(p1[off + i], b, a);
            }
        }
        for (int o = 0; o < 64; ++o) Tm[o * 64 + tid] = F::canon(x[o]);  // column tid of the map
    }
    __syncthreads();
    for (uint32_t e2 = tid; e2 < 64 * 64; e2 += 256) {
        const uint32_t o = e2 >> 6, i = e2 & 63, c = Tm[e2];
        const uint32_t rt = o >> 3, oo = o & 7, ks = i >> 3, ii = i & 7;
        for (uint32_t j = 0; j < 4; ++j) {
            uint32_t cj = j ? (((c << (8 * j)) | (c >> (31 - 8 * j))) & F::P) : c;       // c * 2^(8j) mod p: 31-bit rotation
            if (cj == F::P) cj = 0;
            const uint32_t pat = cj > 0x7F7F7F7Fu ? cj - F::P : cj;                       // value cj or cj - p in [-0x80808080, 0x7f7f7f7f]
            const uint32_t z = pat + 0x80808080u;
            const uint32_t kbyte = 4 * ii + j, hA = kbyte >> 4, q = kbyte & 15;
            for (uint32_t b = 0; b < 4; ++b) {
                const uint32_t m = b + 4 * (oo >> 2) + 8 * (oo & 3);                       // row of (output oo, digit b) inside the wave's 32
                Amat[(((size_t)rt * 8 + ks) * 64 + (m + 32 * hA)) * 16 + q] = (uint8_t)(((z >> (8 * b)) & 0xffu) ^ 0x80u);
            }
        }
    }
    if (tid < 64) {
        uint32_t s = 0;
        for (int i = 0; i < 64; ++i) s = F::add(s, Tm[tid * 64 + i]);
        const uint32_t bias = F::mul(s, 0x00808080u % F::P);                               // 128 * (1 + 2^8 + 2^16) * sum_i T[o][i]
        Kc[tid] = M31Blk64::kOff + bias;
    }
}

}  // namespace ecfft
