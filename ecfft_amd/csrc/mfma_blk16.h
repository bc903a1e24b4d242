// The innermost butterfly stages of the secp256k1 EXTEND on the int8 matrix cores (gfx950 v_mfma_i32_32x32x32_i8).
//
// Replaces, inside an LDS tile, the arithmetic of /root/reference/src/utils.rs:338-347 (Mat2x2 * [F;2]) for the stages whose
// constants are shared by >= 32 butterflies of the tile: the decompose stages with pair distance h = 8, 4, 2, the merged
// innermost pair (h = 1) and the recombine stages h = 2, 4, 8 (src/fftree.rs:83-118 for the last four levels of the recursion).
// On every aligned block of 16 points those 7 sweeps are ONE linear map  out_o = sum_i T[o][i] * x_i  (mod p)  with a
// 16 x 16 matrix of field constants that depends only on the tree and the source parity (stage tables are indexed by
// position mod h).  Every multiply on the path is data x constant, so each constant c = T[o][i] can be
// stored as the 32 x 32 int8 matrix
//        C[b][j] = digit b (radix 256, SIGNED, in [-128, 127]) of  c * 2^(8j) mod p          (j = data byte, b = result digit)
// and then, for data x = sum_j x_j 2^(8j),     c * x  ==  sum_b 2^(8b) * sum_j C[b][j] * x_j      (mod p):
// a 32x32x32 int8 contraction per constant and per 32 data elements.  One MFMA takes the same constant for the same
// position of 32 different blocks (N = 32), 16 MFMAs accumulate the 16 inputs of a block into one 32 x 32 int32 tile, and
// what is left for the integer VALU is ONE carry normalisation per output element (32 column sums < 2^24 -> eight 32-bit words,
// fold of the 19-bit top by 2^256 = 2^32 + 977) instead of seven 169-instruction modular multiplies.
// Data bytes are unsigned; the MFMA is signed x signed: x_j - 128 (xor 0x80) goes in, and 128 * sum_j C[b][j] is part of
// the per-output constants K.  Results are canonical residues, so the re-association is bit-exact (tools/ubench/mfma_mul.hip
// checks the scheme against host arithmetic; profiles/r03/ubench_mfma_mul*.txt has the measured rates).
//
// Layouts (fixed by the instruction, cdna_hip_programming.md section 3):
//   A operand (constants): lane l holds row m = l & 31, bytes k = 16*(l >> 5) .. +15;   B operand (data): lane l holds
//   column n = l & 31 (= block), the same k range = bytes 16*(l >> 5).. of the element;   D: lane l holds column l & 31,
//   rows (r & 3) + 8*(r >> 2) + 4*(l >> 5), r = register 0..15.  Row m carries result digit b(m) = 16*((m >> 2) & 1) + (m & 3)
//   + 4*(m >> 3), so that after one v_permlane32_swap per register a lane owns all 32 digits of ONE element: D1[r] = digit r,
//   D2[r] = digit 16 + r.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "field_secp256k1.h"

#ifndef BLK16_STAMP
#define BLK16_STAMP(k)
#endif

namespace ecfft {

struct Blk16 {
    using F = Secp256k1;
    using E = Fe256;
    using TE = Te256;
    static constexpr int NB = 16;                          // points of the composite map
    static constexpr size_t kABytes = (size_t)NB * NB * 1024;   // 256 int8 matrices
    static constexpr size_t kKWords = (size_t)NB * 8;      // per output: eight 64-bit accumulator seeds
    static constexpr size_t kArenaElems = (kABytes + kKWords * 8) / sizeof(E);   // per (tree, parity), in field elements
    static constexpr int kSub = 1024;                      // elements per MFMA phase (32 blocks x 2 batches): 512 threads

    typedef int v4i __attribute__((ext_vector_type(4)));
    typedef int v16i __attribute__((ext_vector_type(16)));

    // 16-byte chunk q = 2*j + half of element j -> physical chunk.  The operand reads take the SAME half of 32 elements 16
    // apart (512 B stride: one bank); xor-ing the chunk's low 4 bits with the block index spreads them over all banks.
    __host__ __device__ static inline uint32_t phys(uint32_t j, uint32_t hh) {
        const uint32_t q = 2 * j + hh;
        return (q & ~15u) | ((q ^ (j >> 4)) & 15u);
    }
    // row of the constant matrix that carries result digit b (inverse of b(m) above)
    __host__ __device__ static inline uint32_t row_of_digit(uint32_t b) {
        const uint32_t r = b & 15u;
        return (r & 3u) | ((b >> 4) << 2) | ((r >> 2) << 3);
    }

    // 32 signed column sums (digit b = lo[b] for b < 16, hi[b - 16] above) + the output's seeds K -> canonical residue
    // KV = false: K is wave-uniform (scalar loads, SGPR operand); true: K differs per lane (vector loads)
    template <bool KV = false>
    __device__ static __forceinline__ E normalise(const int (&lo)[16], const int (&hi)[16], const unsigned long long* __restrict__ K) {
        long long W[8];
        const uint32_t s8 = 1u << 8, s16 = 1u << 16, s24 = 1u << 24;
#pragma unroll
        for (int g = 0; g < 8; ++g) {
            const int* y = g < 4 ? &lo[4 * g] : &hi[4 * (g - 4)];
            const unsigned long long kg = K[g];
            long long w;
            if constexpr (KV)
                asm("v_mad_i64_i32 %0, vcc, %1, 1, %5\n\t"
                    "v_mad_i64_i32 %0, vcc, %2, %6, %0\n\t"
                    "v_mad_i64_i32 %0, vcc, %3, %7, %0\n\t"
                    "v_mad_i64_i32 %0, vcc, %4, %8, %0"
                    : "=&v"(w) : "v"(y[0]), "v"(y[1]), "v"(y[2]), "v"(y[3]), "v"(kg), "s"(s8), "s"(s16), "s"(s24) : "vcc");
            else
                asm("v_mad_i64_i32 %0, vcc, %1, 1, %5\n\t"
                    "v_mad_i64_i32 %0, vcc, %2, %6, %0\n\t"
                    "v_mad_i64_i32 %0, vcc, %3, %7, %0\n\t"
                    "v_mad_i64_i32 %0, vcc, %4, %8, %0"
                    : "=&v"(w) : "v"(y[0]), "v"(y[1]), "v"(y[2]), "v"(y[3]), "s"(kg), "s"(s8), "s"(s16), "s"(s24) : "vcc");
            W[g] = w;                                      // in [2^50 - 2^48, 2^50 + 2^48 + 2^32)
        }
        uint32_t z0 = (uint32_t)W[0], z1, z2, z3, z4, z5, z6, z7, z8, t, c8;
        uint32_t r0, r1, r2, r3, r4, r5, r6, r7;
        const uint32_t k977 = 977u;
        // words of sum_g W_g 2^(32g) (z8 < 2^19 + 1), then the fold z8 * (2^32 + 977) onto words 0, 1 and its ripple
        asm("v_add_co_u32_e32 %0, vcc, %18, %19\n\t"
            "v_addc_co_u32_e32 %1, vcc, %20, %21, vcc\n\t"
            "v_addc_co_u32_e32 %2, vcc, %22, %23, vcc\n\t"
            "v_addc_co_u32_e32 %3, vcc, %24, %25, vcc\n\t"
            "v_addc_co_u32_e32 %4, vcc, %26, %27, vcc\n\t"
            "v_addc_co_u32_e32 %5, vcc, %28, %29, vcc\n\t"
            "v_addc_co_u32_e32 %6, vcc, %30, %31, vcc\n\t"
            "v_addc_co_u32_e32 %7, vcc, 0, %32, vcc\n\t"
            "v_mul_u32_u24_e32 %8, %34, %7\n\t"
            "v_add_co_u32_e32 %9, vcc, %33, %8\n\t"
            "v_addc_co_u32_e32 %10, vcc, %0, %7, vcc\n\t"
            "v_addc_co_u32_e32 %11, vcc, 0, %1, vcc\n\t"
            "v_addc_co_u32_e32 %12, vcc, 0, %2, vcc\n\t"
            "v_addc_co_u32_e32 %13, vcc, 0, %3, vcc\n\t"
            "v_addc_co_u32_e32 %14, vcc, 0, %4, vcc\n\t"
            "v_addc_co_u32_e32 %15, vcc, 0, %5, vcc\n\t"
            "v_addc_co_u32_e32 %16, vcc, 0, %6, vcc\n\t"
            "v_addc_co_u32_e64 %17, vcc, 0, 0, vcc"
            : "=&v"(z1), "=&v"(z2), "=&v"(z3), "=&v"(z4), "=&v"(z5), "=&v"(z6), "=&v"(z7), "=&v"(z8), "=&v"(t),
              "=&v"(r0), "=&v"(r1), "=&v"(r2), "=&v"(r3), "=&v"(r4), "=&v"(r5), "=&v"(r6), "=&v"(r7), "=&v"(c8)
            : "v"((uint32_t)((unsigned long long)W[0] >> 32)), "v"((uint32_t)W[1]),
              "v"((uint32_t)((unsigned long long)W[1] >> 32)), "v"((uint32_t)W[2]),
              "v"((uint32_t)((unsigned long long)W[2] >> 32)), "v"((uint32_t)W[3]),
              "v"((uint32_t)((unsigned long long)W[3] >> 32)), "v"((uint32_t)W[4]),
              "v"((uint32_t)((unsigned long long)W[4] >> 32)), "v"((uint32_t)W[5]),
              "v"((uint32_t)((unsigned long long)W[5] >> 32)), "v"((uint32_t)W[6]),
              "v"((uint32_t)((unsigned long long)W[6] >> 32)), "v"((uint32_t)W[7]),
              "v"((uint32_t)((unsigned long long)W[7] >> 32)), "v"(z0), "s"(k977)
            : "vcc");
        E r; r.l[0] = r0; r.l[1] = r1; r.l[2] = r2; r.l[3] = r3; r.l[4] = r4; r.l[5] = r5; r.l[6] = r6; r.l[7] = r7;
        // a carry out of word 7 (needs words 2..7 all ones) or a value in [p, 2^256) (needs word 7 all ones): ~2^-32
        if (__builtin_expect(c8 | (uint32_t)(r7 == 0xFFFFFFFFu), 0)) {
            uint32_t sv[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) sv[i] = r.l[i];
            r = F::finish(sv, c8);
        }
        return r;
    }

    // plain tile (T elements, T a multiple of 512) -> operand form: bytes xor 0x80, chunks swizzled.  Ends with a barrier.
    // SWZ: the tile comes in the XOR-swizzled layout of kernels.h (lds_get / lds_put) instead of the plain one
    template <int BLK, bool SWZ = false>
    __device__ static __forceinline__ void to_operand_form(E* tile, uint32_t T, uint32_t tid) {
        uint4* lds = reinterpret_cast<uint4*>(tile);
#pragma unroll 1
        for (uint32_t base = 0; base < T; base += 2 * BLK) {       // 2 elements per thread per round
            const uint32_t j0 = base + tid, j1 = base + BLK + tid;
            const bool two = j1 < T;
            const uint32_t s0 = SWZ ? ((j0 >> 3) & 3u) : 0u, s1 = SWZ ? ((j1 >> 3) & 3u) : 0u;
            uint4 a0 = lds[(2 * j0) ^ s0], a1 = lds[(2 * j0 + 1) ^ s0], b0 = make_uint4(0, 0, 0, 0), b1 = b0;
            if (two) { b0 = lds[(2 * j1) ^ s1]; b1 = lds[(2 * j1 + 1) ^ s1]; }
            __syncthreads();
            const uint32_t X = 0x80808080u;
            lds[phys(j0, 0)] = make_uint4(a0.x ^ X, a0.y ^ X, a0.z ^ X, a0.w ^ X);
            lds[phys(j0, 1)] = make_uint4(a1.x ^ X, a1.y ^ X, a1.z ^ X, a1.w ^ X);
            if (two) { lds[phys(j1, 0)] = make_uint4(b0.x ^ X, b0.y ^ X, b0.z ^ X, b0.w ^ X); lds[phys(j1, 1)] = make_uint4(b1.x ^ X, b1.y ^ X, b1.z ^ X, b1.w ^ X); }
        }
        __syncthreads();
    }

    // one element -> operand form (bytes xor 0x80, swizzled chunks) at position j of an LDS array
    __device__ static __forceinline__ void store_operand(E* arr, uint32_t j, const E& x) {
        uint4* lds = reinterpret_cast<uint4*>(arr);
        const uint32_t X = 0x80808080u;
        lds[phys(j, 0)] = make_uint4(x.l[0] ^ X, x.l[1] ^ X, x.l[2] ^ X, x.l[3] ^ X);
        lds[phys(j, 1)] = make_uint4(x.l[4] ^ X, x.l[5] ^ X, x.l[6] ^ X, x.l[7] ^ X);
    }
    // The constant matrices of inputs 0 and 1 for the wave's two outputs, requested EARLY (they do not depend on the data): the
    // caller issues this before the sweep / barriers that produce the operand form, so the L2 latency of the first requests is
    // hidden; follow the call with __builtin_amdgcn_sched_barrier(0) or the scheduler sinks the loads to their first use.
    struct APre { v4i a[2][2]; };                           // [input][output]
    typedef const __attribute__((address_space(1))) char* gchar;
    typedef const __attribute__((address_space(1))) v4i* gv4;
    __device__ static __forceinline__ gchar a_base(const uint8_t* __restrict__ Amat, uint32_t tid) {
        const uint32_t L = tid & 63, w = (uint32_t)__builtin_amdgcn_readfirstlane((int)(tid >> 6));
        return (gchar)(reinterpret_cast<const char*>(Amat)) + ((size_t)(2 * w) * NB) * 1024 + L * 16;
    }
    __device__ static __forceinline__ v4i ldA(gchar ap, int oo, int i) { return *(gv4)(ap + ((size_t)oo * NB + (size_t)i) * 1024); }
    __device__ static __forceinline__ APre prefetch(const uint8_t* __restrict__ Amat, uint32_t tid) {
        const gchar ap = a_base(Amat, tid);
        APre p; p.a[0][0] = ldA(ap, 0, 0); p.a[0][1] = ldA(ap, 1, 0); p.a[1][0] = ldA(ap, 0, 1); p.a[1][1] = ldA(ap, 1, 1);
        return p;
    }

    // The map on the 64 blocks of one 1024-element sub-tile held in operand form; results are canonical residues (plain bytes)
    // left at the SWIZZLED chunk positions (load_swizzled / from_swizzled).  512 threads: wave w produces outputs o = 2w, 2w+1 of all 64 blocks — four
    // 32 x 32 accumulators, so every data operand is read from LDS once per wave; the constant matrices run two inputs ahead
    // of the MFMAs (L2 latency), the data operands one.  Ends with a barrier.
    __device__ static __forceinline__ void phase(E* sub, const uint8_t* __restrict__ Amat, const unsigned long long* __restrict__ Kc, uint32_t tid, const APre& pre) {
        uint4* lds = reinterpret_cast<uint4*>(sub);
        const uint32_t L = tid & 63, w = (uint32_t)__builtin_amdgcn_readfirstlane((int)(tid >> 6)), n = L & 31, h = L >> 5;
        const gchar ap = a_base(Amat, tid);
        // operand address of (block, input i): phys() reduces to a0 ^ 32*i bytes (2i | h never carries), batch 1 is 16 KiB above
        const uint32_t a0 = 16u * phys(n * NB, h);
        const char* lb = reinterpret_cast<const char*>(lds);
        auto ldB = [&](int r, int i) { const uint4 b = *reinterpret_cast<const uint4*>(lb + ((a0 ^ (32u * (uint32_t)i)) + 16384u * (uint32_t)r)); v4i v = {(int)b.x, (int)b.y, (int)b.z, (int)b.w}; return v; };
        v16i acc00 = {0}, acc01 = {0}, acc10 = {0}, acc11 = {0};           // acc[output][batch]
        v4i A0 = pre.a[0][0], A1 = pre.a[0][1], A0b = pre.a[1][0], A1b = pre.a[1][1];
        BLK16_STAMP(1)
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            v4i nA0 = A0b, nA1 = A1b;
            if (i + 2 < NB) { nA0 = ldA(ap, 0, i + 2); nA1 = ldA(ap, 1, i + 2); }
            const v4i B0 = ldB(0, i), B1 = ldB(1, i);   // LDS latency is covered by the SIMD's other wave; 128 VGPRs leave no room to run ahead
            __builtin_amdgcn_sched_barrier(0);          // keep the requests ABOVE the MFMAs (the scheduler sinks them to their uses)
            acc00 = __builtin_amdgcn_mfma_i32_32x32x32_i8(A0, B0, acc00, 0, 0, 0);
            acc01 = __builtin_amdgcn_mfma_i32_32x32x32_i8(A0, B1, acc01, 0, 0, 0);
            acc10 = __builtin_amdgcn_mfma_i32_32x32x32_i8(A1, B0, acc10, 0, 0, 0);
            acc11 = __builtin_amdgcn_mfma_i32_32x32x32_i8(A1, B1, acc11, 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            A0 = A0b; A1 = A1b; A0b = nA0; A1b = nA1;
        }
        BLK16_STAMP(2)
        E outz[2];
#pragma unroll
        for (int oo = 0; oo < 2; ++oo) {
            int lo[16], hi[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                auto p = __builtin_amdgcn_permlane32_swap((unsigned)(oo ? acc10[r] : acc00[r]), (unsigned)(oo ? acc11[r] : acc01[r]), false, false);
                lo[r] = (int)p[0]; hi[r] = (int)p[1];
            }
            outz[oo] = normalise(lo, hi, Kc + (2 * w + oo) * 8);
        }
        BLK16_STAMP(3)
        __syncthreads();        // every operand read of this phase is done: the tile can be overwritten
        BLK16_STAMP(4)
#pragma unroll
        for (int oo = 0; oo < 2; ++oo) {
            const uint32_t j = L * NB + 2 * w + oo;           // lane L holds block L (batch L >> 5, column L & 31)
            const E& z = outz[oo];
            // SWIZZLED chunk positions (plain bytes): a wave's 64 results are 512 B apart, one bank in the plain layout
            lds[phys(j, 0)] = make_uint4(z.l[0], z.l[1], z.l[2], z.l[3]);
            lds[phys(j, 1)] = make_uint4(z.l[4], z.l[5], z.l[6], z.l[7]);
        }
        __syncthreads();
    }
    // element j of an array whose chunks sit at their swizzled positions (what phase() leaves behind)
    __device__ static __forceinline__ E load_swizzled(const E* arr, uint32_t j) {
        const uint4* lds = reinterpret_cast<const uint4*>(arr);
        const uint4 a = lds[phys(j, 0)], b = lds[phys(j, 1)];
        E r; r.l[0] = a.x; r.l[1] = a.y; r.l[2] = a.z; r.l[3] = a.w; r.l[4] = b.x; r.l[5] = b.y; r.l[6] = b.z; r.l[7] = b.w;
        return r;
    }
    // swizzled chunk positions -> ordinary layout, T elements (a multiple of 512).  Ends with a barrier.
    template <int BLK>
    __device__ static __forceinline__ void from_swizzled(E* tile, uint32_t T, uint32_t tid) {
#pragma unroll 1
        for (uint32_t base = 0; base < T; base += 2 * BLK) {
            const uint32_t j0 = base + tid, j1 = base + BLK + tid;
            const bool two = j1 < T;
            const E a = load_swizzled(tile, j0); E b = a;
            if (two) b = load_swizzled(tile, j1);
            __syncthreads();
            tile[j0] = a; if (two) tile[j1] = b;
        }
        __syncthreads();
    }

    // The same map on a 512-element array (32 blocks: ONE batch, k_exit_low's half-tiles) whose element tid the calling thread
    // holds in registers (x, plain).  Wave w produces outputs 2w and 2w+1 of the 32 blocks; the register swap pairs the two
    // OUTPUTS instead of two batches: afterwards lanes 0..31 own output 2w of block lane, lanes 32..63 output 2w+1 of block
    // lane - 32, and the accumulator seeds differ per lane half.  The constant matrices of the first four inputs are requested
    // before the barrier that lets the array be overwritten (the tables of a low-level kernel's many trees are L2 misses as often
    // as not).  Returns element tid of the result; ends with a barrier.
    __device__ static __forceinline__ E phase512_regs(E* sub, const E& x, const uint8_t* __restrict__ Amat, const unsigned long long* __restrict__ Kc, uint32_t tid) {
        uint4* lds = reinterpret_cast<uint4*>(sub);
        const uint32_t L = tid & 63, w = (uint32_t)__builtin_amdgcn_readfirstlane((int)(tid >> 6)), n = L & 31, h = L >> 5;
        const gchar ap = a_base(Amat, tid);
        constexpr int D = 4;                                            // inputs in flight
        v4i QA0[D], QA1[D];
#pragma unroll
        for (int d = 0; d < D; ++d) { QA0[d] = ldA(ap, 0, d); QA1[d] = ldA(ap, 1, d); }
        __builtin_amdgcn_sched_barrier(0);
        __syncthreads();                                                // readers of the array's previous contents are done
        store_operand(sub, tid, x);
        __syncthreads();
        const uint32_t a0 = 16u * phys(n * NB, h);
        const char* lb = reinterpret_cast<const char*>(lds);
        auto ldB = [&](int i) { const uint4 b = *reinterpret_cast<const uint4*>(lb + (a0 ^ (32u * (uint32_t)i))); v4i v = {(int)b.x, (int)b.y, (int)b.z, (int)b.w}; return v; };
        v16i acc0 = {0}, acc1 = {0};                                     // acc[output]
        v4i B0 = ldB(0);
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            const v4i A0 = QA0[i % D], A1 = QA1[i % D];
            v4i nB0 = B0;
            if (i + D < NB) { QA0[i % D] = ldA(ap, 0, i + D); QA1[i % D] = ldA(ap, 1, i + D); }
            if (i + 1 < NB) nB0 = ldB(i + 1);
            __builtin_amdgcn_sched_barrier(0);
            acc0 = __builtin_amdgcn_mfma_i32_32x32x32_i8(A0, B0, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_i32_32x32x32_i8(A1, B0, acc1, 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            B0 = nB0;
        }
        int lo[16], hi[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            auto p = __builtin_amdgcn_permlane32_swap((unsigned)acc0[r], (unsigned)acc1[r], false, false);
            lo[r] = (int)p[0]; hi[r] = (int)p[1];
        }
        const E z = normalise<true>(lo, hi, Kc + (2 * w + h) * 8);
        __syncthreads();        // every operand read of this phase is done
        {
            const uint32_t j = n * NB + 2 * w + h;
            lds[phys(j, 0)] = make_uint4(z.l[0], z.l[1], z.l[2], z.l[3]);      // swizzled positions: see phase()
            lds[phys(j, 1)] = make_uint4(z.l[4], z.l[5], z.l[6], z.l[7]);
        }
        __syncthreads();
        return load_swizzled(sub, tid);
    }

    // ---------------------------------------------------------------------------------------------------------------------
    // The SMALL-LAUNCH form (round 4): the same map on a 256-element array = 16 blocks (or a 128-element array = 8 blocks), with
    // v_mfma_i32_16x16x64_i8 (N = 16 columns = the blocks; K = 64 = the bytes of TWO inputs; M = 16 = half of an output's 32
    // digit rows).  Launches with fewer 1024-element tiles than CUs run on 256-element tiles, which hold half of
    // a 32 x 32 x 32 MFMA's columns.  Same constant tables as phase(): row group g (0 / 1) of output o and K-step s (inputs 2s,
    // 2s + 1) is rows 16g .. 16g + 15 of the 32 x 32 matrices (o, 2s) and (o, 2s + 1), i.e. lane l = (q, m) (q = l >> 4,
    // m = l & 15) reads the 16 bytes that lane 16g + m + 32 (q & 1) of matrix (o, 2s + (q >> 1)) holds in the 32 x 32 x 32 layout.
    //   A: lane (q, m): row m, K bytes 16q..;  B: lane (q, n): column n = block, K bytes 16q.. = chunk q & 1 of input 2s + (q >> 1);
    //   D: lane (q, n): column n, rows 4q + r (r = register 0..3)  ->  digit 16 (q & 1) + 8g + 4 (q >> 1) + r of the output.
    // NW waves (4 or 2); wave w produces the OPW = 16 / NW outputs OPW w .. OPW w + OPW - 1 in sets of four (eight 16 x 16
    // accumulators per set).  After the MFMAs the four lane quarters hold different digits of the SAME four outputs: a 4 x 4
    // transpose across the quarters (v_permlane32_swap, then v_permlane16_swap, per accumulator register) gives lane (q, n) all
    // 32 digits of output (4 set + q) of block n.
    //   NW = 4            : 256 elements, 256 threads, one result per lane
    //   NW = 2            : 256 elements, 128 threads, two results per lane (k_exit_low<8,128>'s whole tile: the low16 map)
    //   NW = 2, DUP8      : 128 elements = 8 blocks, 128 threads (k_exit_low<8,128>'s half-tiles): columns n and n + 8 both carry
    //                       block n & 7, and lane (q, n) keeps set n >> 3 — one result per lane, every lane busy
    // `conv()` is called after the first constant matrices have been requested (their latency hides behind it): it must leave
    // `sub` in operand form and end with a barrier.  Results: canonical residues (plain bytes) at the swizzled chunk positions,
    // like phase().  Ends with a barrier.  DEPTH: units of 8 constant-matrix loads (1 KiB per wave each) in flight.
    template <int NW, bool DUP8, int DEPTH = 2, class Conv>
    __device__ static __forceinline__ void phase_n16(E* sub, const uint8_t* __restrict__ Amat, const unsigned long long* __restrict__ Kc, uint32_t tid, Conv&& conv) {
        static_assert(NW == 4 || NW == 2, "4 or 2 waves");
        static_assert(!DUP8 || NW == 2, "the 8-block form runs on two waves");
        constexpr int OPW = 16 / NW, SETS = OPW / 4, NU = 8 * SETS;        // unit u = (K-step u / SETS, set u % SETS)
        uint4* lds = reinterpret_cast<uint4*>(sub);
        const uint32_t L = tid & 63, w = (uint32_t)__builtin_amdgcn_readfirstlane((int)(tid >> 6)), n = L & 15, q = L >> 4;
        const uint32_t blk = DUP8 ? (n & 7u) : n;
        // constants: matrix (o, i) at (o * 16 + i) * 1024; lane offset inside it 16 * (16g + n + 32 (q & 1))
        const gchar ap = (gchar)(reinterpret_cast<const char*>(Amat)) + ((size_t)(OPW * w) * NB + (q >> 1)) * 1024 + 16u * (n + 32u * (q & 1u));
        auto ldA16 = [&](int u, int c) { const int s = u / SETS, oo = 4 * (u % SETS) + (c >> 1); return *(gv4)(ap + ((size_t)oo * NB + 2 * (size_t)s) * 1024 + 256 * (c & 1)); };
        v4i QA[DEPTH][8];
#pragma unroll
        for (int d = 0; d < DEPTH; ++d)
#pragma unroll
            for (int c = 0; c < 8; ++c) QA[d][c] = ldA16(d, c);
        __builtin_amdgcn_sched_barrier(0);
        conv();
        // data: chunk index of (block, K-step s, lane quarter q) is 32 block + 4s + q
        const char* lb = reinterpret_cast<const char*>(lds);
        auto ldB = [&](int s) { const uint32_t c = 32u * blk + 4u * (uint32_t)s + q; const uint4 b = *reinterpret_cast<const uint4*>(lb + 16u * ((c & ~15u) | ((c ^ blk) & 15u))); v4i v = {(int)b.x, (int)b.y, (int)b.z, (int)b.w}; return v; };
        v4i Bq[8];
#pragma unroll
        for (int s = 0; s < 8; ++s) Bq[s] = ldB(s);
        v4i acc[SETS][4][2];
#pragma unroll
        for (int st = 0; st < SETS; ++st)
#pragma unroll
            for (int oo = 0; oo < 4; ++oo) { acc[st][oo][0] = v4i{0, 0, 0, 0}; acc[st][oo][1] = v4i{0, 0, 0, 0}; }
        BLK16_STAMP(1)
#pragma unroll
        for (int u = 0; u < NU; ++u) {
            v4i A[8];
#pragma unroll
            for (int c = 0; c < 8; ++c) A[c] = QA[u % DEPTH][c];
            if (u + DEPTH < NU) {
#pragma unroll
                for (int c = 0; c < 8; ++c) QA[u % DEPTH][c] = ldA16(u + DEPTH, c);
            }
            __builtin_amdgcn_sched_barrier(0);          // keep the requests ABOVE the MFMAs
#pragma unroll
            for (int c = 0; c < 8; ++c) acc[u % SETS][c >> 1][c & 1] = __builtin_amdgcn_mfma_i32_16x16x64_i8(A[c], Bq[u / SETS], acc[u % SETS][c >> 1][c & 1], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
        BLK16_STAMP(2)
        // 4 x 4 transpose across the lane quarters: afterwards lane (q, n) holds what quarters 0..3 held of output 4 set + q
        int lo[SETS][16], hi[SETS][16];
#pragma unroll
        for (int st = 0; st < SETS; ++st)
#pragma unroll
            for (int g = 0; g < 2; ++g)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    auto p02 = __builtin_amdgcn_permlane32_swap((unsigned)acc[st][0][g][r], (unsigned)acc[st][2][g][r], false, false);
                    auto p13 = __builtin_amdgcn_permlane32_swap((unsigned)acc[st][1][g][r], (unsigned)acc[st][3][g][r], false, false);
                    auto x01 = __builtin_amdgcn_permlane16_swap(p02[0], p13[0], false, false);
                    auto x23 = __builtin_amdgcn_permlane16_swap(p02[1], p13[1], false, false);
                    lo[st][8 * g + r] = (int)x01[0];            // quarter 0: digit 8g + r
                    hi[st][8 * g + r] = (int)x01[1];            // quarter 1: digit 16 + 8g + r
                    lo[st][4 + 8 * g + r] = (int)x23[0];        // quarter 2: digit 4 + 8g + r
                    hi[st][4 + 8 * g + r] = (int)x23[1];        // quarter 3: digit 20 + 8g + r
                }
        constexpr int NR = DUP8 ? 1 : SETS;                     // results per lane
        E z[NR]; uint32_t jz[NR];
        if constexpr (DUP8) {
            const bool up = (n >> 3) != 0;                       // columns 8..15 duplicate blocks 0..7: they keep the second set
            int l2[16], h2[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) { l2[r] = up ? lo[1][r] : lo[0][r]; h2[r] = up ? hi[1][r] : hi[0][r]; }
            const uint32_t o = OPW * w + 4u * (n >> 3) + q;
            z[0] = normalise<true>(l2, h2, Kc + o * 8); jz[0] = blk * NB + o;
        } else {
#pragma unroll
            for (int st = 0; st < SETS; ++st) { const uint32_t o = OPW * w + 4u * (uint32_t)st + q; z[st] = normalise<true>(lo[st], hi[st], Kc + o * 8); jz[st] = n * NB + o; }
        }
        BLK16_STAMP(3)
        __syncthreads();        // every operand read of this phase is done: the array can be overwritten
        BLK16_STAMP(4)
#pragma unroll
        for (int i = 0; i < NR; ++i) {
            lds[phys(jz[i], 0)] = make_uint4(z[i].l[0], z[i].l[1], z[i].l[2], z[i].l[3]);
            lds[phys(jz[i], 1)] = make_uint4(z[i].l[4], z[i].l[5], z[i].l[6], z[i].l[7]);
        }
        __syncthreads();
    }
    // ... on an array whose element tid the calling thread holds in registers (one element per thread: 256 elements / 4 waves,
    // or 128 elements / 2 waves).  Returns element tid of the result; ends with a barrier.
    template <int NW, int DEPTH = 2>
    __device__ static __forceinline__ E phase_n16_regs(E* sub, const E& x, const uint8_t* __restrict__ Amat, const unsigned long long* __restrict__ Kc, uint32_t tid) {
        phase_n16<NW, NW == 2, DEPTH>(sub, Amat, Kc, tid, [&] {
            __syncthreads();                                    // readers of the array's previous contents are done
            store_operand(sub, tid, x);
            __syncthreads();
        });
        return load_swizzled(sub, tid);
    }

    // ---------------------------------------------------------------------------------------------------------------------
    // A 32-POINT map on a 1024-element array = 32 blocks of 32 (round 4: the FIVE lowest levels of ENTER / EXIT as one map, "low32").
    // 1024 constants = 1 MiB of matrices per direction, streamed once per tile (L2-resident: every tile of the launch reads the same
    // ones); 32 blocks = the 32 columns of ONE v_mfma_i32_32x32x32_i8, so a tile is 32 outputs x 32 inputs = 1024 MFMAs (the 16-point
    // map: 512) and 1024 normalisations.  It costs about 3 us more per tile than the 16-point map and removes a whole level — 33
    // sweep-steps of k_exit_low, 9 of k_enter_low.  (A 64-point map would stream 4 MiB per tile through a 4 MiB L2: not built.)
    // Element j = 32 n + i of block n; chunk q = 2j + half sits at (q & ~15) | ((q ^ n) & 15): the operand reads of a wave take the same
    // (i, half) of the 32 blocks (1 KiB apart), and each of ds_read_b128's lane groups holds 16 different n mod 16.
    static constexpr int NB32 = 32;
    static constexpr size_t kABytes32 = (size_t)NB32 * NB32 * 1024;
    static constexpr size_t kKWords32 = (size_t)NB32 * 8;
    static constexpr size_t kArenaElems32 = (kABytes32 + kKWords32 * 8) / sizeof(E);
    __host__ __device__ static inline uint32_t phys32(uint32_t j, uint32_t hh) {
        const uint32_t q = 2 * j + hh;
        return (q & ~15u) | ((q ^ (j >> 5)) & 15u);
    }
    // plain 1024-element tile -> operand form of the 32-point map (bytes xor 0x80).  512 threads.  Ends with a barrier.
    __device__ static __forceinline__ void to_operand_form32(E* tile, uint32_t tid) {
        uint4* lds = reinterpret_cast<uint4*>(tile);
        const uint32_t j0 = tid, j1 = 512 + tid;
        const uint4 a0 = lds[2 * j0], a1 = lds[2 * j0 + 1], b0 = lds[2 * j1], b1 = lds[2 * j1 + 1];
        __syncthreads();
        const uint32_t X = 0x80808080u;
        lds[phys32(j0, 0)] = make_uint4(a0.x ^ X, a0.y ^ X, a0.z ^ X, a0.w ^ X);
        lds[phys32(j0, 1)] = make_uint4(a1.x ^ X, a1.y ^ X, a1.z ^ X, a1.w ^ X);
        lds[phys32(j1, 0)] = make_uint4(b0.x ^ X, b0.y ^ X, b0.z ^ X, b0.w ^ X);
        lds[phys32(j1, 1)] = make_uint4(b1.x ^ X, b1.y ^ X, b1.z ^ X, b1.w ^ X);
        __syncthreads();
    }
    __device__ static __forceinline__ void from_swizzled32(E* tile, uint32_t tid) {
        const uint4* lds = reinterpret_cast<const uint4*>(tile);
        E x[2];
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const uint32_t j = tid + 512u * k;
            const uint4 a = lds[phys32(j, 0)], b = lds[phys32(j, 1)];
            x[k].l[0] = a.x; x[k].l[1] = a.y; x[k].l[2] = a.z; x[k].l[3] = a.w; x[k].l[4] = b.x; x[k].l[5] = b.y; x[k].l[6] = b.z; x[k].l[7] = b.w;
        }
        __syncthreads();
        tile[tid] = x[0]; tile[tid + 512] = x[1];
        __syncthreads();
    }
    // `sub`: 1024 elements in operand form (phys32).  512 threads: wave w produces outputs 4w .. 4w + 3 of the 32 blocks (four 32 x 32
    // accumulators); the register swap pairs outputs (4w, 4w + 1) and (4w + 2, 4w + 3): lane (h, n) ends with outputs 4w + h and
    // 4w + 2 + h of block n.  Results (plain bytes) at the swizzled positions; ends with a barrier.
    __device__ static __forceinline__ void phase32(E* sub, const uint8_t* __restrict__ Amat, const unsigned long long* __restrict__ Kc, uint32_t tid) {
        uint4* lds = reinterpret_cast<uint4*>(sub);
        const uint32_t L = tid & 63, w = (uint32_t)__builtin_amdgcn_readfirstlane((int)(tid >> 6)), n = L & 31, h = L >> 5;
        const gchar ap = (gchar)(reinterpret_cast<const char*>(Amat)) + ((size_t)(4 * w) * NB32) * 1024 + L * 16;
        const char* lb = reinterpret_cast<const char*>(lds);
        // two passes of two outputs each (two 32 x 32 accumulators): four at once need > 128 VGPRs next to the low-level kernels' other
        // registers and spill; the data operands are read from LDS again in the second pass
        E outz[2];
#pragma unroll
        for (int pr = 0; pr < 2; ++pr) {                                // unrolled: outz[pr] must be a register, not an indexed array
            v16i acc0 = {0}, acc1 = {0};
            gchar pa = ap + (size_t)(2 * pr) * NB32 * 1024;            // matrices (4w + 2 pr, i) at pa + 1024 i, (4w + 2 pr + 1, i) 32 KiB above
            v4i A0 = *(gv4)pa, A1 = *(gv4)(pa + 32 * 1024), nA0 = *(gv4)(pa + 1024), nA1 = *(gv4)(pa + 33 * 1024);     // two inputs in flight
            pa += 2048;
            uint32_t cb = 64u * n + h;                                   // chunk index of (block n, input i, half h) = 64 n + 2 i + h
            auto ldBc = [&](uint32_t c) { const uint4 b = *reinterpret_cast<const uint4*>(lb + 16u * ((c & ~15u) | ((c ^ n) & 15u))); v4i v = {(int)b.x, (int)b.y, (int)b.z, (int)b.w}; return v; };
            v4i B0 = ldBc(cb);
#pragma unroll 2
            for (int i = 0; i < NB32; ++i) {
                v4i n2A0 = nA0, n2A1 = nA1, nB = B0;
                if (i + 2 < NB32) { n2A0 = *(gv4)pa; n2A1 = *(gv4)(pa + 32 * 1024); pa += 1024; }
                if (i + 1 < NB32) { cb += 2; nB = ldBc(cb); }
                __builtin_amdgcn_sched_barrier(0);
                acc0 = __builtin_amdgcn_mfma_i32_32x32x32_i8(A0, B0, acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_i32_32x32x32_i8(A1, B0, acc1, 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                A0 = nA0; A1 = nA1; nA0 = n2A0; nA1 = n2A1; B0 = nB;
            }
            int lo[16], hi[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                auto p = __builtin_amdgcn_permlane32_swap((unsigned)acc0[r], (unsigned)acc1[r], false, false);
                lo[r] = (int)p[0]; hi[r] = (int)p[1];
            }
            outz[pr] = normalise<true>(lo, hi, Kc + (4 * w + 2 * pr + h) * 8);
        }
        __syncthreads();        // every operand read of this phase is done
#pragma unroll
        for (int pr = 0; pr < 2; ++pr) {
            const uint32_t j = n * NB32 + 4 * w + 2 * pr + h;
            lds[phys32(j, 0)] = make_uint4(outz[pr].l[0], outz[pr].l[1], outz[pr].l[2], outz[pr].l[3]);
            lds[phys32(j, 1)] = make_uint4(outz[pr].l[4], outz[pr].l[5], outz[pr].l[6], outz[pr].l[7]);
        }
        __syncthreads();
    }

    // ---- construction: the 16 x 16 matrix of the tree (the kernels' own stage code applied to the unit vectors), its
    // int8 expansion and the accumulator seeds.  One workgroup of 256 threads per (tree, parity).
    __device__ static inline void signed_digits(const E& c, int8_t d[32]) {
        // value c or c - p, whichever lies in [-0x8080..80, 0x7f7f..7f]: digits of (pattern + 0x8080..80) xor 0x80
        bool big = false, decided = false;
        for (int i = 7; i >= 0; --i) { if (!decided && c.l[i] != 0x7f7f7f7fu) { big = c.l[i] > 0x7f7f7f7fu; decided = true; } }
        uint32_t wv[8];
        uint64_t cy = big ? 977u : 0u;
        for (int i = 0; i < 8; ++i) { cy += (uint64_t)c.l[i] + ((big && i == 1) ? 1u : 0u); wv[i] = (uint32_t)cy; cy >>= 32; }
        cy = 0;
        for (int i = 0; i < 8; ++i) { cy += (uint64_t)wv[i] + 0x80808080u; wv[i] = (uint32_t)cy; cy >>= 32; }
        for (int j = 0; j < 32; ++j) d[j] = (int8_t)(((wv[j >> 2] >> (8 * (j & 3))) & 0xffu) ^ 0x80u);
    }
};

__device__ __forceinline__ void blk16_expand(Fe256* Tm, Fe256* csum, uint8_t* __restrict__ Amat, unsigned long long* __restrict__ Kc, uint32_t tid);

// np0 / dinv: decompose tables of the source parity, p0 / p1: recombine tables of the target parity, inner: merged innermost
// pair (all as the row kernel receives them), e = vector length of the tree (>= 16).  grid = 2: block sg builds the map of source
// parity sg.  The unit-vector pass is spread over the workgroup (round 4; it was 16 serial threads with a 16-element array each —
// 528 B of scratch): the 16 x 16 array of images lives in LDS, thread t < 128 runs butterfly t & 7 of column t >> 3 at every one of
// the 7 stages.
struct Blk16BuildArgs { const Te256 *np0, *dinv, *p0, *p1, *inner; uint8_t* A; unsigned long long* K; };
__global__ __launch_bounds__(256) void k_blk16_build(Blk16BuildArgs a0, Blk16BuildArgs a1, size_t e) {
    using F = Secp256k1; using E = Fe256;
    __shared__ E Tm[256];
    __shared__ E csum[256];
    const Blk16BuildArgs& a = blockIdx.x ? a1 : a0;
    const uint32_t tid = threadIdx.x;
    // Tm[o * 16 + c] = component o of the image of unit vector c (column c of T)
    Tm[tid] = ((tid >> 4) == (tid & 15u)) ? F::one() : F::zero();
    __syncthreads();
    const uint32_t c = tid >> 3, g = tid & 7u;
    for (int st = 0; st < 7; ++st) {
        if (tid < 128) {
            if (st < 3) {                                   // decompose, pair distance 8, 4, 2
                const int lh = 3 - st; const uint32_t h = 1u << lh; const size_t off = e - 2 * (size_t)h;
                const uint32_t i = g & (h - 1), idx = ((g >> lh) << (lh + 1)) + i;
                const E x = Tm[idx * 16 + c], y = Tm[(idx + h) * 16 + c];
                const E q1 = F::tmul(a.dinv[off + i], F::sub(y, x));
                Tm[idx * 16 + c] = F::tmul_add(a.np0[off + i], q1, x); Tm[(idx + h) * 16 + c] = q1;
            } else if (st == 3) {                           // merged innermost pair
                const E x = Tm[(2 * g) * 16 + c], d = F::sub(Tm[(2 * g + 1) * 16 + c], x);
                Tm[(2 * g) * 16 + c] = F::tmul_add(a.inner[0], d, x); Tm[(2 * g + 1) * 16 + c] = F::tmul_add(a.inner[1], d, x);
            } else {                                        // recombine, pair distance 2, 4, 8
                const int lh = st - 3; const uint32_t h = 1u << lh; const size_t off = e - 2 * (size_t)h;
                const uint32_t i = g & (h - 1), idx = ((g >> lh) << (lh + 1)) + i;
                const E x = Tm[idx * 16 + c], y = Tm[(idx + h) * 16 + c];
                Tm[idx * 16 + c] = F::tmul_add(a.p0[off + i], y, x); Tm[(idx + h) * 16 + c] = F::tmul_add(a.p1[off + i], y, x);
            }
        }
        __syncthreads();
    }
    blk16_expand(Tm, csum, a.A, a.K, tid);
}

// the 256 constants of a 16 x 16 map (shared memory, row-major [output][input], plain residues) -> int8 matrices + accumulator seeds
__device__ __forceinline__ void blk16_expand(Fe256* Tm, Fe256* csum, uint8_t* __restrict__ Amat, unsigned long long* __restrict__ Kc, uint32_t tid) {
    using F = Secp256k1; using E = Fe256;
    __syncthreads();
    {
        const uint32_t o = tid >> 4, i = tid & 15;
        E c = Tm[tid], sum = F::zero();
        const E f256 = F::from_u32(256);
        uint8_t* A = Amat + ((size_t)o * 16 + i) * 1024;
        for (uint32_t j = 0; j < 32; ++j) {
            int8_t d[32];
            Blk16::signed_digits(c, d);
            const uint32_t hh = j >> 4, q = j & 15;
            for (uint32_t b = 0; b < 32; ++b) A[(Blk16::row_of_digit(b) + 32 * hh) * 16 + q] = (uint8_t)d[b];
            sum = F::add(sum, c);
            c = F::mul(c, f256);
        }
        csum[tid] = sum;
    }
    __syncthreads();
    if (tid < 16) {
        E s = F::zero();
        for (int i = 0; i < 16; ++i) s = F::add(s, csum[tid * 16 + i]);
        // offsets 2^50 on each of the eight 64-bit accumulators: sum_g 2^(50 + 32g) mod p, with 2^274 = 2^18 (2^32 + 977)
        E off; off.l[0] = 977u << 18; off.l[1] = 1u << 19; for (int i = 2; i < 8; ++i) off.l[i] = 1u << 18;
        const E kap = F::sub(F::mul(s, F::from_u32(128)), off);
        for (int g = 0; g < 8; ++g) Kc[tid * 8 + g] = (1ull << 50) + kap.l[g];
    }
}

// the map given explicitly as 256 plain constants, row-major [output][input] (test hook ecfft_selftest_blk16)
// (also DeviceChain::build_low16: the four lowest ENTER / EXIT levels of a 16-block, images of the unit vectors = columns, `transposed`)
__global__ __launch_bounds__(256) void k_blk16_from_matrix(const Fe256* __restrict__ T, uint8_t* __restrict__ Amat, unsigned long long* __restrict__ Kc, bool transposed) {
    __shared__ Fe256 Tm[256];
    __shared__ Fe256 csum[256];
    Tm[threadIdx.x] = T[transposed ? (threadIdx.x & 15u) * 16u + (threadIdx.x >> 4) : threadIdx.x];
    blk16_expand(Tm, csum, Amat, Kc, threadIdx.x);
}
// The 32-point map given as 1024 plain constants T[o * 32 + i] (or transposed: images of the unit vectors = columns): int8 matrices
// (one thread per constant, grid = 4 x 256) and the per-constant sums sum_j c 2^(8j); k_blk32_seeds turns those into the 32 x 8
// accumulator seeds (as blk16_expand does for 16 x 16).
__global__ __launch_bounds__(256) void k_blk32_expand(const Fe256* __restrict__ T, uint8_t* __restrict__ Amat, Fe256* __restrict__ csum, bool transposed) {
    using F = Secp256k1; using E = Fe256;
    const uint32_t idx = blockIdx.x * 256u + threadIdx.x, o = idx >> 5, i = idx & 31u;
    E c = T[transposed ? i * 32u + o : idx], sum = F::zero();
    const E f256 = F::from_u32(256);
    uint8_t* A = Amat + (size_t)idx * 1024;
    for (uint32_t j = 0; j < 32; ++j) {
        int8_t d[32];
        Blk16::signed_digits(c, d);
        const uint32_t hh = j >> 4, q = j & 15;
        for (uint32_t b = 0; b < 32; ++b) A[(Blk16::row_of_digit(b) + 32 * hh) * 16 + q] = (uint8_t)d[b];
        sum = F::add(sum, c);
        c = F::mul(c, f256);
    }
    csum[idx] = sum;
}
__global__ __launch_bounds__(32) void k_blk32_seeds(const Fe256* __restrict__ csum, unsigned long long* __restrict__ Kc) {
    using F = Secp256k1; using E = Fe256;
    const uint32_t o = threadIdx.x;
    E s = F::zero();
    for (int i = 0; i < 32; ++i) s = F::add(s, csum[o * 32 + i]);
    E off; off.l[0] = 977u << 18; off.l[1] = 1u << 19; for (int i = 2; i < 8; ++i) off.l[i] = 1u << 18;
    const E kap = F::sub(F::mul(s, F::from_u32(128)), off);
    for (int g = 0; g < 8; ++g) Kc[o * 8 + g] = (1ull << 50) + kap.l[g];
}
#ifdef ECFFT_TEST_HOOKS     // kernels behind ecfft_selftest_blk16 / _blk16_small / _blk32 (include/ecfft_hip_hooks.h): test builds only
// test hook: the 32-point phase alone on tiles of 1024 elements (grid = tiles, 512 threads), in place
__global__ __launch_bounds__(512) void k_blk32_apply(Fe256* __restrict__ data, const uint8_t* __restrict__ Amat, const unsigned long long* __restrict__ Kc) {
    __shared__ Fe256 tile[1024];
    const uint32_t tid = threadIdx.x;
    Fe256* g = data + (size_t)blockIdx.x * 1024;
    tile[tid] = g[tid]; tile[tid + 512] = g[tid + 512];
    __syncthreads();
    Blk16::to_operand_form32(tile, tid);
    Blk16::phase32(tile, Amat, Kc, tid);
    Blk16::from_swizzled32(tile, tid);
    g[tid] = tile[tid]; g[tid + 512] = tile[tid + 512];
}

// test hook: the matrix-core phase alone on tiles of 1024 elements (grid = tiles, 512 threads), in place
__global__ __launch_bounds__(512) void k_blk16_apply(Fe256* __restrict__ data, const uint8_t* __restrict__ Amat, const unsigned long long* __restrict__ Kc) {
    __shared__ Fe256 tile[Blk16::kSub];
    const uint32_t tid = threadIdx.x;
    Fe256* g = data + (size_t)blockIdx.x * Blk16::kSub;
    for (uint32_t j = tid; j < (uint32_t)Blk16::kSub; j += 512) tile[j] = g[j];
    __syncthreads();
    Blk16::APre pre = Blk16::prefetch(Amat, tid);
    __builtin_amdgcn_sched_barrier(0);
    Blk16::to_operand_form<512>(tile, Blk16::kSub, tid);
    Blk16::phase(tile, Amat, Kc, tid, pre);
    Blk16::from_swizzled<512>(tile, Blk16::kSub, tid);
    for (uint32_t j = tid; j < (uint32_t)Blk16::kSub; j += 512) g[j] = tile[j];
}

// test hook: the small-launch forms (v_mfma_i32_16x16x64_i8) alone, in place.  MODE 1: 256-element tiles, 256 threads, LDS-resident
// array (k_enter_low<8,256>'s low16);  2: 256-element tiles, 128 threads (k_exit_low<8,128>'s low16);  3: 128-element tiles, 128
// threads, element in registers (k_exit_low<8,128>'s half-tiles);  4: 256-element tiles, 256 threads, element in registers (the
// 256-element row kernel, k_enter_low<8,256>'s EXTEND cores)
template <int MODE>
__global__ __launch_bounds__(MODE == 1 || MODE == 4 ? 256 : 128) void k_blk16_apply_n16(Fe256* __restrict__ data, const uint8_t* __restrict__ Amat, const unsigned long long* __restrict__ Kc) {
    constexpr int T = MODE == 3 ? 128 : 256, BLK = (MODE == 1 || MODE == 4) ? 256 : 128;
    __shared__ Fe256 tile[T];
    const uint32_t tid = threadIdx.x;
    Fe256* g = data + (size_t)blockIdx.x * T;
    if constexpr (MODE == 1 || MODE == 2) {
        for (uint32_t j = tid; j < (uint32_t)T; j += BLK) tile[j] = g[j];
        __syncthreads();
        Blk16::phase_n16<MODE == 1 ? 4 : 2, false>(tile, Amat, Kc, tid, [&] { Blk16::to_operand_form<BLK>(tile, T, tid); });
        Blk16::from_swizzled<BLK>(tile, T, tid);
        for (uint32_t j = tid; j < (uint32_t)T; j += BLK) g[j] = tile[j];
    } else {
        const Fe256 x = g[tid];
        g[tid] = Blk16::phase_n16_regs<MODE == 4 ? 4 : 2>(tile, x, Amat, Kc, tid);
    }
}

#endif  // ECFFT_TEST_HOOKS

}  // namespace ecfft
