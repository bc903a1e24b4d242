// secp256k1 base field for the MI355X path: p = 2^256 - 2^32 - 977.
//
// Replaces the arithmetic the reference gets from ark-ff (`Fp256<MontBackend<FqConfig, 4>>`,
// /root/reference/src/lib.rs:31-37; call sites src/utils.rs:341-346, src/fftree.rs:94,115,157-158,
// 217-219,238,253-255,279).
//
// Representation contract.  USER DATA crosses the C-ABI in the crate's in-memory form: four
// little-endian u64 limbs holding x*2^256 mod p (Montgomery form), fully reduced.  Every multiply on
// the ENTER/EXIT/EXTEND path is data x precomputed-constant (never data x data), so all TABLES are
// kept in PLAIN form t and the device computes the plain product (xR)*t mod p = (x t)R: data stays in
// Montgomery form bit-exactly and no Montgomery reduction is ever executed.  The 512-bit product is
// reduced with the pseudo-Mersenne fold 2^256 = 2^32 + 977 (mod p).
//
// Device code uses 8 x u32 limbs so that every partial product is one v_mad_u64_u32 (measured on
// gfx950: ~5 cycles per wave-instruction, see tools/ubench/intops.hip); host code (tree construction
// only) uses 4 x u64 limbs with unsigned __int128.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace ecfft {

struct alignas(16) Fe256 {
    uint32_t l[8];
};

// A table constant t as the butterfly kernels read it: the pair (t, u = t * 2^128 mod p).  t*x = t*x_lo + u*x_hi is then a
// 12-word instead of a 16-word product: a 4-word fold, no second fold on the fast path (tools/gen_mulmod_asm.py) —
// 14% more multiplies per second on MI355X for twice the (L2-resident) table bytes.
struct alignas(16) Te256 {
    Fe256 t, u;
};

struct Secp256k1 {
    using elem = Fe256;
    using telem = Te256;
    static constexpr int kBytes = 32;
    static constexpr int kFieldId = 0;
    static constexpr uint32_t C977 = 977u;  // p = 2^256 - (2^32 + 977)

    __host__ __device__ static inline elem zero() { elem r; for (int i = 0; i < 8; ++i) r.l[i] = 0; return r; }
    __host__ __device__ static inline elem from_u32(uint32_t v) { elem r = zero(); r.l[0] = v; return r; }
    __host__ __device__ static inline elem one() { return from_u32(1); }
    __host__ __device__ static inline bool is_zero(const elem& a) {
        uint32_t o = 0; for (int i = 0; i < 8; ++i) o |= a.l[i]; return o == 0;
    }
    __host__ __device__ static inline bool eq(const elem& a, const elem& b) {
        uint32_t o = 0; for (int i = 0; i < 8; ++i) o |= a.l[i] ^ b.l[i]; return o == 0;
    }
    __host__ __device__ static inline uint32_t p_limb(int i) { return i == 0 ? 0xFFFFFC2Fu : (i == 1 ? 0xFFFFFFFEu : 0xFFFFFFFFu); }

    // a + b mod p, inputs canonical
    __host__ __device__ static inline elem add(const elem& a, const elem& b) {
        uint32_t s[8]; uint64_t c = 0;
#pragma unroll
        for (int i = 0; i < 8; ++i) { c += (uint64_t)a.l[i] + b.l[i]; s[i] = (uint32_t)c; c >>= 32; }
        return finish(s, (uint32_t)c);
    }
    // a - b mod p
    __host__ __device__ static inline elem sub(const elem& a, const elem& b) {
#if defined(__HIP_DEVICE_COMPILE__) && !defined(ECFFT_NO_ASM_MUL)
        // 18 instructions: 8-word borrow chain, then subtract (borrow ? 2^32 + 977 : 0), i.e. add p modulo 2^256
        // (hipcc lowers the portable code below to ~64 instructions)
        uint32_t d0, d1, d2, d3, d4, d5, d6, d7, k0, k1;
        const uint32_t c977 = C977;
        asm("v_sub_co_u32_e32 %0, vcc, %10, %18\n\t"
            "v_subb_co_u32_e32 %1, vcc, %11, %19, vcc\n\t"
            "v_subb_co_u32_e32 %2, vcc, %12, %20, vcc\n\t"
            "v_subb_co_u32_e32 %3, vcc, %13, %21, vcc\n\t"
            "v_subb_co_u32_e32 %4, vcc, %14, %22, vcc\n\t"
            "v_subb_co_u32_e32 %5, vcc, %15, %23, vcc\n\t"
            "v_subb_co_u32_e32 %6, vcc, %16, %24, vcc\n\t"
            "v_subb_co_u32_e32 %7, vcc, %17, %25, vcc\n\t"
            "v_cndmask_b32_e64 %9, 0, 1, vcc\n\t"
            "v_mul_u32_u24_e32 %8, %26, %9\n\t"
            "v_sub_co_u32_e32 %0, vcc, %0, %8\n\t"
            "v_subb_co_u32_e32 %1, vcc, %1, %9, vcc\n\t"
            "v_subbrev_co_u32_e32 %2, vcc, 0, %2, vcc\n\t"
            "v_subbrev_co_u32_e32 %3, vcc, 0, %3, vcc\n\t"
            "v_subbrev_co_u32_e32 %4, vcc, 0, %4, vcc\n\t"
            "v_subbrev_co_u32_e32 %5, vcc, 0, %5, vcc\n\t"
            "v_subbrev_co_u32_e32 %6, vcc, 0, %6, vcc\n\t"
            "v_subbrev_co_u32_e32 %7, vcc, 0, %7, vcc"
            : "=&v"(d0), "=&v"(d1), "=&v"(d2), "=&v"(d3), "=&v"(d4), "=&v"(d5), "=&v"(d6), "=&v"(d7), "=&v"(k0), "=&v"(k1)
            : "v"(a.l[0]), "v"(a.l[1]), "v"(a.l[2]), "v"(a.l[3]), "v"(a.l[4]), "v"(a.l[5]), "v"(a.l[6]), "v"(a.l[7]),
              "v"(b.l[0]), "v"(b.l[1]), "v"(b.l[2]), "v"(b.l[3]), "v"(b.l[4]), "v"(b.l[5]), "v"(b.l[6]), "v"(b.l[7]), "s"(c977)
            : "vcc");
        elem r; r.l[0] = d0; r.l[1] = d1; r.l[2] = d2; r.l[3] = d3; r.l[4] = d4; r.l[5] = d5; r.l[6] = d6; r.l[7] = d7;
        return r;
#else
        uint32_t d[8]; int64_t bw = 0;
#pragma unroll
        for (int i = 0; i < 8; ++i) { int64_t t = (int64_t)a.l[i] - b.l[i] + bw; d[i] = (uint32_t)t; bw = t >> 32; }
        // if borrow: add p  (= subtract 2^32 + 977 modulo 2^256)
        uint32_t m = (uint32_t)bw;  // 0 or 0xFFFFFFFF
        elem r; int64_t c2 = 0;
        uint32_t k0 = m & C977, k1 = m & 1u;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            int64_t t = (int64_t)d[i] - (i == 0 ? k0 : (i == 1 ? k1 : 0u)) + c2; r.l[i] = (uint32_t)t; c2 = t >> 32;
        }
        return r;
#endif
    }
    __host__ __device__ static inline elem neg(const elem& a) { return sub(zero(), a); }

    // value = s + carry*2^256 < 2^256 + 2^255 (say); returns it mod p, canonical
    __host__ __device__ static inline elem finish(const uint32_t s[8], uint32_t carry) {
        // u = s + (2^32 + 977); if that overflows 2^256 (or carry was set) then value >= p -> take u
        uint32_t u[8]; uint64_t c = 0;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            c += (uint64_t)s[i] + (i == 0 ? C977 : (i == 1 ? 1u : 0u)); u[i] = (uint32_t)c; c >>= 32;
        }
        bool ge = (carry | (uint32_t)c) != 0;
        elem r;
#pragma unroll
        for (int i = 0; i < 8; ++i) r.l[i] = ge ? u[i] : s[i];
        return r;
    }

    // w[0..15] = t * x + c   (c may be null => 0).  Operand scanning; each step is one v_mad_u64_u32
    __host__ __device__ static inline void mul_wide(const elem& t, const elem& x, const elem* c, uint32_t w[16]) {
#if !defined(__HIP_DEVICE_COMPILE__)
        // host (tree construction only): 4 x u64 limbs
        typedef unsigned __int128 u128;
        uint64_t T[4], X[4], W[8];
        for (int i = 0; i < 4; ++i) {
            T[i] = (uint64_t)t.l[2 * i] | ((uint64_t)t.l[2 * i + 1] << 32);
            X[i] = (uint64_t)x.l[2 * i] | ((uint64_t)x.l[2 * i + 1] << 32);
            W[i] = c ? ((uint64_t)c->l[2 * i] | ((uint64_t)c->l[2 * i + 1] << 32)) : 0; W[i + 4] = 0;
        }
        for (int i = 0; i < 4; ++i) {
            uint64_t carry = 0;
            for (int j = 0; j < 4; ++j) {
                u128 acc = (u128)T[i] * X[j] + W[i + j] + carry; W[i + j] = (uint64_t)acc; carry = (uint64_t)(acc >> 64);
            }
            W[i + 4] = carry;
        }
        for (int i = 0; i < 8; ++i) { w[2 * i] = (uint32_t)W[i]; w[2 * i + 1] = (uint32_t)(W[i] >> 32); }
        return;
#endif
#pragma unroll
        for (int j = 0; j < 8; ++j) w[j] = c ? c->l[j] : 0u;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            uint64_t carry = 0;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                uint64_t acc = (uint64_t)t.l[i] * x.l[j] + w[i + j] + carry;
                w[i + j] = (uint32_t)acc; carry = acc >> 32;
            }
            w[i + 8] = (uint32_t)carry;
        }
    }
    // 512-bit w -> canonical residue mod p
    __host__ __device__ static inline elem reduce_wide(const uint32_t w[16]) {
        // fold 1: v = lo + hi*977 + (hi << 32)   (10 words)
        uint32_t v[8]; uint64_t acc = 0;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            acc += (uint64_t)w[8 + j] * C977 + w[j] + (j ? w[8 + j - 1] : 0u);
            v[j] = (uint32_t)acc; acc >>= 32;
        }
        acc += w[15];                       // top = acc < 2^34
        // fold 2: v += top*977 + (top << 32)
        uint64_t top = acc;
        uint64_t a0 = top * C977;           // < 2^44
        uint64_t c = (uint64_t)v[0] + (uint32_t)a0; v[0] = (uint32_t)c; c >>= 32;
        c += (uint64_t)v[1] + (a0 >> 32) + (uint32_t)top; v[1] = (uint32_t)c; c >>= 32;
        c += (uint64_t)v[2] + (top >> 32); v[2] = (uint32_t)c; c >>= 32;
#pragma unroll
        for (int j = 3; j < 8; ++j) { c += v[j]; v[j] = (uint32_t)c; c >>= 32; }
        return finish(v, (uint32_t)c);
    }
#if defined(__HIP_DEVICE_COMPILE__) && !defined(ECFFT_NO_ASM_MUL)
    // hand-scheduled gfx950 instruction streams (generated by tools/gen_mulmod_asm.py)
#include "secp256k1_mul_gfx950.inc"
    __device__ static inline elem mul(const elem& t, const elem& x) { return mul_gfx950(t, x); }
    __device__ static inline elem mul_add(const elem& t, const elem& x, const elem& c) { return mul_add_gfx950(t, x, c); }
    // table constant given as the pair (t, u = t * 2^128 mod p): 12-word product, see tools/gen_mulmod_asm.py
    __device__ static inline elem mul2(const elem& t, const elem& u, const elem& x) { return mul2_gfx950(t, u, x); }
    __device__ static inline elem mul2_add(const elem& t, const elem& u, const elem& x, const elem& c) { return mul2_add_gfx950(t, u, x, c); }
#else
    // t*x mod p
    __host__ __device__ static inline elem mul(const elem& t, const elem& x) {
        uint32_t w[16]; mul_wide(t, x, nullptr, w); return reduce_wide(w);
    }
    // t*x + c mod p  (the modular add is free: c seeds the accumulator of the first row)
    __host__ __device__ static inline elem mul_add(const elem& t, const elem& x, const elem& c) {
        uint32_t w[16]; mul_wide(t, x, &c, w); return reduce_wide(w);
    }
    __host__ __device__ static inline elem mul2(const elem& t, const elem&, const elem& x) { return mul(t, x); }
    __host__ __device__ static inline elem mul2_add(const elem& t, const elem&, const elem& x, const elem& c) { return mul_add(t, x, c); }
#endif
    __host__ __device__ static inline telem to_table(const elem& t) {
        telem r; r.t = t; elem k = zero(); k.l[4] = 1; r.u = mul(t, k); return r;
    }
    __host__ __device__ static inline elem tmul(const telem& T, const elem& x) { return mul2(T.t, T.u, x); }
    __host__ __device__ static inline elem tmul_add(const telem& T, const elem& x, const elem& c) { return mul2_add(T.t, T.u, x, c); }
    // the butterfly kernels' values are always canonical for this field (M31 keeps a lazy range, see field_m31.h)
    __host__ __device__ static inline const elem& canon(const elem& x) { return x; }
    __host__ __device__ static inline elem sqr(const elem& a) { return mul(a, a); }

    __host__ __device__ static inline elem pow_u64(const elem& a, uint64_t e) {
        elem r = one();
        for (int i = 63; i >= 0; --i) { r = sqr(r); if ((e >> i) & 1) r = mul(r, a); }
        return r;
    }
    __host__ __device__ static inline elem sqrn(elem a, int n) { for (int i = 0; i < n; ++i) a = sqr(a); return a; }
    // a^(p-2), p-2 = 2^256 - 2^32 - 979 = 223 ones, 0, 22 ones, 0000, 1, 0, 11, 0, 1 (binary): the addition chain that builds
    // 2^k - 1 for k = 2, 3, 6, 9, 11, 22, 44, 88, 176, 220, 223 — 255 squarings + 15 multiplies instead of the 256 + 248 of
    // square-and-multiply.  The chain is a dependent sequence, so its LENGTH is the latency of every batched inversion of the
    // construction (device_tree.h batch_inv: one inversion per chunk, all chunks in parallel).
    __host__ __device__ static inline elem inv(const elem& a) {
        const elem x2 = mul(sqr(a), a), x3 = mul(sqr(x2), a);
        const elem x6 = mul(sqrn(x3, 3), x3), x9 = mul(sqrn(x6, 3), x3), x11 = mul(sqrn(x9, 2), x2);
        const elem x22 = mul(sqrn(x11, 11), x11), x44 = mul(sqrn(x22, 22), x22), x88 = mul(sqrn(x44, 44), x44);
        const elem x176 = mul(sqrn(x88, 88), x88), x220 = mul(sqrn(x176, 44), x44), x223 = mul(sqrn(x220, 3), x3);
        elem t = mul(sqrn(x223, 23), x22);
        t = mul(sqrn(t, 5), a);
        t = mul(sqrn(t, 3), x2);
        return mul(sqrn(t, 2), a);
    }
    // a^((p+1)/4); (p+1)/4 = 2^254 - 2^30 - 244 -> words [0xBFFFFF0C, 0xFFFFFFFF x6, 0x3FFFFFFF]
    __host__ static inline bool sqrt(const elem& a, elem* out) {
        elem r = one();
        for (int i = 255; i >= 0; --i) {
            r = sqr(r);
            uint32_t wv = (i >= 224) ? 0x3FFFFFFFu : (i >= 32 ? 0xFFFFFFFFu : 0xBFFFFF0Cu);
            if ((wv >> (i & 31)) & 1) r = mul(r, a);
        }
        if (!eq(sqr(r), a)) return false;
        *out = r; return true;
    }
    // plain <-> Montgomery (x -> x * 2^256 mod p):  2^256 mod p = 2^32 + 977
    __host__ __device__ static inline elem to_mont(const elem& a) {
        elem r256 = zero(); r256.l[0] = C977; r256.l[1] = 1; return mul(a, r256);
    }
    // host only: decimal literal -> plain element
    __host__ static inline elem from_dec(const char* s) {
        elem r = zero(); elem ten = from_u32(10);
        for (; *s; ++s) r = add(mul(r, ten), from_u32((uint32_t)(*s - '0')));
        return r;
    }
    __host__ static inline int cmp(const elem& a, const elem& b) {
        for (int i = 7; i >= 0; --i) { if (a.l[i] < b.l[i]) return -1; if (a.l[i] > b.l[i]) return 1; }
        return 0;
    }
};

}  // namespace ecfft
