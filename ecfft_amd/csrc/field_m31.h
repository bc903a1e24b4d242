// Mersenne-31 field for the MI355X path: p = 2^31 - 1, one u32 per element holding the plain
// canonical residue — the in-memory form of `ark_ff_optimized::fp31::Fp` (/root/reference/src/lib.rs:196).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace ecfft {

struct M31 {
    using elem = uint32_t;
    using telem = uint32_t;    // table constant as the butterflies read it: the DOUBLED residue 2t (see to_table below; secp256k1 stores a pair)
    static constexpr int kBytes = 4;
    static constexpr int kFieldId = 1;
    static constexpr uint32_t P = 0x7FFFFFFFu;

    __host__ __device__ static inline elem zero() { return 0; }
    __host__ __device__ static inline elem one() { return 1; }
    __host__ __device__ static inline elem from_u32(uint32_t v) { return v % P; }
    __host__ __device__ static inline bool is_zero(elem a) { return a == 0; }
    __host__ __device__ static inline bool eq(elem a, elem b) { return a == b; }
    __host__ __device__ static inline elem add(elem a, elem b) { uint32_t s = a + b; return s >= P ? s - P : s; }
    // a - b for a, b in [0, p]: d = a - b wraps to >= 2^32 - p exactly when a < b, and then d + p (mod 2^32) is the small
    // representative — min(d, d + p) picks the right one in 3 plain VALU instructions with no compare / VCC
    __host__ __device__ static inline elem sub(elem a, elem b) { uint32_t d = a - b, w = d + P; return d < w ? d : w; }
    __host__ __device__ static inline elem neg(elem a) { return a ? P - a : 0; }
    __host__ __device__ static inline elem red64(uint64_t t) {  // t < 2^62 + 2^31 (a product of residues plus a residue)
        uint32_t lo = (uint32_t)t & P, hi = (uint32_t)(t >> 31);     // hi <= 2^31, so lo + hi < 2^32
        uint32_t r = lo + hi;
        r = (r & P) + (r >> 31);                                     // <= 2^31
        uint32_t d = r - P;                                          // wraps to a huge value when r < P
        return d < r ? d : r;
    }
    __host__ __device__ static inline elem mul(elem t, elem x) { return red64((uint64_t)t * x); }
    __host__ __device__ static inline elem mul_add(elem t, elem x, elem c) { return red64((uint64_t)t * x + c); }
    // A table constant is stored DOUBLED, t2 = 2t (< 2^32 for canonical t).  Then the 64-bit product t2*x + 2c = 2(t*x + c) has
    // floor((t*x + c) / 2^31) in its HIGH word and 2*((t*x + c) mod 2^31) in its LOW word: the 31-bit split of the pseudo-Mersenne
    // fold is "low word >> 1" and "high word" — two plain two-operand VALU instructions instead of v_and + the 4-cycle v_alignbit.
    __host__ __device__ static inline telem to_table(elem t) { return t << 1; }
    // Table x data inside the butterfly kernels works on the LAZY range [0, p] (p itself = a second representative of 0):
    // for t <= p-1 and x, c <= p the product t*x + c <= p^2 < 2^62, so lo + hi <= 2^32 - 2 and the second fold lands in
    // [0, p] again — the final conditional subtract of the canonical reduction is dropped — 6 instead of 9 instructions per
    // multiply.  sub() maps [0, p] x [0, p] into [0, p] as written.  Every value that leaves a kernel for HBM goes through canon().
    __host__ __device__ static inline elem fold2_lazy(uint64_t twice) {      // twice = 2*(t*x + c) < 2^63
        uint32_t r = ((uint32_t)twice >> 1) + (uint32_t)(twice >> 32);      // lo31 + hi <= 2^32 - 2
        uint32_t d = r - P;                       // second fold as subtract + unsigned min (2 plain VALU instructions instead of
        return d < r ? d : r;                     // and / shift / add): r >= p -> r - p in [0, p]; r < p -> the difference wraps, keep r
    }
    __host__ __device__ static inline elem red64_lazy(uint64_t t) { return fold2_lazy(t << 1); }   // plain (undoubled) operand; t < 2^62 + 2^31
    __host__ __device__ static inline elem tmul(telem t2, elem x) { return fold2_lazy((uint64_t)t2 * x); }
    __host__ __device__ static inline elem tmul_add(telem t2, elem x, elem c) { return fold2_lazy((uint64_t)t2 * x + ((uint64_t)c << 1)); }
    __host__ __device__ static inline elem canon(elem x) { uint32_t d = x - P; return d < x ? d : x; }   // p -> 0
    __host__ __device__ static inline elem sqr(elem a) { return mul(a, a); }
    __host__ __device__ static inline elem pow_u64(elem a, uint64_t e) {
        elem r = 1;
        for (int i = 63; i >= 0; --i) { r = sqr(r); if ((e >> i) & 1) r = mul(r, a); }
        return r;
    }
    __host__ __device__ static inline elem inv(elem a) { return pow_u64(a, P - 2); }
    __host__ static inline bool sqrt(elem a, elem* out) {
        elem r = pow_u64(a, ((uint64_t)P + 1) / 4);
        if (sqr(r) != a) return false;
        *out = r; return true;
    }
    __host__ __device__ static inline elem to_mont(elem a) { return a; }
    __host__ static inline int cmp(elem a, elem b) { return a < b ? -1 : (a > b ? 1 : 0); }
};

}  // namespace ecfft
