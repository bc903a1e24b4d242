/* TEST INFRASTRUCTURE — CPU oracle, not product code.
 *
 * Mersenne-31 field, the representation of `ark_ff_optimized::fp31::Fp` re-exported at
 * /root/reference/src/lib.rs:196: a tuple struct over one u32 holding the PLAIN canonical residue
 * (the constants at src/lib.rs:203-205 are written as literal `Fp(1048755163)`).  The crate
 * (ark-ff-optimized 0.4, Cargo.toml:24) is not vendored; this restates p = 2^31 - 1 arithmetic.
 */
#ifndef ORACLE_FIELD_M31_H
#define ORACLE_FIELD_M31_H
#include <stdint.h>
#include <string.h>
#include <stdlib.h>

typedef uint32_t fe;
#define FE_BYTES 4
#define FIELD_NAME "m31"
#define M31_P 0x7FFFFFFFu

static inline fe fe_zero(void) { return 0; }
static inline fe fe_one(void) { return 1; }
static inline int fe_is_zero(fe a) { return a == 0; }
static inline int fe_eq(fe a, fe b) { return a == b; }
static inline fe fe_add(fe a, fe b) { uint32_t s = a + b; return s >= M31_P ? s - M31_P : s; }
static inline fe fe_sub(fe a, fe b) { return a >= b ? a - b : a + M31_P - b; }
static inline fe fe_neg(fe a) { return a ? M31_P - a : 0; }
static inline fe fe_dbl(fe a) { return fe_add(a, a); }
static inline fe fe_mul(fe a, fe b) {
    uint64_t t = (uint64_t)a * b;
    uint32_t r = (uint32_t)(t & M31_P) + (uint32_t)(t >> 31);
    return r >= M31_P ? r - M31_P : r;
}
static inline fe fe_sqr(fe a) { return fe_mul(a, a); }
static inline fe fe_pow_u64(fe a, uint64_t e) {
    fe r = 1;
    for (int i = 63; i >= 0; --i) { r = fe_sqr(r); if ((e >> i) & 1) r = fe_mul(r, a); }
    return r;
}
static inline fe fe_inv(fe a) { return fe_pow_u64(a, M31_P - 2); }
static inline int fe_sqrt(fe a, fe* out) { /* p = 3 mod 4 */
    fe r = fe_pow_u64(a, ((uint64_t)M31_P + 1) / 4);
    if (fe_sqr(r) != a) return 0;
    *out = r; return 1;
}
static inline fe fe_from_u64(uint64_t v) { return (fe)(v % M31_P); }
static inline int fe_cmp(fe a, fe b) { return a < b ? -1 : (a > b ? 1 : 0); }
#endif
