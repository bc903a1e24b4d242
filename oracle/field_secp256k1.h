/* TEST INFRASTRUCTURE — CPU oracle, not product code.
 *
 * secp256k1 base field Fp in the in-memory representation of the reference's
 *   `pub type Fp = Fp256<MontBackend<FqConfig, 4>>`            (/root/reference/src/lib.rs:31-37)
 * i.e. four little-endian u64 limbs holding x * 2^256 mod p, always fully reduced.
 * The arithmetic itself lives in the un-vendored crate ark-ff 0.4 (Cargo.toml:23); this is a
 * restatement of its published algorithm (Montgomery CIOS multiplication with a final
 * conditional subtraction; the modulus has no spare bit so the "no-carry" shortcut does not
 * apply), anchored on the call sites in src/utils.rs:341-346 and src/fftree.rs:94,115,157-158.
 *
 * p = 2^256 - 2^32 - 977.
 */
#ifndef ORACLE_FIELD_SECP256K1_H
#define ORACLE_FIELD_SECP256K1_H
#include <stdint.h>
#include <string.h>
#include <stdlib.h>

typedef unsigned __int128 u128;
typedef struct { uint64_t l[4]; } fe;

#define FE_BYTES 32
#define FIELD_NAME "secp256k1"

static const uint64_t FE_P[4] = {0xFFFFFFFEFFFFFC2FULL, 0xFFFFFFFFFFFFFFFFULL, 0xFFFFFFFFFFFFFFFFULL,
                                 0xFFFFFFFFFFFFFFFFULL};
#define FE_PINV 0xD838091DD2253531ULL /* -p^{-1} mod 2^64 */
static const fe FE_R = {{0x00000001000003D1ULL, 0, 0, 0}};  /* 2^256 mod p  = Montgomery 1 */
static const fe FE_R2 = {{0x000007A2000E90A1ULL, 1, 0, 0}}; /* 2^512 mod p */

static inline fe fe_zero(void) { fe r = {{0, 0, 0, 0}}; return r; }
static inline fe fe_one(void) { return FE_R; }
static inline int fe_is_zero(fe a) { return (a.l[0] | a.l[1] | a.l[2] | a.l[3]) == 0; }
static inline int fe_eq(fe a, fe b) { return memcmp(&a, &b, sizeof(fe)) == 0; }

/* r = a - p if a >= p (carry = bit 256 of a) */
static inline fe fe_cond_sub_p(fe a, uint64_t carry) {
    fe r; u128 bw = 0;
    for (int i = 0; i < 4; ++i) {
        u128 d = (u128)a.l[i] - FE_P[i] - (uint64_t)bw;
        r.l[i] = (uint64_t)d; bw = (d >> 64) & 1;
    }
    /* a >= p  <=>  no final borrow, or the 257th bit was set */
    return (carry || !bw) ? r : a;
}
static inline fe fe_add(fe a, fe b) {
    fe r; u128 c = 0;
    for (int i = 0; i < 4; ++i) { c += (u128)a.l[i] + b.l[i]; r.l[i] = (uint64_t)c; c >>= 64; }
    return fe_cond_sub_p(r, (uint64_t)c);
}
static inline fe fe_sub(fe a, fe b) {
    fe r; u128 bw = 0;
    for (int i = 0; i < 4; ++i) {
        u128 d = (u128)a.l[i] - b.l[i] - (uint64_t)bw;
        r.l[i] = (uint64_t)d; bw = (d >> 64) & 1;
    }
    if (bw) { u128 c = 0; for (int i = 0; i < 4; ++i) { c += (u128)r.l[i] + FE_P[i]; r.l[i] = (uint64_t)c; c >>= 64; } }
    return r;
}
static inline fe fe_neg(fe a) { return fe_is_zero(a) ? a : fe_sub(fe_zero(), a); }
static inline fe fe_dbl(fe a) { return fe_add(a, a); }

/* Montgomery product a*b*2^-256 mod p (CIOS) */
static inline fe fe_mul(fe a, fe b) {
    uint64_t t[6] = {0, 0, 0, 0, 0, 0};
    for (int i = 0; i < 4; ++i) {
        u128 c = 0;
        for (int j = 0; j < 4; ++j) { c += (u128)t[j] + (u128)a.l[j] * b.l[i]; t[j] = (uint64_t)c; c >>= 64; }
        c += t[4]; t[4] = (uint64_t)c; t[5] = (uint64_t)(c >> 64);
        uint64_t m = t[0] * FE_PINV;
        c = (u128)t[0] + (u128)m * FE_P[0]; c >>= 64;
        for (int j = 1; j < 4; ++j) { c += (u128)t[j] + (u128)m * FE_P[j]; t[j - 1] = (uint64_t)c; c >>= 64; }
        c += t[4]; t[3] = (uint64_t)c; t[4] = t[5] + (uint64_t)(c >> 64);
    }
    fe r = {{t[0], t[1], t[2], t[3]}};
    return fe_cond_sub_p(r, t[4]);
}
static inline fe fe_sqr(fe a) { return fe_mul(a, a); }

static inline fe fe_pow_u64(fe a, uint64_t e) { /* ark-ff Field::pow: square-and-multiply, MSB first */
    fe r = fe_one();
    for (int i = 63; i >= 0; --i) { r = fe_sqr(r); if ((e >> i) & 1) r = fe_mul(r, a); }
    return r;
}
static inline fe fe_pow_limbs(fe a, const uint64_t e[4]) {
    fe r = fe_one();
    for (int i = 255; i >= 0; --i) { r = fe_sqr(r); if ((e[i / 64] >> (i % 64)) & 1) r = fe_mul(r, a); }
    return r;
}
/* ark-ff 0.4 `Fp::inverse` for Montgomery elements: binary extended Euclid (Guajardo, Kumar, Paar, Pelzl, Algorithm 16)
 * on the raw limbs with b starting at R^2, so that (aR) -> a^-1 R.  Kept close to the crate's cost (a few microseconds)
 * because the reference's REDC inverts on every call (src/fftree.rs:235) and this file is also the CPU baseline. */
static inline int u256_is_one(const uint64_t x[4]) { return x[0] == 1 && (x[1] | x[2] | x[3]) == 0; }
static inline int u256_lt(const uint64_t a[4], const uint64_t b[4]) {
    for (int i = 3; i >= 0; --i) { if (a[i] < b[i]) return 1; if (a[i] > b[i]) return 0; }
    return 0;
}
static inline void u256_sub(uint64_t a[4], const uint64_t b[4]) {
    u128 bw = 0;
    for (int i = 0; i < 4; ++i) { u128 d = (u128)a[i] - b[i] - (uint64_t)bw; a[i] = (uint64_t)d; bw = (d >> 64) & 1; }
}
static inline void u256_shr1(uint64_t a[4], uint64_t top) {
    a[0] = (a[0] >> 1) | (a[1] << 63); a[1] = (a[1] >> 1) | (a[2] << 63); a[2] = (a[2] >> 1) | (a[3] << 63); a[3] = (a[3] >> 1) | (top << 63);
}
static inline void fe_half(fe* b) { /* b/2 mod p */
    uint64_t carry = 0;
    if (b->l[0] & 1) { u128 c = 0; for (int i = 0; i < 4; ++i) { c += (u128)b->l[i] + FE_P[i]; b->l[i] = (uint64_t)c; c >>= 64; } carry = (uint64_t)c; }
    u256_shr1(b->l, carry);
}
static inline fe fe_inv(fe a) { /* a != 0 */
    uint64_t u[4], v[4];
    memcpy(u, a.l, 32); memcpy(v, FE_P, 32);
    fe b = FE_R2, c = fe_zero();
    while (!u256_is_one(u) && !u256_is_one(v)) {
        while (!(u[0] & 1)) { u256_shr1(u, 0); fe_half(&b); }
        while (!(v[0] & 1)) { u256_shr1(v, 0); fe_half(&c); }
        if (u256_lt(v, u)) { u256_sub(u, v); b = fe_sub(b, c); }
        else { u256_sub(v, u); c = fe_sub(c, b); }
    }
    return u256_is_one(u) ? b : c;
}
/* ark-ff sqrt for p = 3 mod 4: candidate a^((p+1)/4), accepted iff its square is a. returns 1 if QR */
static inline int fe_sqrt(fe a, fe* out) {
    static const uint64_t e[4] = {0xFFFFFFFFBFFFFF0CULL, ~0ULL, ~0ULL, 0x3FFFFFFFFFFFFFFFULL};
    fe r = fe_pow_limbs(a, e);
    if (!fe_eq(fe_sqr(r), a)) return 0;
    *out = r; return 1;
}
static inline fe fe_from_u64(uint64_t v) { fe r = {{v, 0, 0, 0}}; return fe_mul(r, FE_R2); }
/* standard-form little-endian 4xu64 <-> Montgomery */
static inline fe fe_from_std(const uint64_t s[4]) { fe r = {{s[0], s[1], s[2], s[3]}}; return fe_mul(r, FE_R2); }
static inline void fe_to_std(fe a, uint64_t s[4]) {
    fe one = {{1, 0, 0, 0}}; fe r = fe_mul(a, one); memcpy(s, r.l, 32);
}
/* canonical ordering used by `roots.sort()` (ark-ff Ord compares the standard-form integer) */
static inline int fe_cmp(fe a, fe b) {
    uint64_t x[4], y[4]; fe_to_std(a, x); fe_to_std(b, y);
    for (int i = 3; i >= 0; --i) { if (x[i] < y[i]) return -1; if (x[i] > y[i]) return 1; }
    return 0;
}
#endif
