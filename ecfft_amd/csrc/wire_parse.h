// FFTree wire format, READ side — the bounds-checked parse of `impl CanonicalDeserialize for FFTree<F>` (/root/reference/src/fftree.rs:600-660;
// ark-serialize 0.4: Vec = u64 LE length + elements, field element = standard-form integer LE, bool = 1 byte).  Pure host C++, no HIP:
// ecfft_capi.hip (wire_read) builds the device tree from what this returns, and tests/cpp/wire_fuzz.cpp drives the very same code with
// millions of mutated files under AddressSanitizer + UBSan on a machine without a GPU (round 6: the reader takes untrusted input).
// Nothing is copied except the rational maps: every table is a pointer into the caller's buffer, checked to lie inside it.
#pragma once
#include <stddef.h>
#include <stdint.h>
#include <string.h>
#include <vector>
#include "../../include/ecfft_hip.h"

namespace ecfft {
namespace wire {

// cursor over the file: every read is bounds-checked (a truncated or corrupt file is ECFFT_ERR_BAD_ARG, never a wild read)
struct In {
    const uint8_t* p; size_t left;
    bool u64(uint64_t* v) {
        if (left < 8) return false;
        uint64_t r = 0; for (int i = 0; i < 8; ++i) r |= (uint64_t)p[i] << (8 * i);
        p += 8; left -= 8; *v = r; return true;
    }
    const uint8_t* bytes(size_t n) {
        if (left < n) return nullptr;
        const uint8_t* r = p; p += n; left -= n; return r;
    }
};

inline size_t elem_bytes(int field) { return field == ECFFT_FIELD_SECP256K1 ? 32 : (field == ECFFT_FIELD_M31 ? 4 : 0); }
// 32-bit limb w of p = 2^256 - 2^32 - 977 (little endian)
inline uint32_t secp_p_limb(int w) { return w == 0 ? 0xFFFFFC2Fu : (w == 1 ? 0xFFFFFFFEu : 0xFFFFFFFFu); }

// every element < p (ark-serialize rejects non-canonical encodings)
inline bool canonical_elems(int field, const uint8_t* p, size_t cnt) {
    for (size_t i = 0; i < cnt; ++i) {
        if (field == ECFFT_FIELD_SECP256K1) {
            uint32_t l[8]; memcpy(l, p + 32 * i, 32);
            bool lt = false;
            for (int w = 7; w >= 0; --w) { const uint32_t pl = secp_p_limb(w); if (l[w] != pl) { lt = l[w] < pl; break; } }
            if (!lt) return false;
        } else {
            uint32_t v; memcpy(&v, p + 4 * i, 4);
            if (v >= 0x7FFFFFFFu) return false;
        }
    }
    return true;
}

// one level of a parsed file: pointers into the file for every table (standard form), in ECFFT_TBL_* order
struct Level { size_t n = 0; const uint8_t* tbl[11] = {}; size_t cnt[11] = {}; };
// a rational map as the file holds it: 3 numerator + 3 denominator coefficients (low -> high, zero padded), raw standard-form bytes
struct Map { uint8_t num[3][32]; uint8_t den[3][32]; };
struct File { std::vector<Level> levels; std::vector<Map> maps; };     // maps: those of the TOP tree (a subtree keeps its parent's first ones)

inline unsigned ilog2_sz(size_t v) { unsigned r = 0; while (v >>= 1) ++r; return r; }

// ECFFT_OK: `out` describes a structurally valid file (lengths, canonical elements, map shapes, subtree chain, no trailing bytes);
// ECFFT_ERR_NOT_POW2: an `f` whose length is not a power of two; ECFFT_ERR_BAD_ARG: everything else.
inline int parse(int field, const uint8_t* data, size_t len, int compress, File& out) {
    const size_t eb = elem_bytes(field);
    if (!eb || !data) return ECFFT_ERR_BAD_ARG;
    In in{data, len};
    out.levels.clear(); out.maps.clear();
    for (size_t expect = 0;; expect >>= 1) {
        Level lv;
        auto vec = [&](int which, size_t per_entry, size_t want_entries, bool check_len) -> bool {
            uint64_t n_ent = 0;
            if (!in.u64(&n_ent)) return false;
            if (check_len && n_ent != want_entries) return false;
            if (n_ent > in.left / (per_entry * eb)) return false;
            const uint8_t* q = in.bytes((size_t)n_ent * per_entry * eb);
            if (!q || !canonical_elems(field, q, (size_t)n_ent * per_entry)) return false;
            lv.tbl[which] = q; lv.cnt[which] = (size_t)n_ent * per_entry;
            return true;
        };
        if (!vec(ECFFT_TBL_F, 1, 0, false)) return ECFFT_ERR_BAD_ARG;
        const size_t two_m = lv.cnt[ECFFT_TBL_F];
        if (two_m < 2 || (two_m & (two_m - 1))) return two_m >= 2 ? ECFFT_ERR_NOT_POW2 : ECFFT_ERR_BAD_ARG;
        const size_t m = two_m / 2;
        if (out.levels.empty()) expect = m; else if (m != expect) return ECFFT_ERR_BAD_ARG;
        lv.n = m;
        const unsigned lm = ilog2_sz(m);
        if (!vec(ECFFT_TBL_RECOMBINE, 4, m, true) || !vec(ECFFT_TBL_DECOMPOSE, 4, m, true)) return ECFFT_ERR_BAD_ARG;
        uint64_t nmaps = 0;
        if (!in.u64(&nmaps) || nmaps != lm) return ECFFT_ERR_BAD_ARG;
        for (unsigned k = 0; k < lm; ++k) {
            Map mp; memset(&mp, 0, sizeof(mp));
            for (int side = 0; side < 2; ++side) {
                uint64_t nc = 0;
                if (!in.u64(&nc) || nc > 3) return ECFFT_ERR_BAD_ARG;
                const uint8_t* q = in.bytes((size_t)nc * eb);
                if (!q || !canonical_elems(field, q, (size_t)nc)) return ECFFT_ERR_BAD_ARG;
                for (uint64_t i = 0; i < nc; ++i) memcpy(side ? mp.den[i] : mp.num[i], q + i * eb, eb);
            }
            bool den2_zero = true;
            for (size_t b = 0; b < eb; ++b) den2_zero = den2_zero && mp.den[2][b] == 0;
            if (!den2_zero) return ECFFT_ERR_BAD_ARG;                       // x-map denominators have degree 1
            if (out.levels.empty()) out.maps.push_back(mp);
            else if (k >= out.maps.size() || memcmp(&out.maps[k], &mp, sizeof(mp)) != 0) return ECFFT_ERR_BAD_ARG;   // a subtree keeps its parent's first maps
        }
        const size_t e = m / 2, zz = m > 1 ? m : 0;
        if (!vec(ECFFT_TBL_XNN_S, 1, m, true) || !vec(ECFFT_TBL_Z0_S1, 1, e, true) || !vec(ECFFT_TBL_Z1_S0, 1, e, true)) return ECFFT_ERR_BAD_ARG;
        if (!compress && (!vec(ECFFT_TBL_XNN_S_INV, 1, m, true) || !vec(ECFFT_TBL_Z0_INV_S1, 1, e, true) || !vec(ECFFT_TBL_Z1_INV_S0, 1, e, true))) return ECFFT_ERR_BAD_ARG;
        if (!vec(ECFFT_TBL_Z0Z0_REM_XNN_S, 1, zz, true) || !vec(ECFFT_TBL_Z1Z1_REM_XNN_S, 1, zz, true)) return ECFFT_ERR_BAD_ARG;
        const uint8_t* hs = in.bytes(1);
        if (!hs || *hs > 1) return ECFFT_ERR_BAD_ARG;
        out.levels.push_back(lv);
        if (!*hs) { if (m != 1) return ECFFT_ERR_BAD_ARG; break; }
        if (m == 1) return ECFFT_ERR_BAD_ARG;
    }
    if (in.left != 0) return ECFFT_ERR_BAD_ARG;                            // trailing bytes
    return ECFFT_OK;
}

}  // namespace wire
}  // namespace ecfft
