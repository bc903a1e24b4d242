#!/usr/bin/env python3
"""Generates the golden fixtures under tests/golden/ with an INDEPENDENT pure-Python big-integer
implementation (no code shared with oracle/ or ecfft_amd/).

Why this exists: the reference (andrewmilson/ecfft, Rust + un-vendored ark-* crates) cannot be
built in this image, and its own tests hold no literal ENTER/EXIT vectors — they assert identities
against naive polynomial evaluation on the FFTree's leaves (/root/reference/src/lib.rs:108-152,
239-278).  This script restates exactly those expected sides:

  * leaves      x(coset_offset + i*G_n) from the curve constants of src/lib.rs:45-59 (secp256k1)
                and src/lib.rs:201-206 (Mersenne-31),
  * ENTER       naive Horner evaluation of seeded random coefficients on the leaves,
  * EXTEND      naive evaluation of a degree < n/2 polynomial on S0 (even leaves) and S1 (odd leaves),
  * EXIT        the coefficients that ENTER started from,
  * tables      <Z_0 | S_1>, <Z_1 | S_0>, <X^(n/2) | S> and <Z_0^2 mod X^(n/2) | S> from their product
                definitions (src/fftree.rs:30-37) at n = 64,
  * the isogeny x-maps: secp256k1 psi(x) = (x - b)^2 / x with b the ark-ff square root
                bb^((p+1)/4) (src/ec.rs:42, 84); M31 first map (x^2 + 1)/x (Velu, src/ec.rs:230-236).

Values are stored in STANDARD form (plain integers): secp256k1 as uint64[n,4] little-endian limbs,
M31 as uint32[n].  Run:  python tests/golden/gen_golden.py   (takes ~1 min; n=4096 naive is O(n^2)).
"""
import os
import random

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))


# --------------------------------------------------------------------------- generic EC
def ec_add(P, Q, p, a1, a2, a3, a4, a6):
    """Silverman III.2.3 group law on y^2 + a1xy + a3y = x^3 + a2x^2 + a4x + a6; None = infinity."""
    if P is None:
        return Q
    if Q is None:
        return P
    x1, y1 = P
    x2, y2 = Q
    if x1 == x2 and (y1 + y2 + a1 * x2 + a3) % p == 0:
        return None
    if x1 == x2:
        den = pow((2 * y1 + a1 * x1 + a3) % p, -1, p)
        lam = (3 * x1 * x1 + 2 * a2 * x1 + a4 - a1 * y1) * den % p
        nu = (-x1 ** 3 + a4 * x1 + 2 * a6 - a3 * y1) * den % p
    else:
        den = pow((x2 - x1) % p, -1, p)
        lam = (y2 - y1) * den % p
        nu = (y1 * x2 - y2 * x1) * den % p
    x3 = (lam * lam + a1 * lam - a2 - x1 - x2) % p
    y3 = (-(lam + a1) * x3 - nu - a3) % p
    return (x3, y3)


def leaves_of(offset, gen, n, p, curve):
    out, acc = [], None
    for _ in range(n):
        out.append(ec_add(offset, acc, p, *curve)[0])
        acc = ec_add(acc, gen, p, *curve)
    return out


def order_is_pow2(P, p, curve):
    k = 0
    while P is not None:
        P = ec_add(P, P, p, *curve)
        k += 1
        assert k < 64
    return k


# --------------------------------------------------------------------------- secp256k1
SECP_P = 2 ** 256 - 2 ** 32 - 977
SECP_A = 31172306031375832341232376275243462303334845584808513005362718476441963632613
SECP_BB = 45508371059383884471556188660911097844526467659576498497548207627741160623272
SECP_OFF = (105623886150579165427389078198493427091405550492761682382732004625374789850161,
            7709812624542158994629670452026922591039826164720902911013234773380889499231)
SECP_GEN = (41293412487153066667050767300223451435019201659857889215769525847559135483332,
            73754924733368840065089190002333366411120578552679996887076912271884749237510)


def secp_tree_params(log_n):
    p = SECP_P
    a, b = SECP_A, pow(SECP_BB, (p + 1) // 4, p)
    assert b * b % p == SECP_BB
    curve = (0, a, 0, b * b % p, 0)
    g = SECP_GEN
    for _ in range(36 - log_n):
        g = ec_add(g, g, p, *curve)
    assert order_is_pow2(g, p, curve) == log_n
    leaves = leaves_of(SECP_OFF, g, 1 << log_n, p, curve)
    bs = []
    for _ in range(log_n):  # good isogeny chain: a' = a + 6b, B' = 4ab + 8b^2, psi = (x-b)^2/x
        bs.append(b)
        a, B = (a + 6 * b) % p, (4 * a * b + 8 * b * b) % p
        b = pow(B, (p + 1) // 4, p)
        assert b * b % p == B, "codomain B' must be a square"
    maps = [([bb_ * bb_ % p, (-2 * bb_) % p, 1], [0, 1, 0]) for bb_ in bs]
    return p, leaves, maps


# --------------------------------------------------------------------------- M31
M31_P = 2 ** 31 - 1


def m31_tree_params(log_n):
    """Only the leaves and the FIRST map are derived independently here (the map choice needs cubic
    root finding, which the oracle restates and tests separately); deeper layers are checked by the
    2-to-1 property in tests."""
    p = M31_P
    curve = (0, 0, 0, 1, 0)
    off, g = (1048755163, 279503108), (1273083559, 804329170)
    for _ in range(28 - log_n):
        g = ec_add(g, g, p, *curve)
    assert order_is_pow2(g, p, curve) == log_n
    return p, leaves_of(off, g, 1 << log_n, p, curve)


# --------------------------------------------------------------------------- naive polynomial side
def horner(c, x, p):
    r = 0
    for a in reversed(c):
        r = (r * x + a) % p
    return r


def poly_mul(a, b, p):
    r = [0] * (len(a) + len(b) - 1)
    for i, x in enumerate(a):
        if x:
            for j, y in enumerate(b):
                r[i + j] = (r[i + j] + x * y) % p
    return r


def poly_from_roots(roots, p):
    r = [1]
    for s in roots:
        r = poly_mul(r, [(-s) % p, 1], p)
    return r


def pack(vals, limbs):
    if limbs == 1:
        return np.array(vals, dtype=np.uint32)
    out = np.zeros((len(vals), limbs), dtype=np.uint64)
    for i, v in enumerate(vals):
        for l in range(limbs):
            out[i, l] = (v >> (64 * l)) & 0xFFFFFFFFFFFFFFFF
    return out


def vectors(name, p, leaves, limbs, seed):
    n = len(leaves)
    rng = random.Random(seed)
    coeffs = [rng.randrange(p) for _ in range(n)]
    evals = [horner(coeffs, x, p) for x in leaves]
    half = [rng.randrange(p) for _ in range(n // 2)] if n > 1 else []
    s0, s1 = leaves[0::2], leaves[1::2]
    out = {
        "leaves": pack(leaves, limbs),
        "enter_coeffs": pack(coeffs, limbs),
        "enter_evals": pack(evals, limbs),
        "extend_coeffs": pack(half, limbs),
        "extend_s0": pack([horner(half, x, p) for x in s0], limbs),
        "extend_s1": pack([horner(half, x, p) for x in s1], limbs),
    }
    print(f"  {name} n={n}: done")
    return out


def tables(p, leaves, limbs):
    """Product-definition tables of the tree with n = len(leaves) leaves (src/fftree.rs:30-37)."""
    n = len(leaves)
    s0, s1 = leaves[0::2], leaves[1::2]
    z0, z1 = poly_from_roots(s0, p), poly_from_roots(s1, p)
    z0z0 = poly_mul(z0, z0, p)[: n // 2]  # Z_0^2 mod X^(n/2)
    z1z1 = poly_mul(z1, z1, p)[: n // 2]
    return {
        "xnn_s": pack([pow(x, n // 2, p) for x in leaves], limbs),
        "z0_s1": pack([horner(z0, x, p) for x in s1], limbs),
        "z1_s0": pack([horner(z1, x, p) for x in s0], limbs),
        "z0z0_rem_xnn_s": pack([horner(z0z0, x, p) for x in leaves], limbs),
        "z1z1_rem_xnn_s": pack([horner(z1z1, x, p) for x in leaves], limbs),
    }


def main():
    # secp256k1
    for log_n in (2, 6, 12):
        p, leaves, maps = secp_tree_params(log_n)
        d = vectors("secp256k1", p, leaves, 4, 0x5EED0000 + log_n)
        d["map_num"] = pack([c for m in maps for c in m[0]], 4)
        d["map_den"] = pack([c for m in maps for c in m[1]], 4)
        if log_n == 6:
            d.update({"tbl_" + k: v for k, v in tables(p, leaves, 4).items()})
            # subtree with 16 leaves = every 4th leaf (src/fftree.rs:471-478)
            d.update({"tbl16_" + k: v for k, v in tables(p, leaves[::4], 4).items()})
        np.savez_compressed(os.path.join(HERE, f"secp256k1_n{1 << log_n}.npz"), **d)
    # Mersenne-31
    for log_n in (2, 6, 12):
        p, leaves = m31_tree_params(log_n)
        d = vectors("m31", p, leaves, 1, 0x5EED1000 + log_n)
        d["map0_num"] = pack([1, 0, 1], 1)
        d["map0_den"] = pack([0, 1, 0], 1)
        if log_n == 6:
            d.update({"tbl_" + k: v for k, v in tables(p, leaves, 1).items()})
        np.savez_compressed(os.path.join(HERE, f"m31_n{1 << log_n}.npz"), **d)


if __name__ == "__main__":
    main()
