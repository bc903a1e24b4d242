#!/usr/bin/env python3
"""Paper kill of "overflow-free table multiplies through short-limbed constants" (round 6).

The 53 v_addc_co_u32 that count the carry-outs of the 64-bit column accumulators are 31 % of the 169-instruction secp256k1 table
multiply.  They would vanish if every constant limb were < 2^29 (8 products of < 2^61 plus a 32-bit addend fit a 64-bit pair), and a
constant may be stored in ANY form: as m > 8 radix-2^32 limbs c_i with sum c_i 2^(32 i) = c (mod p), taken from the lattice
L = { v in Z^m : sum v_i 2^(32 i) = 0 (mod p) } by a closest-vector step at table-build time (10 limbs of <= 2^29: 80 + 6 mads and no
carry counting, ~125 instead of 169 instructions).  For a generic 256-bit prime L is random-like, det^(1/m) = 2^25.6 at m = 10, and
it works.  For secp256k1 it does not: 2^256 = 2^32 + 977 (mod p) puts (977, 1, 0, 0, 0, 0, 0, 0, -1, 0, ...) and its shifts into L —
the limbs above 2^256 are worth 2^10, not 2^32, the other eight directions keep successive minima of ~2^32, and the reduced basis
below shows per-limb ranges W_i = sum_k |b_k[i]| of 2^31 .. 2^32 for limbs 2..7 whatever m is.  The pseudo-Mersenne shape that makes
the fold cheap is what makes extra limbs worthless.  usage: limb_lattice.py > profiles/r06/limb_lattice.txt"""
import math
from fractions import Fraction
p = 2**256 - 2**32 - 977


def lll(B, delta=Fraction(99, 100)):
    n = len(B); B = [list(r) for r in B]
    dot = lambda a, b: sum(x * y for x, y in zip(a, b))

    def gs():
        Bs = []; mu = [[Fraction(0)] * n for _ in range(n)]
        for i in range(n):
            v = [Fraction(x) for x in B[i]]
            for j in range(i):
                mu[i][j] = dot(B[i], Bs[j]) / dot(Bs[j], Bs[j])
                v = [a - mu[i][j] * b for a, b in zip(v, Bs[j])]
            Bs.append(v)
        return Bs, mu
    Bs, mu = gs(); k = 1
    while k < n:
        for j in range(k - 1, -1, -1):
            q = round(mu[k][j])
            if q:
                B[k] = [a - q * b for a, b in zip(B[k], B[j])]; Bs, mu = gs()
        if dot(Bs[k], Bs[k]) >= (delta - mu[k][k - 1] ** 2) * dot(Bs[k - 1], Bs[k - 1]):
            k += 1
        else:
            B[k], B[k - 1] = B[k - 1], B[k]; Bs, mu = gs(); k = max(k - 1, 1)
    return B


for m in (9, 10, 11, 12):
    B = [[0] * m for _ in range(m)]
    B[0][0] = p
    for i in range(1, m):
        B[i][0] = -pow(2, 32 * i, p); B[i][i] = 1
    R = lll(B)
    assert all(sum(v * 2 ** (32 * i) for i, v in enumerate(r)) % p == 0 for r in R)
    W = [sum(abs(R[k][i]) for k in range(m)) for i in range(m)]
    print(f"m = {m}: det^(1/m) = 2^{256 / m:.1f}; log2 of the per-limb range W_i of a reduced basis:", " ".join(f"{math.log2(w):.1f}" for w in W))
    print("   shortest basis vectors:", *[r for r in sorted(R, key=lambda r: max(map(abs, r)))[:2]])
