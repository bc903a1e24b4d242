"""Directed tests of the DEVICE field arithmetic (hand-written gfx950 multiply, asm subtraction, M31 reduction) through
ecfft_selftest_field: operands chosen so that results land next to p and 2^256, the second fold carries out, and the
canonicalisation branch is taken — cases that uniformly random data reaches with probability < 2^-32 per lane."""
import random

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

P256 = 2**256 - 2**32 - 977
C = 2**32 + 977
P31 = 2**31 - 1


def pack256(vals):
    out = np.zeros((len(vals), 4), dtype=np.uint64)
    for i, v in enumerate(vals):
        for l in range(4):
            out[i, l] = (v >> (64 * l)) & 0xFFFFFFFFFFFFFFFF
    return out


def unpack256(a):
    return [sum(int(a[i, l]) << (64 * l) for l in range(4)) for i in range(a.shape[0])]


def directed_triples():
    rnd = random.Random(7)
    T = []
    edge = [0, 1, 2, C, C - 1, C + 1, P256 - 1, P256 - 2, P256 - C, 2**255, 2**255 - 1, 2**128, 2**128 - 1, 2**224, (1 << 256) - C - 1 - 977,
            0xFFFFFFFF, 0xFFFFFFFF00000000, (2**256 - 1) // 3, P256 // 2, P256 // 2 + 1]
    for a in edge:
        for b in edge:
            T.append((a, b, rnd.choice(edge)))
    # results that are exactly p-1, 0, 1 (value before canonicalisation in [p, 2^256) or just below p)
    for _ in range(200):
        t = rnd.randrange(1, P256)
        for target in (P256 - 1, 0, 1, P256 - rnd.randrange(1, 2**40), rnd.randrange(0, 2**40)):
            x = target * pow(t, -1, P256) % P256
            T.append((t, x, 0))
            c = rnd.randrange(P256)
            x2 = (target - c) * pow(t, -1, P256) % P256
            T.append((t, x2, c))
    # carry-out of the second fold: t = 2^255, x even -> V = (x/2)*C just below a multiple of 2^256
    for j in range(2, 400, 7):
        half = (j * 2**256 - 1) // C
        x = 2 * half
        if x < P256:
            T.append((2**255, x, 0))
            T.append((2**255, x, P256 - 1))
    # high words all ones before the fold
    for _ in range(200):
        T.append((P256 - rnd.randrange(1, 2**33), P256 - rnd.randrange(1, 2**33), P256 - rnd.randrange(1, 2**33)))
    for _ in range(3000):
        T.append((rnd.randrange(P256), rnd.randrange(P256), rnd.randrange(P256)))
    return T


def test_secp_mul_add_directed(hooks_lib):
    import ecfft_amd
    F = ecfft_amd.secp256k1
    T = directed_triples()
    a, b, c = pack256([t[0] for t in T]), pack256([t[1] for t in T]), pack256([t[2] for t in T])
    got = unpack256(F.selftest(0, a, b, c))
    assert got == [(x * y + z) % P256 for x, y, z in T]
    got = unpack256(F.selftest(1, a, b))
    assert got == [(x * y) % P256 for x, y, z in T]


def test_secp_table_multiply_directed(hooks_lib):
    """the two-table (t, t*2^128) multiply the butterfly kernels use: canonical results on the same directed operands"""
    import ecfft_amd
    F = ecfft_amd.secp256k1
    T = directed_triples()
    a, b, c = pack256([t[0] for t in T]), pack256([t[1] for t in T]), pack256([t[2] for t in T])
    assert unpack256(F.selftest(4, a, b, c)) == [(x * y + z) % P256 for x, y, z in T]
    assert unpack256(F.selftest(5, a, b)) == [(x * y) % P256 for x, y, z in T]


def test_secp_add_sub_directed(hooks_lib):
    import ecfft_amd
    F = ecfft_amd.secp256k1
    T = directed_triples()
    a, b = pack256([t[0] for t in T]), pack256([t[1] for t in T])
    assert unpack256(F.selftest(2, a, b)) == [(x - y) % P256 for x, y, z in T]
    assert unpack256(F.selftest(3, a, b)) == [(x + y) % P256 for x, y, z in T]


def test_m31_directed(hooks_lib):
    import ecfft_amd
    F = ecfft_amd.m31
    rnd = random.Random(3)
    edge = [0, 1, 2, P31 - 1, P31 - 2, 2**30, 2**30 - 1, 2**16, 46341, 46340, 65535, 65536]
    T = [(a, b, c) for a in edge for b in edge for c in (0, 1, P31 - 1)]
    T += [(rnd.randrange(P31), rnd.randrange(P31), rnd.randrange(P31)) for _ in range(20000)]
    a = np.array([t[0] for t in T], dtype=np.uint32); b = np.array([t[1] for t in T], dtype=np.uint32); c = np.array([t[2] for t in T], dtype=np.uint32)
    assert [int(v) for v in F.selftest(0, a, b, c)] == [(x * y + z) % P31 for x, y, z in T]
    assert [int(v) for v in F.selftest(1, a, b)] == [(x * y) % P31 for x, y, z in T]
    assert [int(v) for v in F.selftest(2, a, b)] == [(x - y) % P31 for x, y, z in T]
    assert [int(v) for v in F.selftest(3, a, b)] == [(x + y) % P31 for x, y, z in T]


def test_m31_lazy_table_multiply(hooks_lib):
    """M31's kernel-internal multiply keeps values in [0, p] (p = second representative of 0): table constant canonical,
    data and addend anywhere in [0, p], including the extreme (p-1)*p + p = p^2 that the range proof rests on"""
    import ecfft_amd
    F = ecfft_amd.m31
    rnd = random.Random(5)
    edge_t = [0, 1, 2, P31 - 1, P31 - 2, 2**30, 2**30 + 1, 65535, 65536]
    edge_x = [0, 1, 2, P31, P31 - 1, P31 - 2, 2**30, 2**30 - 1, 2**16]
    T = [(a, b, c) for a in edge_t for b in edge_x for c in edge_x]
    T += [(rnd.randrange(P31), rnd.randrange(P31 + 1), rnd.randrange(P31 + 1)) for _ in range(50000)]
    a = np.array([t[0] for t in T], dtype=np.uint32); b = np.array([t[1] for t in T], dtype=np.uint32); c = np.array([t[2] for t in T], dtype=np.uint32)
    for op in (4, 5):
        got = [int(v) for v in F.selftest(op, a, b, c)]
        assert max(got) <= P31
        want = [(x * y + (z if op == 4 else 0)) % P31 for x, y, z in T]
        assert [g % P31 for g in got] == want
    # sub on the lazy range stays in the lazy range
    aa = np.array([0, P31, 0, P31, 5, P31 - 1], dtype=np.uint32); bb = np.array([P31, 0, 0, P31, P31, P31], dtype=np.uint32)
    d = [int(v) for v in F.selftest(2, aa, bb)]
    assert max(d) <= P31 and [v % P31 for v in d] == [(int(x) - int(y)) % P31 for x, y in zip(aa, bb)]


def _blk16(T, xs, mode=0):
    """ecfft_selftest_blk16 (mode 0: the 32x32x32 form on 1024-element tiles) / ecfft_selftest_blk16_small (modes 1..4: the
    16x16x64 forms of the small-launch kernels): T = 16 x 16 ints, xs = ints (multiple of 1024) -> ints"""
    from ecfft_amd import fftree as FT
    m = pack256([T[o][i] for o in range(16) for i in range(16)])
    x = pack256(xs)
    out = np.zeros_like(x)
    if mode == 0:
        assert FT.lib().ecfft_selftest_blk16(m.ctypes.data, x.ctypes.data, out.ctypes.data, len(xs), 0) == 0
    else:
        assert FT.lib().ecfft_selftest_blk16_small(m.ctypes.data, x.ctypes.data, out.ctypes.data, len(xs), mode, 0) == 0
    return unpack256(out)


BLK16_FORMS = [0, 1, 2, 3, 4]


@pytest.mark.parametrize("mode", BLK16_FORMS)
def test_blk16_matrix_core_map_directed(mode, hooks_lib):
    """the matrix-core form of the innermost 16-point map (mfma_blk16.h) with explicit constants: identity, -1, 0 and small
    constants put the pre-reduction value next to multiples of 2^256, so the carry-out of the fold and the canonicalisation branch
    of the normalisation run (with a tree's random-looking constants they have probability ~2^-200); results must be canonical"""
    rnd = random.Random(11)
    edge = [0, 1, 2, C - 1, C, C + 1, 2**32, 2**50, 2**50 - 1, 2**51, 2**64, 2**128, 2**255, P256 - 1, P256 - 2, P256 - C, P256 - C - 1, P256 // 2,
            0x7F7F7F7F7F7F7F7F7F7F7F7F7F7F7F7F7F7F7F7F7F7F7F7F7F7F7F7F7F7F7F7F, 0x8080808080808080808080808080808080808080808080808080808080808080 % P256]
    xs = (edge * 52)[:1024]
    xs[512:] = [rnd.randrange(P256) for _ in range(512)]
    ident = [[1 if o == i else 0 for i in range(16)] for o in range(16)]
    assert _blk16(ident, xs, mode) == xs                                            # out = x: every edge value comes back canonical
    neg = [[P256 - 1 if o == i else 0 for i in range(16)] for o in range(16)]
    assert _blk16(neg, xs, mode) == [(-v) % P256 for v in xs]
    zero = [[0] * 16 for _ in range(16)]
    assert _blk16(zero, xs, mode) == [0] * 1024
    ones = [[1] * 16 for _ in range(16)]                                       # every output = sum of the block
    want = []
    for b in range(0, 1024, 16):
        want += [sum(xs[b:b + 16]) % P256] * 16
    assert _blk16(ones, xs, mode) == want
    # constants at the signed-digit recoding threshold and its neighbours, in every position of one row
    thr = 0x7F7F7F7F7F7F7F7F7F7F7F7F7F7F7F7F7F7F7F7F7F7F7F7F7F7F7F7F7F7F7F7F
    row = [thr, thr + 1, thr - 1, P256 - thr, 2**255, 2**255 - 1, P256 - 2**255, 255, 256, 2**248, 2**248 - 1, P256 - 256, C, P256 - C, 3, P256 - 3]
    Tm = [[row[(o + i) % 16] for i in range(16)] for o in range(16)]
    want = []
    for b in range(0, 1024, 16):
        want += [sum(Tm[o][i] * xs[b + i] for i in range(16)) % P256 for o in range(16)]
    assert _blk16(Tm, xs, mode) == want


@pytest.mark.parametrize("mode", BLK16_FORMS)
def test_blk16_matrix_core_map_random_and_targeted_outputs(mode, hooks_lib):
    """random 16 x 16 maps against big-int arithmetic, plus inputs solved for so that chosen outputs are exactly 0, 1, p-1 and values
    within 2^40 of 0 and p"""
    rnd = random.Random(12)
    for trial in range(3):
        Tm = [[rnd.randrange(P256) for _ in range(16)] for _ in range(16)]
        xs = [rnd.randrange(P256) for _ in range(2048)]
        # targeted: fix x_1..x_15 of a block, solve x_0 so that output 0 hits a chosen value
        inv00 = pow(Tm[0][0], -1, P256)
        targets = [0, 1, P256 - 1, P256 - rnd.randrange(1, 2**40), rnd.randrange(2**40), C, C - 1, P256 - C]
        for k, tgt in enumerate(targets):
            b = 16 * k
            rest = sum(Tm[0][i] * xs[b + i] for i in range(1, 16)) % P256
            xs[b] = (tgt - rest) * inv00 % P256
        got = _blk16(Tm, xs, mode)
        want = []
        for b in range(0, 2048, 16):
            want += [sum(Tm[o][i] * xs[b + i] for i in range(16)) % P256 for o in range(16)]
        assert got == want
        for k, tgt in enumerate(targets):
            assert got[16 * k] == tgt


def _blk32(T, xs):
    """ecfft_selftest_blk32: T = 32 x 32 ints, xs = ints (multiple of 1024) -> ints"""
    from ecfft_amd import fftree as FT
    m = pack256([T[o][i] for o in range(32) for i in range(32)])
    x = pack256(xs)
    out = np.zeros_like(x)
    assert FT.lib().ecfft_selftest_blk32(m.ctypes.data, x.ctypes.data, out.ctypes.data, len(xs), 0) == 0
    return unpack256(out)


def test_blk32_matrix_core_map_directed_and_random(hooks_lib):
    """round 4: the 32-point form (the five lowest ENTER / EXIT levels of the 1024-element low-level kernels as one map,
    Blk16::phase32): identity / -1 / all-ones maps on edge operands (carry-out and canonicalisation branches of the normalisation),
    constants at the signed-digit recoding threshold, random maps against big-int arithmetic and outputs solved for 0, 1, p - 1"""
    rnd = random.Random(32)
    edge = [0, 1, 2, C - 1, C, C + 1, 2**32, 2**50, 2**51, 2**64, 2**128, 2**255, P256 - 1, P256 - 2, P256 - C, P256 - C - 1, P256 // 2,
            0x7F7F7F7F7F7F7F7F7F7F7F7F7F7F7F7F7F7F7F7F7F7F7F7F7F7F7F7F7F7F7F7F, 0x8080808080808080808080808080808080808080808080808080808080808080 % P256]
    xs = (edge * 54)[:1024]
    xs[512:] = [rnd.randrange(P256) for _ in range(512)]
    ident = [[1 if o == i else 0 for i in range(32)] for o in range(32)]
    assert _blk32(ident, xs) == xs
    neg = [[P256 - 1 if o == i else 0 for i in range(32)] for o in range(32)]
    assert _blk32(neg, xs) == [(-v) % P256 for v in xs]
    ones = [[1] * 32 for _ in range(32)]
    want = []
    for b in range(0, 1024, 32):
        want += [sum(xs[b:b + 32]) % P256] * 32
    assert _blk32(ones, xs) == want
    thr = 0x7F7F7F7F7F7F7F7F7F7F7F7F7F7F7F7F7F7F7F7F7F7F7F7F7F7F7F7F7F7F7F7F
    row = [thr, thr + 1, thr - 1, P256 - thr, 2**255, 2**255 - 1, P256 - 2**255, 255, 256, 2**248, 2**248 - 1, P256 - 256, C, P256 - C, 3, P256 - 3] * 2
    for Tm in ([[row[(o + i) % 32] for i in range(32)] for o in range(32)], [[rnd.randrange(P256) for _ in range(32)] for _ in range(32)]):
        xs2 = [rnd.randrange(P256) for _ in range(2048)]
        inv00 = pow(Tm[0][0], -1, P256)
        targets = [0, 1, P256 - 1, P256 - rnd.randrange(1, 2**40), rnd.randrange(2**40), C, C - 1, P256 - C]
        for k, tgt in enumerate(targets):
            b = 32 * k
            rest = sum(Tm[0][i] * xs2[b + i] for i in range(1, 32)) % P256
            xs2[b] = (tgt - rest) * inv00 % P256
        got = _blk32(Tm, xs2)
        want = []
        for b in range(0, 2048, 32):
            want += [sum(Tm[o][i] * xs2[b + i] for i in range(32)) % P256 for o in range(32)]
        assert got == want
        for k, tgt in enumerate(targets):
            assert got[32 * k] == tgt
