"""GPU tests of the remaining FFTree algorithms: MEXTEND, REDC_z0/z1, MOD, VANISH, DEGREE
through the C-ABI against the oracle's restatement of src/fftree.rs:128-141, 169-198, 261-316, bit for bit."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

FIELDS = ["secp256k1", "m31"]


def rand_elems(F, n, seed, nonzero=False):
    rng = np.random.default_rng(seed)
    lo = 1 if nonzero else 0
    if F.limbs == 1:
        return rng.integers(lo, 2**31 - 1, n, dtype=np.uint32)
    p = 2**256 - 2**32 - 977
    return F.from_ints([lo + int.from_bytes(rng.bytes(32), "little") % (p - lo) for _ in range(n)])


@pytest.fixture(scope="module")
def trees(oracle_mod):
    import ecfft_amd
    out = {}
    for f in FIELDS:
        F = oracle_mod.field(f)
        out[f] = (F, F.build_fftree(1 << 11), ecfft_amd.FIELDS[f].build_fftree(1 << 11), ecfft_amd)
    return out


@pytest.mark.parametrize("field", FIELDS)
@pytest.mark.parametrize("n", [2, 8, 256, 2048])
def test_mextend_redc_mod(trees, oracle_mod, field, n):
    F, ot, gt, G = trees[field]
    o = oracle_mod
    if 2 * n <= gt.n:
        x = rand_elems(F, n, n)
        assert np.array_equal(gt.mextend(x, G.Moiety.S1), ot.mextend(x, o.S1))
        assert np.array_equal(gt.mextend(x, G.Moiety.S0), ot.mextend(x, o.S0))
    ev = rand_elems(F, n, 100 + n)
    a = rand_elems(F, n, 200 + n, nonzero=True)          # arbitrary `a` with no zero on S0
    assert np.array_equal(gt.redc_z0(ev, a), ot.redc(ev, a, o.S0))
    assert np.array_equal(gt.redc_z1(ev, a), ot.redc(ev, a, o.S1))
    c = rand_elems(F, n, 300 + n)
    assert np.array_equal(gt.modular_reduce(ev, a, c), ot.modular_reduce(ev, a, c))
    # the way EXIT uses it: P mod X^(n/2) with the tree's own tables (src/fftree.rs:206-207)
    xnn, zz = ot.table(o.T_XNN_S, n), ot.table(o.T_Z0Z0, n)
    coeffs = rand_elems(F, n, 400 + n)
    low = coeffs.copy(); low[n // 2:] = 0
    assert np.array_equal(gt.modular_reduce(ot.enter(coeffs), xnn, zz), ot.enter(low))


@pytest.mark.parametrize("field", FIELDS)
@pytest.mark.parametrize("nd", [1, 4, 128, 1024])
def test_vanish(trees, field, nd):
    F, ot, gt, G = trees[field]
    dom = rand_elems(F, nd, 7 + nd, nonzero=True)
    assert np.array_equal(gt.vanish(dom), ot.vanish(dom))


@pytest.mark.parametrize("field", FIELDS)
def test_degree(trees, field):
    F, ot, gt, G = trees[field]
    for n, d in [(8, 5), (64, 0), (64, 31), (64, 32), (64, 63), (1024, 700), (2048, 2047), (2048, 1)]:
        c = rand_elems(F, n, n + d, nonzero=True)
        c[d + 1:] = 0
        ev = ot.enter(c)
        assert ot.degree(ev) == d
        assert gt.degree(ev) == d
    # the reference's own test (src/lib.rs:266-278)
    if field == "m31":
        ev = ot.enter(np.array([1, 1, 1, 0, 0, 1, 0, 0], dtype=np.uint32))
        assert gt.degree(ev) == 5


@pytest.mark.parametrize("field", FIELDS)
def test_algorithm_errors(trees, field):
    F, ot, gt, G = trees[field]
    with pytest.raises(ValueError, match="FFTree is too small"):
        gt.mextend(rand_elems(F, 2048, 1), G.Moiety.S1)        # needs T_4096
    with pytest.raises(ValueError, match="FFTree is too small"):
        gt.vanish(rand_elems(F, 2048, 1))
    with pytest.raises(AssertionError):
        gt.degree(rand_elems(F, 24, 1))


@pytest.mark.parametrize("field", FIELDS)
def test_table_fma_building_block(trees, oracle_mod, field):
    """ecfft_table_fma — the pointwise step of the multi-GPU ENTER / EXIT — against oracle tables and field ops"""
    F, ot, gt, G = trees[field]
    o = oracle_mod
    m, cnt, off, stride = 256, 50, 7, 2
    x, y = rand_elems(F, cnt, 1), rand_elems(F, cnt, 2)
    for which in (o.T_XNN_S, o.T_XNN_S_INV, o.T_Z0Z0):
        T = ot.table(which, m)[off + np.arange(cnt) * stride]
        assert np.array_equal(gt.table_fma(x, None, m, which, off, stride, 0), F.mul(x, T))
        assert np.array_equal(gt.table_fma(x, y, m, which, off, stride, 1), F.add(F.mul(x, T), y))
        assert np.array_equal(gt.table_fma(x, y, m, which, off, stride, 2), F.sub(y, F.mul(x, T)))
        assert np.array_equal(gt.table_fma(x, y, m, which, off, stride, 3), F.mul(F.sub(y, x), T))
    T = ot.table(o.T_Z0_INV_S1, m)[3:3 + cnt]
    assert np.array_equal(gt.table_fma(x, None, m, o.T_Z0_INV_S1, 3, 1, 0), F.mul(x, T))
    with pytest.raises(Exception):
        gt.table_fma(x, None, m, o.T_XNN_S, 250, 2, 0)          # index range beyond the table


@pytest.mark.gpu
@pytest.mark.parametrize("field,n", [("secp256k1", 1 << 12), ("m31", 1 << 16)])
def test_pointwise_vanishing_tables_match_the_reference_construction(field, n, hooks_lib):
    """z0_s1 / z1_s0 of every tree of the chain: the EXTEND-based construction of the reference (src/fftree.rs:386-397, already
    compared with the oracle in test_tables_match_oracle) == the pointwise isogeny-chain formula used by the sharded builds"""
    import ecfft_amd
    from ecfft_amd import fftree as FT
    tree = ecfft_amd.FIELDS[field].build_fftree(n)
    m = 2
    while m <= n:
        assert FT.lib().ecfft_selfcheck_pointwise_z(tree._h, m) == 0, (field, m)
        m *= 2


@pytest.mark.parametrize("field", FIELDS)
def test_temporary_pool_is_counted_bounded_and_trimmable(field):
    """the algorithm wrappers' pooled temporaries are part of ecfft_ctx_device_bytes, a tiny request does not take a
    huge block, and ecfft_ctx_trim gives the pool back"""
    import ecfft_amd
    F = ecfft_amd.FIELDS[field]
    t = F.build_fftree(1 << 14)
    base = t.device_bytes
    rng = np.random.default_rng(1)
    ev = t.enter(rng.integers(1, 1000, size=(1 << 14,) + ((4,) if field == "secp256k1" else ()), dtype=F.dtype))
    dom = rng.integers(1, 1000, size=(1 << 13,) + ((4,) if field == "secp256k1" else ()), dtype=F.dtype)
    r1 = t.vanish(dom); d1 = t.degree(ev)
    grown = t.device_bytes
    assert grown > base                                   # the pool is visible in the footprint
    r2 = t.vanish(dom); d2 = t.degree(ev)
    assert t.device_bytes == grown                        # steady state: the same blocks are reused, nothing new is allocated
    assert np.array_equal(r1, r2) and d1 == d2
    t.trim()
    assert t.device_bytes == base
    assert np.array_equal(t.vanish(dom), r1) and t.degree(ev) == d1       # and the wrappers still work after a trim
