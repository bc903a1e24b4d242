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


# ------------------------------------------------------------------------------------ at scale (round 6)
class _OracleAlgorithms17:
    """VERDICT r05 missing #4: the f3 wrappers never met the oracle above 2^11 — the column passes, the 1024-element tiles, the matrix-core
    phases and the device batch inversion at scale were reached through ENTER / EXIT only.  The oracle's 2^17 trees (secp256k1: ~10 s
    to build) and the expected sides of MEXTEND (e = 2^16), REDC_z0 / z1 with an arbitrary `a`, MOD, VANISH (nd = 2^15) and DEGREE at
    n = 2^16, computed on a background host thread while the small tests above run (ctypes releases the GIL)."""
    LOG_TREE, LOG_N = 17, 16

    def __init__(self, oracle_mod, field):
        import threading
        self.o, self.F = oracle_mod, oracle_mod.field(field)
        self.res, self.err = {}, []
        self.th = threading.Thread(target=self._run, daemon=True)
        self.th.start()

    def inputs(self):
        F, n = self.F, 1 << self.LOG_N
        big = lambda k, seed, nz=False: _rand_big(F, k, seed, nz)      # noqa: E731
        d = 40_000
        cd = big(n, 5, True); cd[d + 1:] = 0
        return dict(x=big(n, 1), ev=big(n, 2), a=big(n, 3, True), c=big(n, 4), dom=big(n // 2, 6, True), coeffs_deg=cd, deg=d)

    def _run(self):
        try:
            o, i = self.o, self.inputs()
            ot = self.F.build_fftree(1 << self.LOG_TREE)
            self.res = dict(mext_s1=ot.mextend(i["x"], o.S1), mext_s0=ot.mextend(i["x"], o.S0),
                            redc_z0=ot.redc(i["ev"], i["a"], o.S0), redc_z1=ot.redc(i["ev"], i["a"], o.S1),
                            mod=ot.modular_reduce(i["ev"], i["a"], i["c"]), vanish=ot.vanish(i["dom"]),
                            ev_deg=ot.enter(i["coeffs_deg"]))
            self.res["deg"] = ot.degree(self.res["ev_deg"])
        except Exception as e:  # pragma: no cover
            self.err.append(e)

    def get(self):
        self.th.join()
        assert not self.err, self.err
        return self.res


def _rand_big(F, n, seed, nonzero=False):
    """large arrays without a Python loop: any 256-bit pattern below p is a valid element (top bit cleared); nonzero: low limb |= 1"""
    rng = np.random.default_rng(0x5EED1700 + seed)
    if F.limbs == 1:
        return rng.integers(1 if nonzero else 0, 2**31 - 1, n, dtype=np.uint32)
    a = rng.integers(0, 2**64, size=(n, 4), dtype=np.uint64); a[:, 3] >>= np.uint64(1)
    if nonzero:
        a[:, 0] |= np.uint64(1)
    return a


_alg17 = {}


@pytest.fixture(scope="module", autouse=True)
def alg17(oracle_mod):
    """started with the first test of this module"""
    if not _alg17:
        for f in FIELDS:
            _alg17[f] = _OracleAlgorithms17(oracle_mod, f)
    return _alg17


@pytest.mark.parametrize("field", FIELDS)
def test_algorithms_at_2e16_vs_oracle(alg17, oracle_mod, field):
    """MEXTEND, REDC_z0 / z1 (arbitrary a: the device batch inversion of 2^15 elements), MOD, VANISH (nd = 2^15) and DEGREE at n = 2^16 on
    a 2^17 tree against the oracle's restatement of src/fftree.rs:128-141, 169-198, 261-316, element for element"""
    import ecfft_amd as G
    job = alg17[field]
    gt = G.FIELDS[field].build_fftree(1 << job.LOG_TREE)
    i = job.inputs()
    want = job.get()
    assert np.array_equal(gt.mextend(i["x"], G.Moiety.S1), want["mext_s1"])
    assert np.array_equal(gt.mextend(i["x"], G.Moiety.S0), want["mext_s0"])
    assert np.array_equal(gt.redc_z0(i["ev"], i["a"]), want["redc_z0"])
    assert np.array_equal(gt.redc_z1(i["ev"], i["a"]), want["redc_z1"])
    assert np.array_equal(gt.modular_reduce(i["ev"], i["a"], i["c"]), want["mod"])
    assert np.array_equal(gt.vanish(i["dom"]), want["vanish"])
    assert want["deg"] == i["deg"]
    assert gt.degree(want["ev_deg"]) == i["deg"]
    assert np.array_equal(gt.enter(i["coeffs_deg"]), want["ev_deg"])


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
