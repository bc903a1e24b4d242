"""CPU tests that PIN the oracle (oracle/*.c): against the golden vectors produced by the independent
Python big-integer implementation (tests/golden/gen_golden.py), the identities asserted by the
reference's own tests, and the few literal known answers that exist."""
import os

import numpy as np
import pytest

from conftest import load_golden, std_to_field

FIELDS = ["secp256k1", "m31"]
SIZES = [4, 64, 4096]


@pytest.mark.parametrize("field", FIELDS)
@pytest.mark.parametrize("n", SIZES)
def test_leaves_match_curve_constants(oracle_tree, field, n):
    F, t = oracle_tree(field, n)
    g = load_golden(field, n)
    assert np.array_equal(t.leaves(), std_to_field(F, g["leaves"]))


@pytest.mark.parametrize("field", FIELDS)
def test_leaves_at_equals_the_golden_leaves_and_the_subtree_rule(oracle_tree, oracle_mod, field):
    """F.leaves_at (individual leaves by double-and-add, used by the GPU spot checks at sizes whose oracle tree would take
    minutes) against the golden leaves, and the subtree relation leaves_n[i] = leaves_2n[2i] (src/fftree.rs:471-478)"""
    F = oracle_mod.field(field)
    g = load_golden(field, 4096)
    idx = np.array([0, 1, 2, 3, 77, 2047, 2048, 4094, 4095])
    assert np.array_equal(F.leaves_at(4096, idx), std_to_field(F, g["leaves"])[idx])
    assert np.array_equal(F.leaves_at(1 << 22, 2 * idx), F.leaves_at(1 << 21, idx))


@pytest.mark.parametrize("field", FIELDS)
def test_extend_only_tree_extends_like_the_full_tree(oracle_tree, oracle_mod, field):
    """oracle.build_extend_tree (f layers + matrices of ONE tree: what extend_impl reads, src/fftree.rs:72-126, 341-363 — used for
    BASELINE configs[3], whose whole 2^23 chain would take a quarter of an hour) gives the EXTEND of the full tree, both directions"""
    F, full = oracle_tree(field, 4096)
    xt = F.build_extend_tree(2048)
    g = load_golden(field, 4096)
    s0, s1 = std_to_field(F, g["extend_s0"]), std_to_field(F, g["extend_s1"])
    assert np.array_equal(xt.extend(s0, oracle_mod.S1), s1) and np.array_equal(xt.extend(s1, oracle_mod.S0), s0)
    x = std_to_field(F, g["enter_evals"])[:2048]
    assert np.array_equal(xt.extend(x, oracle_mod.S1), full.extend(x, oracle_mod.S1))


@pytest.mark.parametrize("n", SIZES)
def test_secp_rational_maps(oracle_tree, n):
    F, t = oracle_tree("secp256k1", n)
    g = load_golden("secp256k1", n)
    num, den = std_to_field(F, g["map_num"]), std_to_field(F, g["map_den"])
    for k in range(n.bit_length() - 1):
        onum, oden = t.rational_map(k)
        assert np.array_equal(onum, num[3 * k:3 * k + 3]) and np.array_equal(oden, den[3 * k:3 * k + 3])


def test_m31_first_map_and_two_to_one_layers(oracle_tree, oracle_mod):
    F, t = oracle_tree("m31", 64)
    g = load_golden("m31", 64)
    num, den = t.rational_map(0)
    assert np.array_equal(num, g["map0_num"]) and np.array_equal(den, g["map0_den"])
    # every layer is 2-to-1: L_{k+1}[j] = psi_k(L_k[j]) = psi_k(L_k[j + |L_{k+1}|])  (src/fftree.rs:63-66)
    f = t.table(oracle_mod.T_F)
    P = 2**31 - 1
    for k in range(6):
        num, den = [int(x) for x in t.rational_map(k)[0]], [int(x) for x in t.rational_map(k)[1]]
        prev = [int(x) for x in f[64 >> k:2 * (64 >> k)]]
        nxt = [int(x) for x in f[64 >> (k + 1):2 * (64 >> (k + 1))]]
        ev = lambda c, x: (c[0] + c[1] * x + c[2] * x * x) % P
        for j, y in enumerate(nxt):
            for x in (prev[j], prev[j + len(nxt)]):
                assert ev(num, x) * pow(ev(den, x), -1, P) % P == y


@pytest.mark.parametrize("field", FIELDS)
@pytest.mark.parametrize("n", SIZES)
def test_enter_equals_naive_evaluation(oracle_tree, field, n):
    """reference tests `evaluates_polynomial` (src/lib.rs:108-120, 239-251)"""
    F, t = oracle_tree(field, n)
    g = load_golden(field, n)
    assert np.array_equal(t.enter(std_to_field(F, g["enter_coeffs"])), std_to_field(F, g["enter_evals"]))


@pytest.mark.parametrize("field", FIELDS)
@pytest.mark.parametrize("n", SIZES)
def test_extend_both_directions(oracle_tree, oracle_mod, field, n):
    """reference tests `extends_evaluations_from_s0_to_s1` / `_s1_to_s0` (src/lib.rs:122-152)"""
    F, t = oracle_tree(field, n)
    g = load_golden(field, n)
    s0, s1 = std_to_field(F, g["extend_s0"]), std_to_field(F, g["extend_s1"])
    assert np.array_equal(t.extend(s0, oracle_mod.S1), s1)
    assert np.array_equal(t.extend(s1, oracle_mod.S0), s0)


@pytest.mark.parametrize("field", FIELDS)
@pytest.mark.parametrize("n", SIZES)
def test_exit_inverts_enter(oracle_tree, field, n):
    """examples/interp_eval.rs:33, src/lib.rs:253-264"""
    F, t = oracle_tree(field, n)
    g = load_golden(field, n)
    assert np.array_equal(t.exit(std_to_field(F, g["enter_evals"])), std_to_field(F, g["enter_coeffs"]))


@pytest.mark.parametrize("field", FIELDS)
def test_tables_match_product_definitions(oracle_tree, oracle_mod, field):
    F, t = oracle_tree(field, 64)
    g = load_golden(field, 64)
    o = oracle_mod
    for key, which in [("xnn_s", o.T_XNN_S), ("z0_s1", o.T_Z0_S1), ("z1_s0", o.T_Z1_S0),
                       ("z0z0_rem_xnn_s", o.T_Z0Z0), ("z1z1_rem_xnn_s", o.T_Z1Z1)]:
        assert np.array_equal(t.table(which), std_to_field(F, g["tbl_" + key])), key
        if field == "secp256k1":
            assert np.array_equal(t.table(which, 16), std_to_field(F, g["tbl16_" + key])), key
    # inverse tables really are inverses
    one = F.from_ints([1] * 32)
    assert np.array_equal(F.mul(t.table(o.T_Z0_S1), t.table(o.T_Z0_INV_S1)), one)
    assert np.array_equal(F.mul(t.table(o.T_Z1_S0), t.table(o.T_Z1_INV_S0)), one)
    assert np.array_equal(F.mul(t.table(o.T_XNN_S), t.table(o.T_XNN_S_INV)), F.from_ints([1] * 64))


def test_secp_n4_known_answer(oracle_tree):
    """SURVEY.md section 8(c) tiny KAT (standard-form hex)."""
    F, t = oracle_tree("secp256k1", 4)
    leaves = [0xe9850041b13ea03fadc4bee2afd2959604bf64c290bf3fc15165f15163fd5431,
              0x7be2fbf3ae9d273ba6a49ae6971b19c400293594f01c786c23541ca05aea1feb,
              0x5572aa81222943ef459b1357e86d495a48b05f6792944c2934caee772e365b51,
              0xf46f032ad793329690fbda0834d955c01dc04bd575a28770c905354772132ec5]
    evals = [0x38458de4970df6dc40d99fdf5fb522517dab07b01cf8e66ae9d7648425654445,
             0x1513fee40dcbeb1496db83f21cbeb9c162144d9d4c02b73185afdad428e4758c,
             0x5ee430b0c6edd5d6d822066373b7e63bfda0dfbcf07a1e56b3bbdb4773c2ed2d,
             0x690ef0d1cbf147b059a60ec6d38eaad3c2ac55201ec98d488779b7b3fc200061]
    assert F.to_ints(t.leaves()) == leaves
    assert F.to_ints(t.enter(F.from_ints([1, 2, 3, 4]))) == evals


def test_secp_montgomery_encoding():
    """in-memory form = x * 2^256 mod p as four little-endian u64 (ark-ff MontBackend<_, 4>)."""
    from oracle import oracle
    F = oracle.field("secp256k1")
    p = 2**256 - 2**32 - 977
    xs = [0, 1, 2, p - 1, 0x1000003D1, 12345678901234567890123456789]
    a = F.from_ints(xs)
    for i, x in enumerate(xs):
        assert sum(int(a[i, l]) << (64 * l) for l in range(4)) == x * 2**256 % p
    assert F.to_ints(a) == xs
    # field ops against python ints
    ys = [7, p - 5, 3, 2, p - 1, 987654321]
    b = F.from_ints(ys)
    assert F.to_ints(F.mul(a, b)) == [x * y % p for x, y in zip(xs, ys)]
    assert F.to_ints(F.add(a, b)) == [(x + y) % p for x, y in zip(xs, ys)]
    assert F.to_ints(F.sub(a, b)) == [(x - y) % p for x, y in zip(xs, ys)]
    assert F.to_ints(F.inv(b)) == [pow(y, -1, p) for y in ys]


def test_m31_cubic_roots_known_answer(oracle_mod):
    """the only literal KAT in the reference: src/utils.rs:401-413"""
    import ctypes
    F = oracle_mod.field("m31")
    roots = (ctypes.c_uint32 * 3)()
    n = F.lib.ora_m31_find_roots_cubic(0, 2**31 - 1 - 4, 0, roots)
    assert n == 3 and list(roots) == [0, 2, 2147483645]


def test_m31_reference_unit_tests(oracle_tree):
    """`interpolates_evaluations` (src/lib.rs:253-264) and `determines_degree` (src/lib.rs:266-278)"""
    F, t = oracle_tree("m31", 64)
    coeffs = np.array([1, 1, 5, 0, 0, 1, 0, 0], dtype=np.uint32)
    assert np.array_equal(t.exit(t.enter(coeffs)), coeffs)
    coeffs = np.array([1, 1, 1, 0, 0, 1, 0, 0], dtype=np.uint32)
    assert t.degree(t.enter(coeffs)) == 5


@pytest.mark.parametrize("field", FIELDS)
def test_subtree_dispatch_and_edge_cases(oracle_tree, oracle_mod, field):
    """subtree_with_size (src/fftree.rs:489-496): a length-k call on a bigger tree uses T_k whose
    leaves are every (N/k)-th leaf; size-1 transforms are identities; too-large panics."""
    F, t = oracle_tree(field, 64)
    leaves = t.leaves()
    rng = np.random.default_rng(7)
    for k in (1, 2, 8, 32):
        c = F.from_ints([int(x) for x in rng.integers(0, 2**31 - 1, k)])
        ev = t.enter(c)
        assert np.array_equal(ev, F.horner(c, leaves[::64 // k]))
        assert np.array_equal(t.exit(ev), c)
    with pytest.raises(ValueError):
        t.enter(F.from_ints(list(range(128))))
    with pytest.raises(ValueError):
        t.enter(F.from_ints(list(range(3))))
    with pytest.raises(ValueError):
        t.extend(F.from_ints(list(range(64))), oracle_mod.S1)  # needs T_128


def test_build_fftree_size_limits(oracle_mod):
    """src/lib.rs:62-64 (secp256k1: log n < 36) and src/ec.rs:513-515 (M31: log n <= 28)"""
    assert oracle_mod.field("secp256k1").build_fftree(1 << 36) is None
    assert oracle_mod.field("m31").build_fftree(1 << 29) is None
    with pytest.raises(AssertionError):
        oracle_mod.field("m31").build_fftree(48)


@pytest.mark.parametrize("field", FIELDS)
def test_redc_mod_vanish_definitions(oracle_tree, oracle_mod, field):
    """MOD (src/fftree.rs:283-289): modular_reduce(<P|S>, xnn_s, z0z0) = <P mod X^(n/2) | S>;
    VANISH (src/fftree.rs:310-316): evaluations of prod (x - a_i) on the leaves."""
    F, t = oracle_tree(field, 64)
    o = oracle_mod
    rng = np.random.default_rng(11)
    c = F.from_ints([int(x) for x in rng.integers(0, 2**31 - 1, 64)])
    ev = t.enter(c)
    low = c.copy(); low[32:] = 0
    got = t.modular_reduce(ev, t.table(o.T_XNN_S), t.table(o.T_Z0Z0))
    assert np.array_equal(got, t.enter(low))
    dom = F.from_ints([int(x) for x in rng.integers(1, 2**31 - 1, 32)])
    van = t.vanish(dom)
    leaves = t.leaves()
    acc = F.from_ints([1] * 64)
    for i in range(32):
        acc = F.mul(acc, F.sub(leaves, np.repeat(dom[i:i + 1], 64, axis=0)))
    assert np.array_equal(van, acc)


_SAN_CHILD = r'''
import os, sys
import numpy as np
sys.path.insert(0, os.environ["ECFFT_ROOT"]); sys.path.insert(0, os.path.join(os.environ["ECFFT_ROOT"], "tests"))
from oracle import oracle
from conftest import load_golden, std_to_field
for name in ("secp256k1", "m31"):
    F = oracle.field(name)
    assert "asan" in F.lib._name, F.lib._name
    for n in (4, 64, 4096):
        t = F.build_fftree(n)
        g = load_golden(name, n)
        c, ev = std_to_field(F, g["enter_coeffs"]), std_to_field(F, g["enter_evals"])
        assert np.array_equal(t.enter(c), ev) and np.array_equal(t.exit(ev), c)
        s0, s1 = std_to_field(F, g["extend_s0"]), std_to_field(F, g["extend_s1"])
        assert np.array_equal(t.extend(s0, oracle.S1), s1) and np.array_equal(t.extend(s1, oracle.S0), s0)
    t = F.build_fftree(4096, check_chain=True)
    rng = np.random.default_rng(3)
    mk = (lambda k: rng.integers(1, 2**31 - 1, k, dtype=np.uint32)) if F.limbs == 1 else (lambda k: F.from_ints([int(x) for x in rng.integers(1, 2**62, k)]))
    for n in (1, 2, 8, 512, 4096):                                   # every algorithm, every size class incl. the degenerate ones
        x = mk(n)
        assert np.array_equal(t.exit(t.enter(x)), x)
        if n >= 2:
            t.redc(x, mk(n), oracle.S0); t.redc(x, mk(n), oracle.S1); t.modular_reduce(x, mk(n), mk(n))
            assert 0 <= t.degree(t.enter(x)) < n
        if 2 * n <= 4096:
            t.mextend(x, oracle.S0); t.mextend(x, oracle.S1); t.vanish(x)
            assert np.array_equal(t.extend(t.extend(x, oracle.S1), oracle.S0), x)
    et = F.build_extend_tree(512)
    assert np.array_equal(et.extend(mk(512), oracle.S1).shape, mk(512).shape)
    F.horner(mk(100), F.leaves_at(4096, np.arange(0, 4096, 97)))
    F.inv(mk(33)); F.to_ints(mk(5))
    del t, et
print("ORACLE_SANITIZED_OK")
'''


def test_oracle_under_sanitizers(oracle_mod):
    """the oracle is the checker of every parity claim, so its own memory safety is checked — the same C sources built
    with -fsanitize=address,undefined (oracle/Makefile `asan`) and run in a child process (libasan preloaded) over the golden vectors and
    every algorithm at n <= 4096: no heap overflow, no use after free, no signed overflow, no misaligned access, no shift past the width."""
    import subprocess
    import sys
    from conftest import ROOT
    subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "asan"], check=True)
    libasan = subprocess.run(["gcc", "-print-file-name=libasan.so"], capture_output=True, text=True).stdout.strip()
    libubsan = subprocess.run(["gcc", "-print-file-name=libubsan.so"], capture_output=True, text=True).stdout.strip()
    env = dict(os.environ, LD_PRELOAD=f"{libasan}:{libubsan}", ASAN_OPTIONS="detect_leaks=0", UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=1",
               ECFFT_ORACLE_LIBDIR=os.path.join(ROOT, "oracle", "asan"), ECFFT_ROOT=ROOT)
    r = subprocess.run([sys.executable, "-c", _SAN_CHILD], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and "ORACLE_SANITIZED_OK" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]
    assert "AddressSanitizer" not in r.stderr and "runtime error" not in r.stderr, r.stderr[-4000:]
