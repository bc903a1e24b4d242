"""Consumer of the Rust pin kit (tools/rust_pin): when a dump produced by the REAL ecfft crate is dropped into
tests/golden/rust/pin.txt, every record is checked against the oracle, the wire-format writer and (with -m gpu) the HIP path.
That turns the two "parity unpinned" items of DESIGN.md section 6 — ark-ff's in-memory limb encoding and the ark-serialize
layout of FFTree<F> (src/fftree.rs:507-660) — into a two-minute job on any machine with cargo.  Without the file the
crate-backed tests skip; the consumer itself is exercised on a dump of the same format written by the oracle."""
import os

import numpy as np
import pytest

from conftest import GOLDEN

PIN = os.path.join(GOLDEN, "rust", "pin.txt")
FIELDS = ["secp256k1", "m31"]


def parse_pin(text):
    rec = {}
    for line in text.splitlines():
        line = line.strip()
        if not line or line.startswith("#") or " = " not in line:
            continue
        k, v = line.split(" = ", 1)
        rec[k.strip()] = v.strip()
    return rec


def elems(F, hexstr):
    """in-memory element bytes -> array in the oracle's / C ABI's representation"""
    raw = np.frombuffer(bytes.fromhex(hexstr), dtype=np.uint8)
    return raw.view(F.dtype).reshape(F.shape(raw.size // (F.dtype.itemsize * F.limbs))).copy()


def ints_1_to(F, n):
    return F.from_ints(list(range(1, n + 1)))


def check_records(rec, name, F, tree_of, serializer=None, P=None):
    """every record of field `name`; tree_of(n) -> object with enter/exit/extend/leaves/table"""
    p = 2**256 - 2**32 - 977 if F.limbs > 1 else 2**31 - 1
    seen = 0
    k = f"{name}.size_of"
    if k in rec:
        assert int(rec[k], 16) == F.dtype.itemsize * F.limbs; seen += 1
    k = f"{name}.mem.0_1_2_977_minus1"
    if k in rec:                                     # THE encoding check: crate bytes == our from_ints
        assert np.array_equal(elems(F, rec[k]), F.from_ints([0, 1, 2, 977, p - 1])); seen += 1
    for n in (4, 64):
        t = tree_of(n)
        pre = f"{name}.n{n}."
        c = ints_1_to(F, n)
        table_ids = {"xnn_s": 3, "z0z0_rem_xnn_s": 9}
        checks = {"leaves.mem": lambda: t.leaves(), "enter_1_to_n.mem": lambda: t.enter(c), "exit_1_to_n.mem": lambda: t.exit(c),
                  "extend_s1_1_to_half.mem": lambda: t.extend(c[: n // 2], 1), "extend_s0_1_to_half.mem": lambda: t.extend(c[: n // 2], 0),
                  "xnn_s.mem": lambda: t.table(table_ids["xnn_s"]) if not hasattr(t, "_oracle_ids") else t.table(t._oracle_ids["xnn_s"]),
                  "z0z0_rem_xnn_s.mem": lambda: t.table(table_ids["z0z0_rem_xnn_s"]) if not hasattr(t, "_oracle_ids") else t.table(t._oracle_ids["z0z0_rem_xnn_s"])}
        for suffix, fn in checks.items():
            if pre + suffix in rec:
                assert np.array_equal(np.asarray(fn()), elems(F, rec[pre + suffix])), pre + suffix
                seen += 1
        if serializer is not None:
            for mode, compress in (("serialize_compressed", True), ("serialize_uncompressed", False)):
                if pre + mode in rec:
                    want = bytes.fromhex(rec[pre + mode])
                    assert serializer.serialize_fftree(t, P, compress) == want, pre + mode
                    if hasattr(t, "serialize"):         # device tree: the C-ABI writer and reader (ecfft_fftree_serialize / _deserialize)
                        assert t.serialize(compress) == want, pre + mode + " (C ABI writer)"
                        back = serializer.deserialize_fftree(P, want, compress, verify=True)
                        assert back.n == n and back.serialize(compress) == want, pre + mode + " (C ABI reader)"
                    seen += 1
            if pre + "serialized_size_compressed" in rec:
                assert serializer.serialized_size(P, n, True) == int(rec[pre + "serialized_size_compressed"], 16); seen += 1
    return seen


def _oracle_tree_of(oracle_mod, F):
    cache = {}

    def get(n):
        if n not in cache:
            t = F.build_fftree(n)
            t._oracle_ids = {"xnn_s": oracle_mod.T_XNN_S, "z0z0_rem_xnn_s": oracle_mod.T_Z0Z0}
            cache[n] = t
        return cache[n]
    return get


def oracle_dump(oracle_mod):
    """a pin file in the kit's format, written by the ORACLE (used to test the consumer; not evidence about the crate)"""
    import ecfft_amd
    from ecfft_amd import serialize as S
    lines = []
    for name in FIELDS:
        F = oracle_mod.field(name)
        P = ecfft_amd.FIELDS[name]
        p = 2**256 - 2**32 - 977 if F.limbs > 1 else 2**31 - 1
        lines.append(f"{name}.size_of = {F.dtype.itemsize * F.limbs:02x}")
        lines.append(f"{name}.mem.0_1_2_977_minus1 = {F.from_ints([0, 1, 2, 977, p - 1]).tobytes().hex()}")
        for n in (4, 64):
            t = F.build_fftree(n)
            c = ints_1_to(F, n)
            lines.append(f"{name}.n{n}.leaves.mem = {np.ascontiguousarray(t.leaves()).tobytes().hex()}")
            lines.append(f"{name}.n{n}.enter_1_to_n.mem = {t.enter(c).tobytes().hex()}")
            lines.append(f"{name}.n{n}.exit_1_to_n.mem = {t.exit(c).tobytes().hex()}")
            lines.append(f"{name}.n{n}.extend_s1_1_to_half.mem = {t.extend(c[: n // 2], 1).tobytes().hex()}")
            lines.append(f"{name}.n{n}.extend_s0_1_to_half.mem = {t.extend(c[: n // 2], 0).tobytes().hex()}")
            lines.append(f"{name}.n{n}.xnn_s.mem = {np.ascontiguousarray(t.table(oracle_mod.T_XNN_S)).tobytes().hex()}")
            lines.append(f"{name}.n{n}.serialize_compressed = {S.serialize_fftree(t, P, True).hex()}")
            lines.append(f"{name}.n{n}.serialize_uncompressed = {S.serialize_fftree(t, P, False).hex()}")
            lines.append(f"{name}.n{n}.serialized_size_compressed = {S.serialized_size(P, n, True):x}")
    return "\n".join(lines) + "\n"


def test_consumer_accepts_a_dump_in_the_kit_format(oracle_mod):
    """the consumer parses the format and re-derives every record (self-check on an oracle-written dump)"""
    import ecfft_amd
    from ecfft_amd import serialize as S
    rec = parse_pin(oracle_dump(oracle_mod))
    for name in FIELDS:
        F = oracle_mod.field(name)
        seen = check_records(rec, name, F, _oracle_tree_of(oracle_mod, F), S, ecfft_amd.FIELDS[name])
        assert seen >= 20
    # and it really compares: a corrupted record is rejected
    bad = dict(rec); k = "m31.n4.enter_1_to_n.mem"; bad[k] = ("00" if bad[k][:2] != "00" else "01") + bad[k][2:]
    with pytest.raises(AssertionError):
        check_records(bad, "m31", oracle_mod.field("m31"), _oracle_tree_of(oracle_mod, oracle_mod.field("m31")))


@pytest.mark.skipif(not os.path.exists(PIN), reason="no Rust-produced pin file (tools/rust_pin) in tests/golden/rust/")
@pytest.mark.parametrize("name", FIELDS)
def test_oracle_and_wire_format_against_the_crate(oracle_mod, name):
    import ecfft_amd
    from ecfft_amd import serialize as S
    rec = parse_pin(open(PIN).read())
    F = oracle_mod.field(name)
    assert check_records(rec, name, F, _oracle_tree_of(oracle_mod, F), S, ecfft_amd.FIELDS[name]) >= 10


@pytest.mark.gpu
@pytest.mark.skipif(not os.path.exists(PIN), reason="no Rust-produced pin file (tools/rust_pin) in tests/golden/rust/")
@pytest.mark.parametrize("name", FIELDS)
def test_hip_path_against_the_crate(oracle_mod, name):
    import ecfft_amd
    from ecfft_amd import serialize as S
    rec = parse_pin(open(PIN).read())
    F = oracle_mod.field(name)
    P = ecfft_amd.FIELDS[name]
    cache = {}

    def tree_of(n):
        if n not in cache:
            cache[n] = P.build_fftree(n)
        return cache[n]
    assert check_records(rec, name, F, tree_of, S, P) >= 10


@pytest.mark.gpu
@pytest.mark.parametrize("name", FIELDS)
def test_consumer_drives_the_c_abi_wire_format_on_device_trees(oracle_mod, name):
    """the same consumer on DEVICE trees against an oracle-written dump: every record incl. the two serialisations goes through
    the HIP path and the C-ABI writer / reader (what test_hip_path_against_the_crate will do with a crate-written dump)"""
    import ecfft_amd
    from ecfft_amd import serialize as S
    rec = parse_pin(oracle_dump(oracle_mod))
    P = ecfft_amd.FIELDS[name]
    cache = {}

    def tree_of(n):
        if n not in cache:
            cache[n] = P.build_fftree(n)
        return cache[n]
    assert check_records(rec, name, oracle_mod.field(name), tree_of, S, P) >= 20
