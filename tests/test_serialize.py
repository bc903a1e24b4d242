"""FFTree wire format (ecfft_amd/serialize.py, reference src/fftree.rs:507-660).  Self-consistency only: no Rust-produced
file is available in this image (the module's header says so: parity unpinned)."""
import numpy as np
import pytest

FIELDS = ["secp256k1", "m31"]


@pytest.mark.parametrize("field", FIELDS)
@pytest.mark.parametrize("compress", [True, False])
def test_write_parse_roundtrip_cpu(oracle_tree, field, compress):
    """writer driven by the oracle's tree (no GPU): byte count == FFTree::serialized_size, parser returns every table"""
    import ecfft_amd
    from ecfft_amd import serialize as S
    P = ecfft_amd.FIELDS[field]
    F, ot = oracle_tree(field, 64)
    data = S.serialize_fftree(ot, P, compress)
    assert len(data) == S.serialized_size(P, 64, compress)
    levels = S.parse_fftree(P, data, compress)
    assert [lv.n for lv in levels] == [64, 32, 16, 8, 4, 2, 1]
    for lv in levels:
        assert len(lv.maps) == lv.n.bit_length() - 1
        for which, arr in lv.tables.items():
            assert np.array_equal(arr, ot.table(which, lv.n)), (lv.n, which)
    with pytest.raises(ValueError):
        S.parse_fftree(P, data + b"\x00", compress)
    # a field element is its standard-form integer, little endian: first leaf of the file == x(coset offset)
    leaf0 = int.from_bytes(data[8 + 64 * P.elem_bytes: 8 + 65 * P.elem_bytes], "little")
    assert leaf0 == F.to_ints(ot.leaves()[:1])[0]


@pytest.mark.gpu
@pytest.mark.parametrize("field", FIELDS)
@pytest.mark.parametrize("compress", [True, False])
def test_deserialized_tree_works(oracle_tree, field, compress):
    """reference tests deserialized_{un,}compressed_tree_works (src/lib.rs:154-186): a tree rebuilt from the bytes
    evaluates polynomials; here additionally every table of the file must equal the GPU-rebuilt one, and the GPU tree
    re-serialises to the same bytes"""
    import ecfft_amd
    from ecfft_amd import serialize as S
    P = ecfft_amd.FIELDS[field]
    F, ot = oracle_tree(field, 64)
    data = S.serialize_fftree(ot, P, compress)
    t = S.deserialize_fftree(P, data, compress, verify=True)
    rng = np.random.default_rng(3)
    c = F.from_ints([int(x) for x in rng.integers(0, 2**31 - 1, 64)])
    assert np.array_equal(t.enter(c), ot.enter(c))
    assert S.serialize_fftree(t, P, compress) == data
    built = P.build_fftree(64)
    assert S.serialize_fftree(built, P, compress) == data
