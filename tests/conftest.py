import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_golden(field, n):
    return dict(np.load(os.path.join(GOLDEN, f"{field}_n{n}.npz")))


@pytest.fixture(scope="session")
def oracle_mod():
    from oracle import oracle
    oracle.build()
    return oracle


@pytest.fixture
def hooks_lib():
    """the HOOKS build of the library (tests/hooks/libecfft_hip_hooks.so = the same sources with -DECFFT_TEST_HOOKS: the entry points
    of include/ecfft_hip_hooks.h and the A/B switches read from the environment) for the duration of one test.  Everything the test
    makes (contexts, communicators) must be made inside it — never a tree from the module-level caches."""
    from ecfft_amd import fftree as FT
    with FT.use_hooks_library() as L:
        yield L


_tree_cache = {}


@pytest.fixture(scope="session")
def oracle_tree(oracle_mod):
    """oracle_tree(field, n) -> (Field, OracleFFTree), cached per session."""
    def get(field, n):
        key = (field, n)
        if key not in _tree_cache:
            F = oracle_mod.field(field)
            _tree_cache[key] = (F, F.build_fftree(n))
        return _tree_cache[key]
    return get


def std_to_field(F, std):
    """golden arrays are in standard form; convert to the in-memory (Montgomery for secp) form."""
    import ctypes
    std = np.ascontiguousarray(std)
    out = np.zeros_like(std)
    F._from_std(std.ctypes.data_as(ctypes.c_void_p), out.ctypes.data_as(ctypes.c_void_p), std.shape[0])
    return out


def horner_mt(F, coeffs, xs, threads=None):
    """F.horner (the oracle's naive evaluation, the expected side of the reference's tests, src/lib.rs:108-120) on many
    points, the points split over host threads (the oracle is called through ctypes, which releases the GIL)."""
    from concurrent.futures import ThreadPoolExecutor
    threads = threads or max(1, min(16, (os.cpu_count() or 1)))
    parts = [p for p in np.array_split(np.arange(len(xs)), threads) if len(p)]
    with ThreadPoolExecutor(len(parts)) as ex:
        outs = list(ex.map(lambda p: F.horner(coeffs, xs[p]), parts))
    return np.concatenate(outs, axis=0)


def spread_indices(n, count=1024, seed=0):
    """`count` seeded positions in [0, n) that hit every residue mod `count` once (every position inside a `count`-element
    tile) and both halves of the array (the two half-streams of a single large transform) about equally, plus the corners."""
    rng = np.random.default_rng(seed)
    idx = (np.arange(count) + count * rng.integers(0, max(1, n // count), count)) % n
    return np.union1d(idx, np.array([0, 1, 2, n // 2 - 1, n // 2, n // 2 + 1, n - 2, n - 1]) % n)
