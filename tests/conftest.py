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
