"""TEST INFRASTRUCTURE — ctypes loader for the CPU oracle (oracle/*.c).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module;
the product (ecfft_amd/) never does.  Mirrors the reference surface
`FftreeField::build_fftree` / `FFTree::{extend, enter, exit, ...}` (/root/reference/src/lib.rs:14-16,
src/fftree.rs:123,164,227) on numpy arrays in the crate's in-memory element representation:
secp256k1 -> uint64[n,4] Montgomery limbs, m31 -> uint32[n].
"""
import ctypes
import os
import subprocess

import numpy as np

_DIR = os.path.dirname(os.path.abspath(__file__))

T_F, T_RECOMBINE, T_DECOMPOSE, T_XNN_S, T_XNN_S_INV, T_Z0_S1, T_Z1_S0, T_Z0_INV_S1, T_Z1_INV_S0, T_Z0Z0, T_Z1Z1 = range(11)
S0, S1 = 0, 1


def build():
    """Compile the oracle libraries (gcc) if they are missing or stale."""
    subprocess.run(["make", "-s", "-C", _DIR], check=True)


class _Field:
    def __init__(self, name, prefix, dtype, limbs):
        self.name, self.prefix, self.dtype, self.limbs = name, prefix, np.dtype(dtype), limbs
        # ECFFT_ORACLE_LIBDIR: another build of the same sources (oracle/asan: `make asan`, the AddressSanitizer + UBSan build that
        # tests/test_oracle.py runs in a child process with libasan preloaded)
        libdir = os.environ.get("ECFFT_ORACLE_LIBDIR") or _DIR
        path = os.path.join(libdir, f"libecfft_oracle_{name}.so")
        if not os.path.exists(path):
            build()
        self.lib = ctypes.CDLL(path)
        self._sig("build_fftree", ctypes.c_void_p, [ctypes.c_uint, ctypes.c_int])
        self._sig("build_extend_tree", ctypes.c_void_p, [ctypes.c_uint])
        self._sig("free_tree", None, [ctypes.c_void_p])
        self._sig("tree_size", ctypes.c_size_t, [ctypes.c_void_p])
        self._sig("table", ctypes.c_void_p, [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int, ctypes.POINTER(ctypes.c_size_t)])
        self._sig("ratmap", ctypes.c_int, [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p])
        self._sig("tree_new", ctypes.c_void_p, [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p, ctypes.c_void_p])
        for f in ("extend", "mextend"):
            self._sig(f, ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int])
        for f in ("enter", "exit", "vanish"):
            self._sig(f, ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t])
        self._sig("redc", ctypes.c_int, [ctypes.c_void_p] * 4 + [ctypes.c_size_t, ctypes.c_int])
        self._sig("modular_reduce", ctypes.c_int, [ctypes.c_void_p] * 5 + [ctypes.c_size_t])
        self._sig("degree", ctypes.c_long, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t])
        self._sig("horner", None, [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p])
        for f in ("mul_vec", "add_vec", "sub_vec"):
            self._sig(f, None, [ctypes.c_void_p] * 3 + [ctypes.c_size_t])
        self._sig("inv_vec", None, [ctypes.c_void_p] * 2 + [ctypes.c_size_t])
        for f in ("from_std", "to_std"):
            self._sig(f, None, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t])
        self._sig("leaves_at", ctypes.c_int, [ctypes.c_uint, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p])

    def _sig(self, name, res, args):
        fn = getattr(self.lib, f"{self.prefix}{name}")
        fn.restype, fn.argtypes = res, args
        setattr(self, "_" + name, fn)

    # ---- element helpers -------------------------------------------------------------
    def shape(self, n):
        return (n, self.limbs) if self.limbs > 1 else (n,)

    def empty(self, n):
        return np.zeros(self.shape(n), dtype=self.dtype)

    def _c(self, a):
        a = np.ascontiguousarray(a, dtype=self.dtype)
        return a, a.ctypes.data_as(ctypes.c_void_p)

    def count(self, a):
        return a.shape[0]

    def from_ints(self, ints):
        """python ints (standard form) -> element array in the crate's in-memory form."""
        n = len(ints)
        std = self.empty(n)
        if self.limbs == 1:
            std[:] = [int(x) for x in ints]
        else:
            for i, x in enumerate(ints):
                for l in range(self.limbs):
                    std[i, l] = (int(x) >> (64 * l)) & 0xFFFFFFFFFFFFFFFF
        out = self.empty(n)
        self._from_std(std.ctypes.data_as(ctypes.c_void_p), out.ctypes.data_as(ctypes.c_void_p), n)
        return out

    def to_ints(self, a):
        a, p = self._c(a)
        n = self.count(a)
        std = self.empty(n)
        self._to_std(p, std.ctypes.data_as(ctypes.c_void_p), n)
        if self.limbs == 1:
            return [int(x) for x in std]
        return [sum(int(std[i, l]) << (64 * l) for l in range(self.limbs)) for i in range(n)]

    def build_fftree(self, n, check_chain=False):
        """FftreeField::build_fftree (src/lib.rs:14-16): None if n exceeds the curve's 2-adicity."""
        assert n > 0 and n & (n - 1) == 0, "n must be a power of two"  # src/lib.rs:41
        h = self._build_fftree(n.bit_length() - 1, int(check_chain))
        return OracleFFTree(self, h) if h else None

    def build_extend_tree(self, e):
        """TEST INFRASTRUCTURE: T_2e with only what FFTree::extend of e evaluations reads (f layers + matrices, src/fftree.rs:72-126,
        341-363) — for configs[3], whose full 2^23 chain would take a quarter of an hour.  Only `.extend` of exactly e values works."""
        assert e > 0 and e & (e - 1) == 0
        h = self._build_extend_tree(e.bit_length())            # log2(2e)
        return OracleFFTree(self, h) if h else None

    def _binop(self, fn, a, b):
        a, pa = self._c(a); b, pb = self._c(b)
        out = np.zeros_like(a)
        fn(pa, pb, out.ctypes.data_as(ctypes.c_void_p), self.count(a))
        return out

    def mul(self, a, b): return self._binop(self._mul_vec, a, b)
    def add(self, a, b): return self._binop(self._add_vec, a, b)
    def sub(self, a, b): return self._binop(self._sub_vec, a, b)

    def inv(self, a):
        a, pa = self._c(a)
        out = np.zeros_like(a)
        self._inv_vec(pa, out.ctypes.data_as(ctypes.c_void_p), self.count(a))
        return out

    def leaves_at(self, n, idx):
        """leaves idx of the n-leaf point set `build_fftree(n)` would use, x(coset_offset + i*G) (src/lib.rs:72-78), computed
        one by one by double-and-add — independent of any tree, for spot checks at sizes whose oracle tree takes minutes"""
        idx = np.ascontiguousarray(idx, dtype=np.uint64)
        out = self.empty(len(idx))
        rc = self._leaves_at(n.bit_length() - 1, idx.ctypes.data_as(ctypes.c_void_p), len(idx), out.ctypes.data_as(ctypes.c_void_p))
        assert rc == 0
        return out

    def horner(self, coeffs, xs):
        c, pc = self._c(coeffs); x, px = self._c(xs)
        out = np.zeros_like(x)
        self._horner(pc, self.count(c), px, self.count(x), out.ctypes.data_as(ctypes.c_void_p))
        return out


class OracleFFTree:
    """CPU restatement of `FFTree<F>` (src/fftree.rs:23-38) — recursive, allocating, single-threaded."""

    def __init__(self, field, handle):
        self.field, self._h = field, handle
        self.n = field._tree_size(handle)

    def __del__(self):
        try:
            self.field._free_tree(self._h)
        except Exception:
            pass

    def _run(self, fn, arr, out_n, *extra):
        a, p = self.field._c(arr)
        out = self.field.empty(out_n)
        rc = fn(self._h, p, out.ctypes.data_as(ctypes.c_void_p), self.field.count(a), *extra)
        if rc == -2:
            raise ValueError("length must be a power of two")
        if rc != 0:
            raise ValueError("FFTree is too small")  # src/fftree.rs:494
        return out

    def extend(self, evals, moiety): return self._run(self.field._extend, evals, len(evals), moiety)
    def mextend(self, evals, moiety): return self._run(self.field._mextend, evals, len(evals), moiety)
    def enter(self, coeffs): return self._run(self.field._enter, coeffs, len(coeffs))
    def exit(self, evals): return self._run(self.field._exit, evals, len(evals))
    def vanish(self, dom): return self._run(self.field._vanish, dom, 2 * len(dom))

    def redc(self, evals, a, moiety):
        e, pe = self.field._c(evals); a_, pa = self.field._c(a)
        out = np.zeros_like(e)
        rc = self.field._redc(self._h, pe, pa, out.ctypes.data_as(ctypes.c_void_p), len(e), moiety)
        assert rc == 0
        return out

    def modular_reduce(self, evals, a, c):
        e, pe = self.field._c(evals); a_, pa = self.field._c(a); c_, pc = self.field._c(c)
        out = np.zeros_like(e)
        rc = self.field._modular_reduce(self._h, pe, pa, pc, out.ctypes.data_as(ctypes.c_void_p), len(e))
        assert rc == 0
        return out

    def degree(self, evals):
        e, pe = self.field._c(evals)
        return int(self.field._degree(self._h, pe, len(e)))

    def table(self, which, m=None):
        """Copy of one table of the subtree with m leaves (default: this tree)."""
        m = self.n if m is None else m
        cnt = ctypes.c_size_t()
        p = self.field._table(self._h, m, which, ctypes.byref(cnt))
        if not p:
            raise ValueError("FFTree is too small")
        ct = ctypes.c_uint64 if self.field.limbs > 1 else ctypes.c_uint32
        flat = np.ctypeslib.as_array(ctypes.cast(p, ctypes.POINTER(ct)), shape=(cnt.value * self.field.limbs,))
        return flat.reshape(self.field.shape(cnt.value)).copy()

    def leaves(self, m=None):
        m = self.n if m is None else m
        return self.table(T_F, m)[m:]

    def rational_map(self, k):
        num, den = self.field.empty(3), self.field.empty(3)
        rc = self.field._ratmap(self._h, k, num.ctypes.data_as(ctypes.c_void_p), den.ctypes.data_as(ctypes.c_void_p))
        assert rc == 0
        return num, den


_fields = {}


def field(name):
    if name not in _fields:
        if name == "secp256k1":
            _fields[name] = _Field(name, "ora_secp_", np.uint64, 4)
        elif name == "m31":
            _fields[name] = _Field(name, "ora_m31_", np.uint32, 1)
        else:
            raise KeyError(name)
    return _fields[name]
