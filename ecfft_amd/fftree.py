"""Host-side mirror of the reference's `FFTree<F>` / `FftreeField` surface over the C-ABI library.

Reference interface mirrored (names, argument meaning, error behaviour):
    trait FftreeField { fn build_fftree(n) -> Option<FFTree<Self>> }      /root/reference/src/lib.rs:14-16
    enum Moiety { S0, S1 }                                                 src/fftree.rs:17-21
    FFTree::{new, extend, enter, exit, subtree_with_size} + pub tables     src/fftree.rs:24-38, 42, 123, 164, 227, 489

Elements are passed in the crate's in-memory representation (secp256k1: uint64[n, 4] Montgomery limbs;
m31: uint32[n]) as numpy arrays (host) or torch CUDA tensors (device-resident, zero copy).  Where the
reference panics this raises: ValueError("FFTree is too small") / AssertionError for non powers of two.
There is no CPU fallback: if libecfft_hip.so or a HIP device is missing the call fails loudly.
"""
import ctypes
import enum
import os

import numpy as np

from . import build as _build

_DIR = os.path.dirname(os.path.abspath(__file__))

OK, ERR_NOT_POW2, ERR_TREE_TOO_SMALL, ERR_TREE_TOO_LARGE, ERR_HIP, ERR_BAD_ARG = range(6)
MEM_HOST, MEM_DEVICE = 0, 1
TBL_F, TBL_RECOMBINE, TBL_DECOMPOSE, TBL_XNN_S, TBL_XNN_S_INV, TBL_Z0_S1, TBL_Z1_S0, TBL_Z0_INV_S1, TBL_Z1_INV_S0, TBL_Z0Z0, TBL_Z1Z1 = range(11)

EXPORTS = ["ecfft_elem_size", "ecfft_build_fftree", "ecfft_fftree_new", "ecfft_ctx_destroy", "ecfft_tree_size",
           "ecfft_field", "ecfft_enter", "ecfft_exit", "ecfft_extend", "ecfft_tree_table", "ecfft_build_points",
           "ecfft_device_info", "ecfft_profile_enable", "ecfft_profile_classes", "ecfft_profile_read",
           "ecfft_extend_top_cyclic", "ecfft_extend_local_block",
           "ecfft_mul_ceiling", "ecfft_elems_to_standard", "ecfft_elems_from_standard", "ecfft_table_fma", "ecfft_enter_many", "ecfft_exit_many", "ecfft_mextend", "ecfft_redc", "ecfft_modular_reduce", "ecfft_vanish", "ecfft_degree",
           "ecfft_comm_get_unique_id", "ecfft_comm_init_rank", "ecfft_comm_init_callback", "ecfft_comm_destroy", "ecfft_comm_rank", "ecfft_comm_world",
           "ecfft_comm_stats_enable", "ecfft_comm_stats_read", "ecfft_extend_sharded", "ecfft_enter_sharded", "ecfft_exit_sharded", "ecfft_device_copy", "ecfft_shader_clock", "ecfft_device_alloc", "ecfft_device_free", "ecfft_device_sync", "ecfft_build_extend_shard", "ecfft_ctx_device_bytes", "ecfft_extend_sharded_layout", "ecfft_build_enter_shard", "ecfft_build_exit_shard", "ecfft_build_exit_shard_opts",
           "ecfft_fftree_serialize", "ecfft_fftree_deserialize", "ecfft_tree_rational_maps", "ecfft_ctx_trim", "ecfft_comm_abort", "ecfft_comm_set_rccl_library", "ecfft_comm_set_link_striping"]

# include/ecfft_hip_hooks.h: only in a build with -DECFFT_TEST_HOOKS (tests/hooks/libecfft_hip_hooks.so), never in the shipped library
HOOK_EXPORTS = ['ecfft_selftest_field', 'ecfft_selfcheck_pointwise_z', 'ecfft_test_fail_next_collective', 'ecfft_selftest_blk16', 'ecfft_selftest_blk16_small', 'ecfft_test_fail_build_rank', 'ecfft_comm_init_projection', 'ecfft_selftest_blk32', 'ecfft_ctx_low_map']

EXCHANGE_FN = ctypes.CFUNCTYPE(ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_void_p),
                               ctypes.POINTER(ctypes.c_size_t), ctypes.c_int, ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_void_p),
                               ctypes.POINTER(ctypes.c_size_t), ctypes.c_void_p)


class Moiety(enum.IntEnum):
    S0 = 0
    S1 = 1


class EcfftError(RuntimeError):
    pass


_lib = None
HOOKS_LIB = os.path.join(os.path.dirname(_DIR), "tests", "hooks", "libecfft_hip_hooks.so")


def _bind(L):
    vp, sz, ci = ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int
    L.ecfft_elem_size.restype, L.ecfft_elem_size.argtypes = sz, [ci]
    L.ecfft_build_fftree.restype, L.ecfft_build_fftree.argtypes = ci, [ci, sz, ci, ctypes.POINTER(vp)]
    L.ecfft_fftree_new.restype, L.ecfft_fftree_new.argtypes = ci, [ci, vp, sz, vp, vp, ci, ctypes.POINTER(vp)]
    L.ecfft_ctx_destroy.restype, L.ecfft_ctx_destroy.argtypes = None, [vp]
    L.ecfft_tree_size.restype, L.ecfft_tree_size.argtypes = sz, [vp]
    L.ecfft_field.restype, L.ecfft_field.argtypes = ci, [vp]
    L.ecfft_enter.restype, L.ecfft_enter.argtypes = ci, [vp, vp, vp, sz, ci, vp]
    L.ecfft_exit.restype, L.ecfft_exit.argtypes = ci, [vp, vp, vp, sz, ci, vp]
    L.ecfft_extend.restype, L.ecfft_extend.argtypes = ci, [vp, vp, vp, sz, ci, sz, ci, vp]
    L.ecfft_tree_table.restype, L.ecfft_tree_table.argtypes = ci, [vp, sz, ci, vp, sz, ctypes.POINTER(sz)]
    L.ecfft_build_points.restype, L.ecfft_build_points.argtypes = ci, [ci, sz, vp, vp, vp]
    L.ecfft_device_info.restype, L.ecfft_device_info.argtypes = ci, [ci, ctypes.c_char_p, sz]
    L.ecfft_extend_top_cyclic.restype, L.ecfft_extend_top_cyclic.argtypes = ci, [vp, vp, sz, ci, ctypes.c_uint, ctypes.c_uint, ci, ci, vp]
    L.ecfft_extend_local_block.restype, L.ecfft_extend_local_block.argtypes = ci, [vp, vp, sz, ci, ctypes.c_uint, ci, vp]
    L.ecfft_mul_ceiling.restype, L.ecfft_mul_ceiling.argtypes = ci, [ci, ci, ci, ctypes.POINTER(ctypes.c_double)]
    L.ecfft_elems_to_standard.restype, L.ecfft_elems_to_standard.argtypes = ci, [ci, vp, vp, sz]
    L.ecfft_elems_from_standard.restype, L.ecfft_elems_from_standard.argtypes = ci, [ci, vp, vp, sz]
    L.ecfft_table_fma.restype, L.ecfft_table_fma.argtypes = ci, [vp, vp, vp, vp, sz, sz, ci, sz, sz, ci, ci, vp]
    L.ecfft_enter_many.restype, L.ecfft_enter_many.argtypes = ci, [vp, vp, vp, sz, sz, ci, vp]
    L.ecfft_exit_many.restype, L.ecfft_exit_many.argtypes = ci, [vp, vp, vp, sz, sz, ci, vp]
    L.ecfft_mextend.restype, L.ecfft_mextend.argtypes = ci, [vp, vp, vp, sz, ci, sz, ci, vp]
    L.ecfft_redc.restype, L.ecfft_redc.argtypes = ci, [vp, vp, vp, vp, sz, ci, ci, vp]
    L.ecfft_modular_reduce.restype, L.ecfft_modular_reduce.argtypes = ci, [vp, vp, vp, vp, vp, sz, ci, vp]
    L.ecfft_vanish.restype, L.ecfft_vanish.argtypes = ci, [vp, vp, vp, sz, ci, vp]
    L.ecfft_degree.restype, L.ecfft_degree.argtypes = ci, [vp, vp, sz, ci, vp, ctypes.POINTER(sz)]
    L.ecfft_comm_get_unique_id.restype, L.ecfft_comm_get_unique_id.argtypes = ci, [vp]
    L.ecfft_comm_init_rank.restype, L.ecfft_comm_init_rank.argtypes = ci, [vp, ci, ci, ci, ctypes.POINTER(vp)]
    L.ecfft_comm_init_callback.restype, L.ecfft_comm_init_callback.argtypes = ci, [ci, ci, ci, EXCHANGE_FN, vp, ctypes.POINTER(vp)]
    L.ecfft_comm_destroy.restype, L.ecfft_comm_destroy.argtypes = None, [vp]
    L.ecfft_comm_rank.restype, L.ecfft_comm_rank.argtypes = ci, [vp]
    L.ecfft_comm_world.restype, L.ecfft_comm_world.argtypes = ci, [vp]
    L.ecfft_comm_stats_enable.restype, L.ecfft_comm_stats_enable.argtypes = ci, [vp, ci]
    L.ecfft_comm_stats_read.restype, L.ecfft_comm_stats_read.argtypes = ci, [vp, ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_double)]
    L.ecfft_extend_sharded.restype, L.ecfft_extend_sharded.argtypes = ci, [vp, vp, vp, vp, sz, ci, vp]
    L.ecfft_enter_sharded.restype, L.ecfft_enter_sharded.argtypes = ci, [vp, vp, vp, vp, sz, vp]
    L.ecfft_extend_sharded_layout.restype, L.ecfft_extend_sharded_layout.argtypes = ci, [vp, vp, vp, vp, sz, ci, ci, ci, vp]
    L.ecfft_build_enter_shard.restype, L.ecfft_build_enter_shard.argtypes = ci, [ci, sz, ci, ci, ci, ctypes.POINTER(vp)]
    L.ecfft_build_exit_shard.restype, L.ecfft_build_exit_shard.argtypes = ci, [ci, sz, ci, vp, ctypes.POINTER(vp)]
    L.ecfft_build_exit_shard_opts.restype, L.ecfft_build_exit_shard_opts.argtypes = ci, [ci, sz, ci, vp, ci, ctypes.POINTER(vp)]
    L.ecfft_ctx_device_bytes.restype, L.ecfft_ctx_device_bytes.argtypes = sz, [vp]
    L.ecfft_build_extend_shard.restype, L.ecfft_build_extend_shard.argtypes = ci, [ci, sz, ci, ci, ci, ctypes.POINTER(vp)]
    L.ecfft_exit_sharded.restype, L.ecfft_exit_sharded.argtypes = ci, [vp, vp, vp, vp, sz, vp]
    L.ecfft_fftree_serialize.restype, L.ecfft_fftree_serialize.argtypes = ci, [vp, ci, vp, sz, ctypes.POINTER(sz)]
    L.ecfft_fftree_deserialize.restype, L.ecfft_fftree_deserialize.argtypes = ci, [ci, vp, sz, ci, ci, ci, ctypes.POINTER(vp)]
    L.ecfft_tree_rational_maps.restype, L.ecfft_tree_rational_maps.argtypes = ci, [vp, vp, vp]
    L.ecfft_ctx_trim.restype, L.ecfft_ctx_trim.argtypes = ci, [vp]
    L.ecfft_device_copy.restype, L.ecfft_device_copy.argtypes = ci, [vp, vp, sz, ci]
    L.ecfft_shader_clock.restype, L.ecfft_shader_clock.argtypes = ci, [ci, ci, ctypes.POINTER(ctypes.c_double)]
    L.ecfft_profile_enable.restype, L.ecfft_profile_enable.argtypes = ci, [vp, ci]
    L.ecfft_profile_classes.restype, L.ecfft_profile_classes.argtypes = ci, []
    L.ecfft_profile_read.restype, L.ecfft_profile_read.argtypes = ci, [vp, ci, ctypes.c_char_p, sz, ctypes.POINTER(ctypes.c_uint64),
                                                                           ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_double)]
    L.ecfft_comm_set_rccl_library.restype, L.ecfft_comm_set_rccl_library.argtypes = ci, [ctypes.c_char_p]
    L.ecfft_comm_set_link_striping.restype, L.ecfft_comm_set_link_striping.argtypes = ci, [vp, sz]
    L.has_hooks = hasattr(L, "ecfft_selftest_field")
    if L.has_hooks:      # include/ecfft_hip_hooks.h (test builds only)
        L.ecfft_selftest_field.restype, L.ecfft_selftest_field.argtypes = ci, [ci, ci, vp, vp, vp, vp, sz, ci]
        L.ecfft_selfcheck_pointwise_z.restype, L.ecfft_selfcheck_pointwise_z.argtypes = ctypes.c_long, [vp, sz]
        L.ecfft_test_fail_next_collective.restype, L.ecfft_test_fail_next_collective.argtypes = ci, [vp]
        L.ecfft_test_fail_build_rank.restype, L.ecfft_test_fail_build_rank.argtypes = ci, [ci]
        L.ecfft_selftest_blk16.restype, L.ecfft_selftest_blk16.argtypes = ci, [vp, vp, vp, sz, ci]
        L.ecfft_selftest_blk16_small.restype, L.ecfft_selftest_blk16_small.argtypes = ci, [vp, vp, vp, sz, ci, ci]
        L.ecfft_selftest_blk32.restype, L.ecfft_selftest_blk32.argtypes = ci, [vp, vp, vp, sz, ci]
        L.ecfft_ctx_low_map.restype, L.ecfft_ctx_low_map.argtypes = ci, [vp, ci]
        L.ecfft_comm_init_projection.restype = ci
        L.ecfft_comm_init_projection.argtypes = [ci, ci, ci, ctypes.c_double, ctypes.c_double, ctypes.POINTER(vp)]
    return L


def _load(path):
    # PyTorch-ROCm bundles its own libamdhip64.so.7 / libhsa-runtime64; load it FIRST so that this
    # library binds to the same HIP runtime instance (two runtimes in one process cannot share the
    # GPU: torch then reports "No HIP GPUs are available").  Plain C users link /opt/rocm directly.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    return _bind(ctypes.CDLL(path))


def lib():
    """Loads (building if needed) the HIP extension — the SHIPPED library unless ECFFT_LIB names another build (tuning variants;
    a variable of this Python mirror, not of the C library).  Raises if it cannot be built or loaded."""
    global _lib
    if _lib is None:
        path = os.environ.get("ECFFT_LIB") or os.path.join(_DIR, "libecfft_hip.so")
        if not os.path.exists(path):
            _build.build()
        _lib = _load(path)
    return _lib


class use_library:
    """`with use_library(path):` — every call of this module goes to that build of the library for the duration (tests and
    measurement tools: the hooks build, tests/hooks/libecfft_hip_hooks.so).  Contexts and communicators must be created AND used
    inside the same block: a handle belongs to the library instance that made it."""
    _loaded = {}

    def __init__(self, path):
        self.path = os.path.abspath(path)

    def __enter__(self):
        global _lib
        self.prev = _lib
        if self.path not in use_library._loaded:
            use_library._loaded[self.path] = _load(self.path)
        _lib = use_library._loaded[self.path]
        return _lib

    def __exit__(self, *exc):
        global _lib
        _lib = self.prev
        return False


def use_hooks_library():
    """the hooks build (include/ecfft_hip_hooks.h), built on demand by tests/hooks/build_hooks.py"""
    if not os.path.exists(HOOKS_LIB) or _build.stale(HOOKS_LIB):
        _build.build(out=HOOKS_LIB, defines=("ECFFT_TEST_HOOKS",))
    return use_library(HOOKS_LIB)


def _check(rc):
    if rc == OK:
        return
    if rc == ERR_NOT_POW2:
        raise AssertionError("length must be a power of two")        # assert!(n.is_power_of_two())
    if rc == ERR_TREE_TOO_SMALL:
        raise ValueError("FFTree is too small")                        # panic!("FFTree is too small")
    if rc == ERR_HIP:
        raise EcfftError("HIP failure (no usable MI355X device, or a runtime error) — there is no CPU fallback")
    raise EcfftError(f"ecfft error {rc}")


class Field:
    """One of the reference's two `FftreeField` implementors."""

    def __init__(self, name, field_id, dtype, limbs):
        self.name, self.id, self.dtype, self.limbs = name, field_id, np.dtype(dtype), limbs

    @property
    def elem_bytes(self):
        return self.dtype.itemsize * self.limbs

    def shape(self, n):
        return (n, self.limbs) if self.limbs > 1 else (n,)

    def build_fftree(self, n, device=0):
        """`F::build_fftree(n)`: None if n exceeds the curve's 2-adicity (src/lib.rs:62-64, src/ec.rs:513-515)."""
        h = ctypes.c_void_p()
        rc = lib().ecfft_build_fftree(self.id, n, device, ctypes.byref(h))
        if rc == ERR_TREE_TOO_LARGE:
            return None
        _check(rc)
        return FFTree(self, h, device)

    def build_extend_shard(self, e, world, rank, device=0):
        """Sharded EXTEND-only context (include/ecfft_hip.h ecfft_build_extend_shard): this rank's share of the tables of ONE
        EXTEND of e evaluations over `world` GPUs; only `extend_sharded` works on it.  None if T_2e is too large for the curve."""
        h = ctypes.c_void_p()
        rc = lib().ecfft_build_extend_shard(self.id, e, device, world, rank, ctypes.byref(h))
        if rc == ERR_TREE_TOO_LARGE:
            return None
        _check(rc)
        return FFTree(self, h, device)

    def build_enter_shard(self, n, world, rank, device=0):
        """Sharded ENTER-only context (ecfft_build_enter_shard): the chain up to n/world plus this rank's share of the top
        log2(world) trees; only `enter_sharded` works on it.  None if T_n is too large for the curve."""
        h = ctypes.c_void_p()
        rc = lib().ecfft_build_enter_shard(self.id, n, device, world, rank, ctypes.byref(h))
        if rc == ERR_TREE_TOO_LARGE:
            return None
        _check(rc)
        return FFTree(self, h, device)

    def build_exit_shard(self, n, comm, device=0, min_memory=False):
        """Sharded EXIT-only context (ecfft_build_exit_shard[_opts]) — COLLECTIVE over the ranks of `comm`: the chain up to n/world plus
        this rank's share of the top trees, z0z0_rem_xnn_s built distributed; only `exit_sharded` works on it.  min_memory: never
        keep T_2c for the redundant pair level (ECFFT_EXIT_SHARD_MIN_MEMORY)."""
        h = ctypes.c_void_p()
        rc = lib().ecfft_build_exit_shard_opts(self.id, n, device, comm._h, 1 if min_memory else 0, ctypes.byref(h))
        if rc == ERR_TREE_TOO_LARGE:
            return None
        _check(rc)
        return FFTree(self, h, device)

    def selftest(self, op, a, b, c=None, device=0):
        """device field arithmetic on raw residues (test hook): op 0 a*b+c, 1 a*b, 2 a-b, 3 a+b, 4/5 the kernels' table multiply a*b+c / a*b"""
        a = np.ascontiguousarray(a, self.dtype); b = np.ascontiguousarray(b, self.dtype)
        cc = None if c is None else np.ascontiguousarray(c, self.dtype)
        out = np.empty_like(a)
        _check(lib().ecfft_selftest_field(self.id, op, a.ctypes.data, b.ctypes.data, None if cc is None else cc.ctypes.data, out.ctypes.data, a.shape[0], device))
        return out

    def mul_ceiling(self, waves_per_simd=4, device=0):
        """field multiplies per second of the kernels' table multiply as a bare dependent chain on the whole chip"""
        r = ctypes.c_double()
        _check(lib().ecfft_mul_ceiling(self.id, device, waves_per_simd, ctypes.byref(r)))
        return r.value

    def shader_clock_mhz(self, device=0):
        """effective shader clock while the whole chip runs this field's table multiply"""
        r = ctypes.c_double()
        _check(lib().ecfft_shader_clock(self.id, device, ctypes.byref(r)))
        return r.value

    def to_standard(self, a):
        """in-memory elements -> standard-form little-endian integers (same array shape)"""
        a = np.ascontiguousarray(a, self.dtype); out = np.empty_like(a)
        _check(lib().ecfft_elems_to_standard(self.id, a.ctypes.data, out.ctypes.data, a.shape[0]))
        return out

    def from_standard(self, a):
        a = np.ascontiguousarray(a, self.dtype); out = np.empty_like(a)
        _check(lib().ecfft_elems_from_standard(self.id, a.ctypes.data, out.ctypes.data, a.shape[0]))
        return out

    def build_points(self, n):
        """host-only: (f[2n], map_num[log n, 3], map_den[log n, 3]) — leaves are f[n:]."""
        ln = max(n.bit_length() - 1, 0)
        f = np.zeros(self.shape(2 * n), self.dtype)
        num = np.zeros(self.shape(3 * max(ln, 1)), self.dtype)
        den = np.zeros(self.shape(3 * max(ln, 1)), self.dtype)
        rc = lib().ecfft_build_points(self.id, n, f.ctypes.data, num.ctypes.data, den.ctypes.data)
        if rc == ERR_TREE_TOO_LARGE:
            return None
        _check(rc)
        return f, num[:3 * ln], den[:3 * ln]

    def new_fftree(self, leaves, map_num, map_den, device=0):
        """`FFTree::new(leaves, rational_maps)` (src/fftree.rs:42-70)."""
        leaves = np.ascontiguousarray(leaves, self.dtype)
        map_num = np.ascontiguousarray(map_num, self.dtype); map_den = np.ascontiguousarray(map_den, self.dtype)
        n = leaves.shape[0]
        h = ctypes.c_void_p()
        _check(lib().ecfft_fftree_new(self.id, leaves.ctypes.data, n, map_num.ctypes.data, map_den.ctypes.data, device, ctypes.byref(h)))
        return FFTree(self, h, device, maps=(map_num, map_den))


def deserialize_fftree(field, data, compress, device=0, verify=True):
    """`FFTree::deserialize_compressed / deserialize_uncompressed` (src/fftree.rs:600-660) through the C ABI
    (ecfft_fftree_deserialize: bounds-checked parse, rebuild on the GPU, optional table-by-table verification)"""
    data = bytes(data)
    h = ctypes.c_void_p()
    rc = lib().ecfft_fftree_deserialize(field.id, data, len(data), int(bool(compress)), device, int(bool(verify)), ctypes.byref(h))
    if rc in (ERR_BAD_ARG, ERR_NOT_POW2):        # a file whose `f` length is not a power of two is malformed too
        raise ValueError("malformed FFTree file (or its tables disagree with its point set)")
    _check(rc)
    t = FFTree(field, h, device)
    return t


secp256k1 = Field("secp256k1", 0, np.uint64, 4)
m31 = Field("m31", 1, np.uint32, 1)
FIELDS = {"secp256k1": secp256k1, "m31": m31}


def _is_torch(x):
    return type(x).__module__.startswith("torch")


class FFTree:
    """Device-resident `FFTree<F>` (whole subtree chain).  `&self` methods, immutable after build."""

    def __init__(self, field, handle, device, maps=None):
        self.field, self._h, self.device = field, handle, device
        # the library that created the handle serves every later call on it, __del__ included: with use_library /
        # use_hooks_library a handle can outlive the with-block it was made in, and lib() would then name another .so instance
        self._L = lib()
        self.n = self._L.ecfft_tree_size(handle)
        self._from_build = maps is None          # build_fftree: the maps are those of build_points(n)
        if maps is not None:
            self._num, self._den = maps

    @property
    def device_bytes(self):
        """HBM the context holds between calls (tables + transform scratch + pooled temporaries)"""
        return self._L.ecfft_ctx_device_bytes(self._h)

    def trim(self):
        """return the pooled temporaries of the algorithm wrappers to the device (ecfft_ctx_trim)"""
        _check(self._L.ecfft_ctx_trim(self._h))

    def __del__(self):
        try:
            if self._h:
                self._L.ecfft_ctx_destroy(self._h)
                self._h = None
        except Exception:
            pass

    # ---- plumbing --------------------------------------------------------------------------
    def _io(self, x, out_like=True):
        """returns (in_ptr, out_obj, out_ptr, mem, stream, count_elems)"""
        if _is_torch(x):
            import torch
            assert x.is_cuda and x.is_contiguous(), "device tensors must be contiguous CUDA tensors"
            out = torch.empty_like(x)
            stream = torch.cuda.current_stream(x.device).cuda_stream
            return x.data_ptr(), out, out.data_ptr(), MEM_DEVICE, stream, x.shape[0]
        a = np.ascontiguousarray(x, self.field.dtype)
        out = np.empty_like(a)
        return a.ctypes.data, out, out.ctypes.data, MEM_HOST, None, a.shape[0]

    # ---- the path --------------------------------------------------------------------------
    def enter(self, coeffs, count=1):
        """coefficients -> evaluations on the leaves of T_len (src/fftree.rs:164-167); count > 1: that many
        polynomials laid end to end (batched form, no reference counterpart)."""
        pin, out, pout, mem, stream, n = self._io(coeffs)
        assert n % count == 0
        _check(self._L.ecfft_enter_many(self._h, pin, pout, n // count, count, mem, stream))
        return out

    def exit(self, evals, count=1):
        """evaluations -> coefficients (src/fftree.rs:227-230)."""
        pin, out, pout, mem, stream, n = self._io(evals)
        assert n % count == 0
        _check(self._L.ecfft_exit_many(self._h, pin, pout, n // count, count, mem, stream))
        return out

    def extend(self, evals, moiety, count=1):
        """extends evaluations onto the TARGET moiety (src/fftree.rs:123-126); `count` > 1 treats the
        input as that many vectors laid end to end (batched form, no reference counterpart)."""
        pin, out, pout, mem, stream, total = self._io(evals)
        assert total % count == 0
        _check(self._L.ecfft_extend(self._h, pin, pout, total // count, int(moiety), count, mem, stream))
        return out

    # ---- the remaining FFTree algorithms (host numpy arrays; synchronous) ---------------------
    def _np(self, x):
        return np.ascontiguousarray(x, self.field.dtype)

    def mextend(self, evals, moiety):
        """src/fftree.rs:138-141"""
        a = self._np(evals); out = np.empty_like(a)
        _check(self._L.ecfft_mextend(self._h, a.ctypes.data, out.ctypes.data, a.shape[0], int(moiety), 1, MEM_HOST, None))
        return out

    def redc_z0(self, evals, a):
        """src/fftree.rs:264-267"""
        return self._redc(evals, a, Moiety.S0)

    def redc_z1(self, evals, a):
        """src/fftree.rs:272-275"""
        return self._redc(evals, a, Moiety.S1)

    def _redc(self, evals, a, moiety):
        e = self._np(evals); a = self._np(a); out = np.empty_like(e)
        assert a.shape[0] == e.shape[0]
        _check(self._L.ecfft_redc(self._h, e.ctypes.data, a.ctypes.data, out.ctypes.data, e.shape[0], int(moiety), MEM_HOST, None))
        return out

    def modular_reduce(self, evals, a, c):
        """src/fftree.rs:286-289"""
        e = self._np(evals); a = self._np(a); c = self._np(c); out = np.empty_like(e)
        assert a.shape[0] == e.shape[0] == c.shape[0]
        _check(self._L.ecfft_modular_reduce(self._h, e.ctypes.data, a.ctypes.data, c.ctypes.data, out.ctypes.data, e.shape[0], MEM_HOST, None))
        return out

    def vanish(self, domain):
        """src/fftree.rs:313-316"""
        d = self._np(domain)
        out = np.empty(self.field.shape(2 * d.shape[0]), self.field.dtype)
        _check(self._L.ecfft_vanish(self._h, d.ctypes.data, out.ctypes.data, d.shape[0], MEM_HOST, None))
        return out

    def degree(self, evals):
        """src/fftree.rs:195-198"""
        e = self._np(evals); deg = ctypes.c_size_t()
        _check(self._L.ecfft_degree(self._h, e.ctypes.data, e.shape[0], MEM_HOST, None, ctypes.byref(deg)))
        return deg.value

    def table_fma(self, x, y, m, which, t_off, t_stride, mode):
        """out[i] = f(x[i], y[i], T_m.table[which][t_off + i*t_stride]); mode 0: x*T, 1: x*T + y, 2: y - x*T, 3: (y - x)*T"""
        pin, out, pout, mem, stream, n = self._io(x)
        py, keep = None, None
        if y is not None:
            keep = y if _is_torch(y) else np.ascontiguousarray(y, self.field.dtype)     # keep the buffer alive across the call
            py = keep.data_ptr() if _is_torch(y) else keep.ctypes.data
        _check(self._L.ecfft_table_fma(self._h, pout, pin, py, n, m, which, t_off, t_stride, mode, mem, stream))
        return out

    # ---- shards of one EXTEND split over P GPUs (in place; see tests/split_model.py for the model) -----
    def _inplace(self, x):
        if _is_torch(x):
            import torch
            assert x.is_cuda and x.is_contiguous()
            return x.data_ptr(), MEM_DEVICE, torch.cuda.current_stream(x.device).cuda_stream
        assert isinstance(x, np.ndarray) and x.flags["C_CONTIGUOUS"] and x.dtype == self.field.dtype
        return x.ctypes.data, MEM_HOST, None

    def extend_top_cyclic(self, shard, e, moiety, log_p, rank, recombine):
        ptr, mem, stream = self._inplace(shard)
        _check(self._L.ecfft_extend_top_cyclic(self._h, ptr, e, int(moiety), log_p, rank, int(recombine), mem, stream))

    def extend_local_block(self, shard, e, moiety, log_p):
        ptr, mem, stream = self._inplace(shard)
        _check(self._L.ecfft_extend_local_block(self._h, ptr, e, int(moiety), log_p, mem, stream))

    # ---- ONE transform split over the GPUs of a communicator (device tensors: this rank's block shard) -------------
    def _sharded(self, fn, comm, x, length, *extra):
        import torch
        assert _is_torch(x) and x.is_cuda and x.is_contiguous(), "sharded transforms take contiguous CUDA tensors"
        out = torch.empty_like(x)
        stream = torch.cuda.current_stream(x.device).cuda_stream
        _check(fn(self._h, comm._h, x.data_ptr(), out.data_ptr(), length, *extra, stream))
        return out

    def extend_sharded(self, comm, x_block, e, moiety, cyclic_in=False, cyclic_out=False):
        """FFTree::extend of ONE length-e vector held block-distributed over the ranks of `comm` (C++ / RCCL path);
        cyclic_in / cyclic_out: the shard on that side is cyclic (local j' = global j' * world + rank), one exchange fewer each"""
        if not (cyclic_in or cyclic_out):
            return self._sharded(self._L.ecfft_extend_sharded, comm, x_block, e, int(moiety))
        return self._sharded(self._L.ecfft_extend_sharded_layout, comm, x_block, e, int(moiety), int(cyclic_in), int(cyclic_out))

    def enter_sharded(self, comm, x_block, n):
        return self._sharded(self._L.ecfft_enter_sharded, comm, x_block, n)

    def exit_sharded(self, comm, y_block, n):
        return self._sharded(self._L.ecfft_exit_sharded, comm, y_block, n)

    # ---- benchmarking aid -------------------------------------------------------------------
    def low_map(self, direction):
        """32 / 16 / 0: the composite map the 1024-element low-level kernels run for the lowest ENTER (0) / EXIT (1) levels"""
        return self._L.ecfft_ctx_low_map(self._h, int(direction))

    def profile(self, on):
        _check(self._L.ecfft_profile_enable(self._h, int(on)))

    def profile_read(self):
        """[{name, launches, ms, alg_bytes}] per kernel class, from HIP events on the launch stream."""
        out = []
        for c in range(self._L.ecfft_profile_classes()):
            name = ctypes.create_string_buffer(64); n = ctypes.c_uint64(); ms = ctypes.c_double(); by = ctypes.c_double()
            _check(self._L.ecfft_profile_read(self._h, c, name, 64, ctypes.byref(n), ctypes.byref(ms), ctypes.byref(by)))
            out.append({"name": name.value.decode(), "launches": n.value, "ms": ms.value, "alg_bytes": by.value})
        return out

    # ---- pub fields ------------------------------------------------------------------------
    def table(self, which, m=None):
        m = self.n if m is None else m
        cnt = ctypes.c_size_t()
        _check(self._L.ecfft_tree_table(self._h, m, which, None, 0, ctypes.byref(cnt)))
        out = np.zeros(self.field.shape(cnt.value), self.field.dtype)
        _check(self._L.ecfft_tree_table(self._h, m, which, out.ctypes.data, cnt.value, ctypes.byref(cnt)))
        return out

    def leaves(self, m=None):
        m = self.n if m is None else m
        return self.table(TBL_F, m)[m:]

    def serialize(self, compress):
        """`FFTree::serialize_compressed` (compress=True) / `serialize_uncompressed` (src/fftree.rs:510-554) through the C ABI"""
        ln = ctypes.c_size_t()
        _check(self._L.ecfft_fftree_serialize(self._h, int(bool(compress)), None, 0, ctypes.byref(ln)))
        buf = ctypes.create_string_buffer(ln.value)
        _check(self._L.ecfft_fftree_serialize(self._h, int(bool(compress)), buf, ln.value, ctypes.byref(ln)))
        return buf.raw[:ln.value]

    def rational_map(self, k):
        """rational_maps[k] (src/fftree.rs:28) as (numerator[3], denominator[3]) coefficients, low -> high, zero padded"""
        if not hasattr(self, "_maps"):
            ln = max(self.n.bit_length() - 1, 1)
            num = np.zeros(self.field.shape(3 * ln), self.field.dtype); den = np.zeros(self.field.shape(3 * ln), self.field.dtype)
            _check(self._L.ecfft_tree_rational_maps(self._h, num.ctypes.data, den.ctypes.data))
            self._maps = (num, den)
        num, den = self._maps
        return num[3 * k:3 * k + 3], den[3 * k:3 * k + 3]

    def subtree_with_size(self, n):
        """src/fftree.rs:489-496 — the chain lives in one context, so this is a size check + view."""
        assert n > 0 and n & (n - 1) == 0
        if n > self.n:
            raise ValueError("FFTree is too small")
        return _SubtreeView(self, n)


class _SubtreeView:
    def __init__(self, tree, n):
        self.tree, self.n = tree, n

    def leaves(self):
        return self.tree.leaves(self.n)

    def table(self, which):
        return self.tree.table(which, self.n)

    def enter(self, c):
        if len(c) > self.n:
            raise ValueError("FFTree is too small")
        return self.tree.enter(c)

    def exit(self, e):
        if len(e) > self.n:
            raise ValueError("FFTree is too small")
        return self.tree.exit(e)


def device_info(device=0):
    buf = ctypes.create_string_buffer(256)
    lib().ecfft_device_info(device, buf, 256)
    return buf.value.decode()
