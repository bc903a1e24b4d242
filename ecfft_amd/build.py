"""Builds the HIP shared library libecfft_hip.so in-tree (hipcc cross-compiles for gfx950 without a GPU)."""
import os
import subprocess

_DIR = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(_DIR, "csrc", "ecfft_capi.hip")
LIB = os.path.join(_DIR, "libecfft_hip.so")
DEPS = [os.path.join(_DIR, "csrc", f) for f in
        ("ecfft_capi.hip", "device_tree.h", "kernels.h", "host_curve.h", "field_secp256k1.h", "field_m31.h", "secp256k1_mul_gfx950.inc", "transport.h", "mfma_blk16.h", "wire_parse.h")]
DEPS.append(os.path.join(os.path.dirname(_DIR), "include", "ecfft_hip.h"))
DEPS.append(os.path.join(os.path.dirname(_DIR), "include", "ecfft_hip_hooks.h"))


def stale(lib=LIB):
    if not os.path.exists(lib):
        return True
    t = os.path.getmtime(lib)
    return any(os.path.getmtime(d) > t for d in DEPS if os.path.exists(d))


def build(force=False, verbose=False, out=None, defines=()):
    """out / defines: another build of the same sources (the tests' hooks library: tests/hooks/build_hooks.py passes
    -DECFFT_TEST_HOOKS; tuning variants pass their -DECFFT_* knobs).  The default is the shipped library."""
    lib = out or LIB
    if not force and not stale(lib):
        return lib
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-Wno-pass-failed"] + [f"-D{d}" for d in defines] + ["-o", lib, SRC]
    if verbose:
        print(" ".join(cmd))
    subprocess.run(cmd, check=True)
    return lib


if __name__ == "__main__":
    build(force=True, verbose=True)
