"""Builds the HIP shared library libecfft_hip.so in-tree (hipcc cross-compiles for gfx950 without a GPU)."""
import os
import subprocess

_DIR = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(_DIR, "csrc", "ecfft_capi.hip")
LIB = os.path.join(_DIR, "libecfft_hip.so")
DEPS = [os.path.join(_DIR, "csrc", f) for f in
        ("ecfft_capi.hip", "device_tree.h", "kernels.h", "host_curve.h", "field_secp256k1.h", "field_m31.h", "secp256k1_mul_gfx950.inc", "transport.h", "mfma_blk16.h")]
DEPS.append(os.path.join(os.path.dirname(_DIR), "include", "ecfft_hip.h"))


def stale():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    return any(os.path.getmtime(d) > t for d in DEPS if os.path.exists(d))


def build(force=False, verbose=False):
    if not force and not stale():
        return LIB
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-Wno-pass-failed", "-o", LIB, SRC]
    if verbose:
        print(" ".join(cmd))
    subprocess.run(cmd, check=True)
    return LIB


if __name__ == "__main__":
    build(force=True, verbose=True)
