"""TEST INFRASTRUCTURE: the hooks build of the library — the same sources as ecfft_amd/libecfft_hip.so compiled with
-DECFFT_TEST_HOOKS, which adds the entry points of include/ecfft_hip_hooks.h (device field arithmetic and matrix-core maps on
explicit operands, failure injection for the sharded calls, the projection transport of tools/split_project.py) and makes the
library read the A/B switches of the tuning experiments from the environment.  The shipped library has none of this."""
import os
import sys

_DIR = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(_DIR))
LIB = os.path.join(_DIR, "libecfft_hip_hooks.so")


def build(force=False, verbose=False):
    if ROOT not in sys.path:
        sys.path.insert(0, ROOT)
    from ecfft_amd import build as b
    return b.build(force=force, verbose=verbose, out=LIB, defines=("ECFFT_TEST_HOOKS",))


if __name__ == "__main__":
    print(build(force=True, verbose=True))
