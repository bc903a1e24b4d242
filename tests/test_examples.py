"""The C++ host mirror (include/ecfft_fftree.hpp) and the two example programs built on it: they must compile against
the C ABI with a plain g++ (CPU check), and run to completion with a passing round trip on a GPU box."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXAMPLES = ["interp_eval", "bench_fftree", "sharded_extend"]


def _build(name, outdir):
    import ecfft_amd.build
    ecfft_amd.build.build()
    exe = os.path.join(str(outdir), name)
    libdir = os.path.join(ROOT, "ecfft_amd")
    cmd = ["g++", "-O2", "-std=c++17", "-I" + os.path.join(ROOT, "include"), os.path.join(ROOT, "examples", name + ".cpp"),
           "-L" + libdir, "-lecfft_hip", "-Wl,-rpath," + libdir, "-Wl,--allow-shlib-undefined", "-o", exe]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    return exe


@pytest.mark.parametrize("name", EXAMPLES)
def test_examples_compile_against_the_c_abi(tmp_path, name):
    assert os.path.exists(_build(name, tmp_path))


@pytest.mark.gpu
@pytest.mark.parametrize("name", EXAMPLES)
def test_examples_run(tmp_path, name):
    exe = _build(name, tmp_path)
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "FAILED" not in r.stdout
