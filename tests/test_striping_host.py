"""CPU test of the link striping of the sharded transforms' exchanges (Transport::exchange_striped, ecfft_amd/csrc/transport.h): W ranks
as host threads over a transport that moves host bytes with grouped-exchange semantics (tests/cpp/striping_host.cpp) — world 4 and 8,
the message patterns of the split ENTER / EXIT and random ones, striped and not: every receive buffer must hold what the plain
exchange delivers, and the number of grouped exchanges issued must be the expected one (2 when striped, 1 otherwise)."""
import os
import subprocess

from conftest import ROOT


def test_striped_exchange_delivers_what_the_plain_exchange_delivers(tmp_path):
    exe = str(tmp_path / "striping_host")
    src = os.path.join(ROOT, "tests", "cpp", "striping_host.cpp")
    subprocess.run(["g++", "-O1", "-std=c++17", "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include", src, "-L/opt/rocm/lib", "-lamdhip64", "-lpthread",
                    "-Wl,-rpath,/opt/rocm/lib", "-o", exe], check=True, capture_output=True)
    r = subprocess.run([exe], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "STRIPING_HOST_OK" in r.stdout, r.stdout[-3000:] + r.stderr[-2000:]
    assert "FAIL" not in r.stdout
