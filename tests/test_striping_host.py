"""CPU test of the link striping of the sharded transforms' exchanges (Transport::exchange_striped, ecfft_amd/csrc/transport.h): W ranks
as host threads over a transport that moves host bytes with grouped-exchange semantics (tests/cpp/striping_host.cpp) — world 4 and 8,
the message patterns of the split ENTER / EXIT and random ones, striped and not: every receive buffer must hold what the plain
exchange delivers, and the number of grouped exchanges issued must be the expected one (2 when striped, 1 otherwise).
Round 6: the planner takes combinatorial input, so the driver is built with AddressSanitizer + UBSan and also
runs seeded random patterns at W = 4, 8, 16, 64 (zero-length and non-divisible messages, self messages, idle ranks) and the error
path (a pattern that disagrees with the call fails on that rank instead of falling back to a plain exchange)."""
import os
import subprocess

from conftest import ROOT


def test_striped_exchange_delivers_what_the_plain_exchange_delivers(tmp_path):
    exe = str(tmp_path / "striping_host")
    src = os.path.join(ROOT, "tests", "cpp", "striping_host.cpp")
    subprocess.run(["g++", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined", "-std=c++17", "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include", src, "-L/opt/rocm/lib", "-lamdhip64", "-lpthread",
                    "-Wl,-rpath,/opt/rocm/lib", "-o", exe], check=True, capture_output=True)
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=0:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1")     # (libamdhip64 is linked for its symbols, never called)
    r = subprocess.run([exe], capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0 and "STRIPING_HOST_OK" in r.stdout, r.stdout[-3000:] + r.stderr[-2000:]
    assert "FAIL" not in r.stdout
    assert "ERROR: AddressSanitizer" not in r.stderr and "runtime error" not in r.stderr, r.stderr[-3000:]
    assert "random patterns at W = 64" in r.stdout and "inconsistent patterns" in r.stdout
