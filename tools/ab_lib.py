#!/usr/bin/env python3
"""A/B of two builds of libecfft_hip.so on ONE box, interleaved: ENTER+EXIT at n = 2^log_n through raw ctypes calls that both
libraries export (build_fftree / enter / exit), median of `reps` timed blocks of 10 pairs each.
usage: ab_lib.py LIB_A LIB_B [field] [log_n]"""
import ctypes, sys, time, statistics
import numpy as np
import torch
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from bench import synth
libs = sys.argv[1:3]
field = sys.argv[3] if len(sys.argv) > 3 else "secp256k1"
log_n = int(sys.argv[4]) if len(sys.argv) > 4 else 20
fid = 0 if field == "secp256k1" else 1
n = 1 << log_n
h = synth(field, n, 3)
x = torch.from_numpy(h.view(np.int64) if fid == 0 else h.view(np.int32)).cuda()
ev = torch.empty_like(x); back = torch.empty_like(x)
ctx = []
for p in libs:
    L = ctypes.CDLL(p)
    L.ecfft_build_fftree.argtypes = [ctypes.c_int, ctypes.c_size_t, ctypes.c_int, ctypes.POINTER(ctypes.c_void_p)]
    for f in (L.ecfft_enter, L.ecfft_exit):
        f.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int, ctypes.c_void_p]
    c = ctypes.c_void_p()
    assert L.ecfft_build_fftree(fid, n, 0, ctypes.byref(c)) == 0
    ctx.append((L, c))
def block(L, c):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10):
        assert L.ecfft_enter(c, x.data_ptr(), ev.data_ptr(), n, 1, None) == 0
        assert L.ecfft_exit(c, ev.data_ptr(), back.data_ptr(), n, 1, None) == 0
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / 10 * 1e3
for L, c in ctx:
    block(L, c); assert torch.equal(back, x)
res = [[], []]
for r in range(9):
    for i, (L, c) in enumerate(ctx):
        res[i].append(block(L, c))
for p, r in zip(libs, res):
    print(f"{p}: median {statistics.median(r):.3f} ms  min {min(r):.3f}  max {max(r):.3f}")
