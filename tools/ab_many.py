#!/usr/bin/env python3
"""Interleaved A/B of several builds of libecfft_hip.so on ONE box: ENTER+EXIT at n = 2^log_n through raw ctypes calls, median of 7
timed blocks of 10 pairs per library, round-robin so that clock drift hits every build alike.
usage: ab_many.py field log_n [--count C] [--extend] LIB [LIB ...]      --count C: batches of C polynomials (ecfft_enter_many / ecfft_exit_many);
the time printed is per polynomial pair.  --extend: the timed pair is EXTEND S0 -> S1 followed by EXTEND S1 -> S0 of C vectors of
2^log_n evaluations (ecfft_extend on T_2^(log_n + 1)); checked by the round trip"""
import ctypes, sys, time, statistics, os
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import synth
field, log_n, libs = sys.argv[1], int(sys.argv[2]), sys.argv[3:]
count = 1
if libs and libs[0] == "--count":
    count, libs = int(libs[1]), libs[2:]
ext = False
if libs and libs[0] == "--extend":
    ext, libs = True, libs[1:]
fid = 0 if field == "secp256k1" else 1
n = 1 << log_n
h = np.concatenate([synth(field, n, 3 + i) for i in range(count)])
x = torch.from_numpy(h.view(np.int64) if fid == 0 else h.view(np.int32)).cuda()
ev = torch.empty_like(x); back = torch.empty_like(x)
res = {}
def block(L, c):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10):
        if ext:
            assert L.ecfft_extend(c, x.data_ptr(), ev.data_ptr(), n, 1, count, 1, None) == 0
            assert L.ecfft_extend(c, ev.data_ptr(), back.data_ptr(), n, 0, count, 1, None) == 0
        elif count == 1:
            assert L.ecfft_enter(c, x.data_ptr(), ev.data_ptr(), n, 1, None) == 0
            assert L.ecfft_exit(c, ev.data_ptr(), back.data_ptr(), n, 1, None) == 0
        else:
            assert L.ecfft_enter_many(c, x.data_ptr(), ev.data_ptr(), n, count, 1, None) == 0
            assert L.ecfft_exit_many(c, ev.data_ptr(), back.data_ptr(), n, count, 1, None) == 0
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / 10 / count * 1e3
ctx = []
for p in libs:
    L = ctypes.CDLL(p)
    L.ecfft_build_fftree.argtypes = [ctypes.c_int, ctypes.c_size_t, ctypes.c_int, ctypes.POINTER(ctypes.c_void_p)]
    for f in (L.ecfft_enter, L.ecfft_exit):
        f.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int, ctypes.c_void_p]
    for f in (L.ecfft_enter_many, L.ecfft_exit_many):
        f.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_size_t, ctypes.c_int, ctypes.c_void_p]
    L.ecfft_extend.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int, ctypes.c_size_t, ctypes.c_int, ctypes.c_void_p]
    L.ecfft_ctx_destroy.argtypes = [ctypes.c_void_p]
    c = ctypes.c_void_p()
    assert L.ecfft_build_fftree(fid, 2 * n if ext else n, 0, ctypes.byref(c)) == 0
    block(L, c); assert torch.equal(back, x), p
    ctx.append((p, L, c)); res[p] = []
for r in range(7):
    for p, L, c in ctx:
        res[p].append(block(L, c))
base = statistics.median(res[libs[0]])
for p in libs:
    m = statistics.median(res[p])
    print(f"{os.path.basename(p):24s} median {m:.3f} ms  min {min(res[p]):.3f}  ({(m / base - 1) * 100:+.1f} %)")
