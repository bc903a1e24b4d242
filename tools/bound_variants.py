#!/usr/bin/env python3
"""Interleaved timing of ENTER and EXIT on several builds of the library WITHOUT checking results — for the bounding experiments
(-DECFFT_EXP_NO_TILE_IO: no tile loads / stores, -DECFFT_EXP_NO_TABLE_LOADS: stage constants made up in registers,
-DECFFT_EXP_NO_MID: no fused pointwise step in k_stages_col_mid), whose outputs are garbage by design.  They bound what perfect tile
prefetch / free constants could buy before anything is built.   usage: bound_variants.py FIELD LOG_N [--count C] LIB [LIB ...]
--count C: batches of C polynomials per call (ecfft_enter_many / ecfft_exit_many — the steady state of deep launches); times are per polynomial"""
import ctypes, sys, time, statistics, os
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import synth
field, log_n, libs = sys.argv[1], int(sys.argv[2]), sys.argv[3:]
count = 1
if libs and libs[0] == "--count":
    count, libs = int(libs[1]), libs[2:]
fid = 0 if field == "secp256k1" else 1
n = 1 << log_n
h = np.concatenate([synth(field, n, 3 + i) for i in range(count)])
x = torch.from_numpy(h.view(np.int64) if fid == 0 else h.view(np.int32)).cuda()
ev = torch.empty_like(x)
ctx = []
for p in libs:
    L = ctypes.CDLL(p)
    L.ecfft_build_fftree.argtypes = [ctypes.c_int, ctypes.c_size_t, ctypes.c_int, ctypes.POINTER(ctypes.c_void_p)]
    for f in (L.ecfft_enter, L.ecfft_exit):
        f.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int, ctypes.c_void_p]
    for f in (L.ecfft_enter_many, L.ecfft_exit_many):
        f.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_size_t, ctypes.c_int, ctypes.c_void_p]
    c = ctypes.c_void_p()
    assert L.ecfft_build_fftree(fid, n, 0, ctypes.byref(c)) == 0
    ctx.append((p, L, c))
res = {p: {"enter": [], "exit": []} for p in libs}
for r in range(9):
    for p, L, c in ctx:
        for op, fn, fnm in (("enter", L.ecfft_enter, L.ecfft_enter_many), ("exit", L.ecfft_exit, L.ecfft_exit_many)):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(5):
                if count == 1:
                    assert fn(c, x.data_ptr(), ev.data_ptr(), n, 1, None) == 0
                else:
                    assert fnm(c, x.data_ptr(), ev.data_ptr(), n, count, 1, None) == 0
            torch.cuda.synchronize()
            if r >= 2:
                res[p][op].append((time.perf_counter() - t0) / 5 / count * 1e3)
for p in libs:
    e, x_ = statistics.median(res[p]["enter"]), statistics.median(res[p]["exit"])
    print(f"{os.path.basename(p):22s} {field} 2^{log_n}" + (f" x {count}" if count > 1 else "") + f": ENTER {e:.3f} ms   EXIT {x_:.3f} ms   sum {e + x_:.3f} ms")
