#!/usr/bin/env python3
"""A/B of an environment switch that is read when a context is built: two contexts in one process, outputs compared bit for bit,
ENTER / EXIT timed interleaved.   usage: ab_env.py FIELD VAR=VALUE LOG_N [LOG_N ...]"""
import os, sys, time, statistics
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, ecfft_amd
ecfft_amd.fftree.use_hooks_library().__enter__()      # A/B switches are read by the hooks build only (tests/hooks)
from bench import synth
field, kv, sizes = sys.argv[1], sys.argv[2], [int(a) for a in sys.argv[3:]]
var, val = kv.split("=")
F = ecfft_amd.FIELDS[field]
for ln in sizes:
    n = 1 << ln
    os.environ.pop(var, None)
    t0 = F.build_fftree(n)
    os.environ[var] = val
    t1 = F.build_fftree(n)
    os.environ.pop(var, None)
    h = synth(field, n, 3)
    x = torch.from_numpy(h.view(np.int64) if field == "secp256k1" else h.view(np.int32)).cuda()
    same = bool(torch.equal(t0.enter(x), t1.enter(x))) and bool(torch.equal(t0.exit(x), t1.exit(x))) and bool(torch.equal(t1.exit(t1.enter(x)), x))
    hx = x[: n // 2].contiguous()
    same = same and bool(torch.equal(t0.extend(hx, ecfft_amd.Moiety.S1), t1.extend(hx, ecfft_amd.Moiety.S1))) and bool(torch.equal(t0.extend(hx, ecfft_amd.Moiety.S0), t1.extend(hx, ecfft_amd.Moiety.S0)))
    res = {0: [], 1: []}
    for r in range(17):
        for k, t in ((0, t0), (1, t1)):
            torch.cuda.synchronize(); a = time.perf_counter()
            for _ in range(4): t.exit(t.enter(x))
            torch.cuda.synchronize()
            if r >= 2: res[k].append((time.perf_counter() - a) / 4 * 1e3)
    m0, m1 = statistics.median(res[0]), statistics.median(res[1])
    print(f"{field} 2^{ln}: default {m0:.4f} ms   {kv} {m1:.4f} ms  ({(m1 / m0 - 1) * 100:+.1f} %)   bit-identical {same}")
    del t0, t1
