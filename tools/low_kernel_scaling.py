#!/usr/bin/env python3
"""k_exit_low / k_enter_low alone: `count` polynomials of 2^10 coefficients in ONE batched call = ONE launch of the low-level kernel with
`count` tiles (nothing else runs).  Time per launch against the number of tiles: how much do the two workgroups a CU holds overlap?
usage: low_kernel_scaling.py [field]      (hooks build: ECFFT_NO_MFMA / ECFFT_NO_LOW16 in the environment decompose it)"""
import os, sys, time, statistics
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, ecfft_amd
from bench import synth
if any(k in os.environ for k in ("ECFFT_NO_MFMA", "ECFFT_NO_LOW16")):
    ecfft_amd.fftree.use_hooks_library().__enter__()
field = sys.argv[1] if len(sys.argv) > 1 else "secp256k1"
F = ecfft_amd.FIELDS[field]
n = 1 << (10 if field == "secp256k1" else 13)
t = F.build_fftree(n)
for count in (64, 128, 256, 512, 768, 1024, 2048, 4096):
    h = synth(field, n * count, 3)
    x = torch.from_numpy(h.view(np.int64) if field == "secp256k1" else h.view(np.int32)).cuda()
    ev = t.enter(x, count=count); back = t.exit(ev, count=count); torch.cuda.synchronize()
    assert torch.equal(back, x)
    def med(fn):
        ts = []
        for _ in range(21):
            torch.cuda.synchronize(); t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e6)
        return statistics.median(ts)
    e, xx = med(lambda: t.enter(x, count=count)), med(lambda: t.exit(ev, count=count))
    print(f"{field} tiles {count:5d}: k_enter_low launch {e:8.1f} us   k_exit_low launch {xx:8.1f} us   (per 256 tiles: {e * 256 / count:7.1f} / {xx * 256 / count:7.1f})")
