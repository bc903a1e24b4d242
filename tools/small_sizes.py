#!/usr/bin/env python3
"""ENTER+EXIT wall time of ONE transform at small sizes (the latency regime), device-resident data: median of `reps` calls.
usage: small_sizes.py [field] ; env ECFFT_NO_SMALL_TILES=1 switches the small-launch tile rule off for an A/B."""
import sys, time, os
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import ecfft_amd
from bench import synth
field = sys.argv[1] if len(sys.argv) > 1 else "secp256k1"
F = ecfft_amd.FIELDS[field]
for ln in [int(a) for a in os.environ.get("SIZES", "8,10,11,12,14,16,17,18").split(",")]:
    n = 1 << ln
    t = F.build_fftree(n)
    h = synth(field, n, 1)
    x = torch.from_numpy(h.view(np.int64) if field == "secp256k1" else h.view(np.int32)).cuda()
    for _ in range(3):
        y = t.exit(t.enter(x))
    assert torch.equal(y, x)
    ts = []
    for _ in range(15):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        ev = t.enter(x); torch.cuda.synchronize(); t1 = time.perf_counter()
        y = t.exit(ev); torch.cuda.synchronize(); t2 = time.perf_counter()
        ts.append((t1 - t0, t2 - t1))
    e = sorted(a for a, _ in ts)[len(ts) // 2] * 1e3; x_ = sorted(b for _, b in ts)[len(ts) // 2] * 1e3
    print(f"{field} n=2^{ln}: ENTER {e:.3f} ms  EXIT {x_:.3f} ms  sum {e + x_:.3f} ms")
