#!/usr/bin/env python3
"""Randomised soak of the matrix-core paths (run on the GPU box): the library with and without ECFFT_NO_MFMA (all-VALU) must agree
bit for bit on ENTER / EXIT / EXTEND for single and batched shapes whose launches reach 2^18 elements (1024-element tiles: the
16-point maps of the row and low-level kernels and the low16 maps), over many seeds and on extreme operands.
usage: soak_mfma.py [seconds]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import ecfft_amd
ecfft_amd.fftree.use_hooks_library().__enter__()      # ECFFT_NO_MFMA is read by the hooks build only (tests/hooks)

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
P = ecfft_amd.FIELDS["secp256k1"]
shapes = [(1 << 18, 1), (1 << 10, 256), (1 << 19, 1), (1 << 12, 64), (1 << 16, 4), (1 << 20, 1), (1 << 14, 32), (1 << 11, 256)]
PM1 = np.array([0xFFFFFFFEFFFFFC2E, 0xFFFFFFFFFFFFFFFF, 0xFFFFFFFFFFFFFFFF, 0xFFFFFFFFFFFFFFFF], dtype=np.uint64)
bad = 0; checks = 0
t_end = time.time() + budget
trees = {}
for n, _ in shapes:
    if n not in trees:
        a = P.build_fftree(n)
        os.environ["ECFFT_NO_MFMA"] = "1"
        b = P.build_fftree(n)
        del os.environ["ECFFT_NO_MFMA"]
        trees[n] = (a, b)
seed = 0
while time.time() < t_end:
    for n, cnt in shapes:
        a, b = trees[n]
        rng = np.random.default_rng(1000 + seed)
        tot = n * cnt
        kind = seed % 4
        if kind == 3:
            x = np.tile(PM1, (tot, 1)); x[rng.integers(0, tot, tot // 7)] = 0
        else:
            x = rng.integers(0, 2**64, size=(tot, 4), dtype=np.uint64); x[:, 3] >>= np.uint64(1)
            if kind == 2:
                x[:, rng.integers(0, 4)] = 0xFFFFFFFFFFFFFFFF if rng.integers(0, 2) else 0x8080808080808080
                x[:, 3] >>= np.uint64(1)
        xt = torch.from_numpy(x.view(np.int64)).cuda()
        ea, eb = a.enter(xt, count=cnt), b.enter(xt, count=cnt)
        xa, xb = a.exit(xt, count=cnt), b.exit(xt, count=cnt)
        ok = torch.equal(ea, eb) and torch.equal(xa, xb) and torch.equal(a.exit(ea, count=cnt), xt)
        if cnt == 1:
            h = xt[: n // 2]
            ok = ok and torch.equal(a.extend(h, ecfft_amd.Moiety.S1), b.extend(h, ecfft_amd.Moiety.S1)) and torch.equal(a.extend(h, ecfft_amd.Moiety.S0), b.extend(h, ecfft_amd.Moiety.S0))
        checks += 1
        if not ok:
            bad += 1; print("MISMATCH n", n, "count", cnt, "seed", seed, "kind", kind, flush=True)
    seed += 1
print(f"soak_mfma: {checks} shape x seed checks in {budget:.0f} s, {bad} mismatches")
sys.exit(1 if bad else 0)
