#!/usr/bin/env python3
"""Is an ENTER+EXIT faster when replayed from a HIP graph?  Captures t.exit(t.enter(x)) on device tensors with torch.cuda.graph
(the library enqueues on torch's current stream — the capture stream — and forks / joins its side streams with events) and
times replay against eager calls, interleaved.  usage: graph_check.py [log_n ...]"""
import sys, os, time, statistics
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, ecfft_amd
from bench import synth
F = ecfft_amd.FIELDS["secp256k1"]
for ln in [int(a) for a in (sys.argv[1:] or ["12", "16", "17", "20"])]:
    n = 1 << ln
    t = F.build_fftree(n)
    x = torch.from_numpy(synth("secp256k1", n, 3).view(np.int64)).cuda()
    for _ in range(3):
        y = t.exit(t.enter(x))
    torch.cuda.synchronize()
    assert torch.equal(y, x)
    try:
        g = torch.cuda.CUDAGraph()
        s = torch.cuda.Stream()
        with torch.cuda.stream(s):
            with torch.cuda.graph(g, stream=s):
                yg = t.exit(t.enter(x))
        torch.cuda.synchronize()
        g.replay(); torch.cuda.synchronize()
        ok = bool(torch.equal(yg, x))
    except Exception as ex:
        print(f"2^{ln}: capture failed: {type(ex).__name__}: {str(ex)[:200]}"); continue
    def eager():
        torch.cuda.synchronize(); t0 = time.perf_counter(); t.exit(t.enter(x)); torch.cuda.synchronize(); return (time.perf_counter() - t0) * 1e3
    def replay():
        torch.cuda.synchronize(); t0 = time.perf_counter(); g.replay(); torch.cuda.synchronize(); return (time.perf_counter() - t0) * 1e3
    e, r = [], []
    for _ in range(15):
        e.append(eager()); r.append(replay())
    print(f"secp256k1 2^{ln}: eager {statistics.median(e):.3f} ms   graph replay {statistics.median(r):.3f} ms   replayed result correct: {ok}")
