"""is the host the bottleneck?  time for the (asynchronous) ENTER / EXIT call to RETURN vs the time until the GPU is done"""
import sys, os, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, ecfft_amd
from bench import synth
log_n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
n = 1 << log_n
F = ecfft_amd.FIELDS["secp256k1"]
t = F.build_fftree(n)
x = torch.from_numpy(synth("secp256k1", n, 3).view(np.int64)).cuda()
for op in ("enter", "exit"):
    f = getattr(t, op)
    for _ in range(3): y = f(x)
    torch.cuda.synchronize()
    ret, tot = [], []
    for _ in range(20):
        torch.cuda.synchronize(); t0 = time.perf_counter(); y = f(x); t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
        ret.append((t1 - t0) * 1e3); tot.append((t2 - t0) * 1e3)
    print(f"secp256k1 2^{log_n} {op}: call returns after {sorted(ret)[10]:.3f} ms, GPU done after {sorted(tot)[10]:.3f} ms")
# back-to-back: 10 ENTER+EXIT pairs enqueued without waiting
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(10):
    y = t.exit(t.enter(x))
t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
print(f"10 pairs: enqueued in {(t1 - t0) * 100:.3f} ms per pair, done in {(t2 - t0) * 100:.3f} ms per pair")
