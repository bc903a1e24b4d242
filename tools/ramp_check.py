import sys, time, numpy as np, torch
sys.path.insert(0, '.')
import ecfft_amd
from bench import synth
F = ecfft_amd.FIELDS["secp256k1"]; n = 1 << 20
t = F.build_fftree(n)
x = torch.from_numpy(synth("secp256k1", n, 3).view(np.int64)).cuda()
torch.cuda.synchronize()
ts = []
for i in range(40):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    y = t.exit(t.enter(x))
    torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
print(" ".join("%.2f" % v for v in ts))
time.sleep(2.0)
ts = []
for i in range(10):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    y = t.exit(t.enter(x))
    torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
print("after 2 s idle:", " ".join("%.2f" % v for v in ts))
