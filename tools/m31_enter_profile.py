"""per-class launch count / event time of ONE M31 ENTER at 2^24 (and of EXIT), from the library's per-launch profiler"""
import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, ecfft_amd, time
from bench import synth
log_n = int(sys.argv[1]) if len(sys.argv) > 1 else 24
field = sys.argv[2] if len(sys.argv) > 2 else "m31"
n = 1 << log_n
F = ecfft_amd.FIELDS[field]
t = F.build_fftree(n)
h = synth(field, n, 3)
x = torch.from_numpy(h.view(np.int32) if field == "m31" else h.view(np.int64)).cuda()
for op in ("enter", "exit"):
    f = getattr(t, op)
    for _ in range(3): y = f(x)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10): y = f(x)
    torch.cuda.synchronize(); wall = (time.perf_counter() - t0) / 10 * 1e3
    t.profile(True)
    for _ in range(5): y = f(x)
    torch.cuda.synchronize()
    cl = t.profile_read(); t.profile(False)
    print(f"{field} 2^{log_n} {op}: {wall:.3f} ms wall")
    for c in cl:
        if c["launches"]:
            print(f"   {c['name']:18s} {c['launches'] / 5:6.1f} launches  {c['ms'] / 5:7.3f} ms event  {c['ms'] / c['launches'] * 1e3:7.1f} us each  alg {c['alg_bytes'] / c['launches'] / 1e6:8.1f} MB/launch")
