import time, sys, os
sys.path.insert(0, '.')
import torch, ecfft_amd
torch.cuda.init(); torch.zeros(1, device="cuda"); torch.cuda.synchronize()
F = ecfft_amd.FIELDS["secp256k1"]
L = ecfft_amd.lib()
if os.environ.get("PREWARM"):
    t0 = time.perf_counter(); tt = F.build_fftree(2); torch.cuda.synchronize(); print("prewarm build(2)", time.perf_counter() - t0)
for i in range(3):
    t0 = time.perf_counter(); t = F.build_fftree(1 << 20); torch.cuda.synchronize(); print("build", time.perf_counter() - t0); del t
