import time, sys
sys.path.insert(0, '.')
import torch, ecfft_amd
F = ecfft_amd.FIELDS["secp256k1"]
for i in range(3):
    t0 = time.perf_counter(); t = F.build_fftree(1 << 20); torch.cuda.synchronize(); print("build", time.perf_counter() - t0); del t
