#!/bin/bash
# round-2 measurement batch for DESIGN.md section 7: small sizes, criterion shape, EXTEND 2^22, big sizes
mkdir -p gpurun_out/fin
python tools/small_sizes.py secp256k1 > gpurun_out/fin/small_secp.txt 2>&1
SIZES=11,12,16,18,20,22 python tools/small_sizes.py m31 > gpurun_out/fin/small_m31.txt 2>&1
g++ -O2 -std=c++17 -Iinclude examples/bench_fftree.cpp -Lecfft_amd -lecfft_hip -Wl,-rpath,$PWD/ecfft_amd -Wl,--allow-shlib-undefined -o /tmp/bench_fftree && /tmp/bench_fftree > gpurun_out/fin/bench_fftree.txt 2>&1
python bench.py --mode extend-split --log-n 22 --steps 10 --warmup 2 2>/dev/null | grep "^{" > gpurun_out/fin/extend_split_2e22.json
python - > gpurun_out/fin/extend_2e22.txt 2>&1 <<'PY'
import time, numpy as np, torch, ecfft_amd
from bench import synth
F=ecfft_amd.secp256k1; e=1<<22; t=F.build_fftree(2*e)
x=torch.from_numpy(synth("secp256k1",e,1).view(np.int64)).cuda()
for _ in range(3): y=t.extend(x, 1)
torch.cuda.synchronize(); t0=time.perf_counter()
for _ in range(10): y=t.extend(x, 1)
torch.cuda.synchronize(); print("secp256k1 EXTEND e=2^22 single GPU: %.3f ms"%((time.perf_counter()-t0)/10*1e3))
PY
python tools/big_sizes_check.py > gpurun_out/fin/big_sizes.txt 2>&1
cat gpurun_out/fin/small_secp.txt gpurun_out/fin/small_m31.txt gpurun_out/fin/bench_fftree.txt gpurun_out/fin/extend_2e22.txt gpurun_out/fin/big_sizes.txt
