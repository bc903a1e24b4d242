#!/bin/bash
# round 3: MFMA micro-benchmark + bench A/B (matrix cores on / off)
mkdir -p gpurun_out/r03
(cd tools/ubench && timeout 300 ./mfma_mul 2048 50) 2>&1 | tee gpurun_out/r03/mfma_mul_v2.txt
python bench.py --steps 10 --warmup 3 --cpu-log-n 0 2>/dev/null | python tools/bench_classes.py
ECFFT_NO_MFMA=1 python bench.py --steps 10 --warmup 3 --cpu-log-n 0 2>/dev/null | python tools/bench_classes.py
