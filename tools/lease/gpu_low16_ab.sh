#!/bin/bash
# the four lowest ENTER / EXIT levels as one 16 x 16 matrix-core map: parity, then bench A/B (ECFFT_NO_LOW16=1 = VALU sweeps)
python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "matrix_core or oracle or golden or roundtrip or round_trip" 2>&1 | tail -5
python __graft_entry__.py smoke 2>&1 | tail -4
for i in 1 2; do
python bench.py --steps 10 --warmup 3 --cpu-log-n 0 --batch 0 2>/dev/null | python tools/bench_classes.py
ECFFT_NO_LOW16=1 python bench.py --steps 10 --warmup 3 --cpu-log-n 0 --batch 0 2>/dev/null | python tools/bench_classes.py
done
