#!/bin/bash
O=gpurun_out/r04low32; rm -rf $O; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_field.py tests/test_gpu_parity.py -m gpu -x -q -k "blk32 or low16 or matrix_core_path or 2e18 or full_size" > $O/pytest.log 2>&1; echo "rc=$?"; grep -E "passed|failed" $O/pytest.log | tail -2; grep -n "^E  " $O/pytest.log | head
for m in 3 2 1 0; do echo "LOW32=$m"; ECFFT_LOW32=$m python bench.py --cpu-log-n 0 2>/dev/null | python tools/bench_classes.py; done
