#!/bin/bash
# round 5: nine column stages in one pass (rows of 2 elements) — A/B against the round-4 grouping, then the parity suite
O=gpurun_out/r05col9; rm -rf $O; mkdir -p $O
for ln in 20 21 19 22; do echo "== secp256k1 2^$ln" >> $O/ab.txt; python tools/ab_many.py secp256k1 $ln ecfft_amd/variants/round4_col8.so ecfft_amd/libecfft_hip.so 2>&1 | grep -v amdgpu >> $O/ab.txt; done
cat $O/ab.txt
(time timeout 1400 python -m pytest tests/test_gpu_parity.py tests/test_distributed.py -m gpu -x -q) > $O/parity.log 2>&1; tail -6 $O/parity.log
