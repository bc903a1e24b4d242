#!/bin/bash
# round 5: A/B of the opaque-tid (zero-scratch) build against the same sources with -DECFFT_CORE_OPAQUE_TID=0, then the parity suite
O=gpurun_out/r05ab; rm -rf $O; mkdir -p $O
for ln in 20 18 16; do echo "== secp256k1 2^$ln" >> $O/ab.txt; python tools/ab_many.py secp256k1 $ln ecfft_amd/variants/no_opaque_tid.so ecfft_amd/libecfft_hip.so >> $O/ab.txt 2>&1; done
cat $O/ab.txt
(time timeout 1400 python -m pytest tests/test_gpu_parity.py tests/test_gpu_algorithms.py -m gpu -x -q) > $O/parity.log 2>&1; tail -6 $O/parity.log
