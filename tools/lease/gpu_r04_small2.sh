#!/bin/bash
O=gpurun_out/r04small2; rm -rf $O; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log; tail -3 $O/pytest.log
python tools/small_sizes.py secp256k1 2>&1 | grep secp | tee $O/small_sizes.txt
echo NO_COL256; ECFFT_NO_COL256=1 SIZES=12,16,17 python tools/small_sizes.py secp256k1 2>&1 | grep secp | tee $O/small_sizes_no_col256.txt
echo MIN_LOGC=1; ECFFT_SMALL_MIN_LOGC=1 SIZES=12,16,17 python tools/small_sizes.py secp256k1 2>&1 | grep secp | tee $O/small_sizes_minlogc1.txt
