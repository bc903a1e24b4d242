#!/bin/bash
# round 4: the small-launch matrix-core path: parity (full GPU suite) + latency table with / without it
O=gpurun_out/r04small; rm -rf $O; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log; tail -5 $O/pytest.log
python tools/small_sizes.py secp256k1 > $O/small_sizes.txt 2>&1; cat $O/small_sizes.txt
ECFFT_NO_MFMA=1 python tools/small_sizes.py secp256k1 > $O/small_sizes_no_mfma.txt 2>&1; cat $O/small_sizes_no_mfma.txt
ECFFT_NO_MFMA=1 ECFFT_NO_ROW256=1 python tools/small_sizes.py secp256k1 > $O/small_sizes_no_mfma_no_row256.txt 2>&1; cat $O/small_sizes_no_mfma_no_row256.txt
ECFFT_NO_LOW16=1 SIZES=16,17 python tools/small_sizes.py secp256k1 > $O/small_sizes_no_low16.txt 2>&1; cat $O/small_sizes_no_low16.txt
