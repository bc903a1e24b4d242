#!/bin/bash
# round 6, lease K: non-temporal result stores as a PER-LAUNCH flag (launches of <= 2^K elements): K = 0 (never: rounds 1-5), 17, 18, 19 (shipped), 20
O=gpurun_out/r06k; rm -rf $O; mkdir -p $O
V=ecfft_amd/variants
{
for ln in 14 16 17 18 19 20 21 22; do echo "== secp256k1 2^$ln  (nts0 = never, ntsNN = launches of <= 2^NN elements, libecfft_hip = 2^19)"; python tools/ab_many.py secp256k1 $ln $V/nts0.so $V/nts17.so $V/nts18.so ecfft_amd/libecfft_hip.so $V/nts20.so 2>&1 | tail -5; done
echo "== secp256k1 2^20 x 8 (batched)"; python tools/ab_many.py secp256k1 20 --count 8 $V/nts0.so $V/nts18.so ecfft_amd/libecfft_hip.so $V/nts20.so 2>&1 | tail -4
} > $O/nt_store_ab.txt 2>&1
cat $O/nt_store_ab.txt
ECFFT_LIB=$PWD/ecfft_amd/libecfft_hip.so timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "golden or matches_oracle or config2 or 2e18" > $O/parity.log 2>&1; tail -3 $O/parity.log
