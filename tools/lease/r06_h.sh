#!/bin/bash
# round 6, lease H: a batch as 2 (shipped) / 4 / 8 concurrent parts (-DECFFT_BATCH_WAYS), interleaved A/B of the batched ENTER+EXIT and EXTEND
O=gpurun_out/r06h; rm -rf $O; mkdir -p $O
V=ecfft_amd/variants
{
echo "== batched ENTER+EXIT, ms per polynomial pair: libecfft_hip = 2 parts, bw4 = up to 4, bw8 = up to 8 (a part has >= 2^19 elements)"
for cfg in "20 8" "20 4" "19 8" "18 16" "19 16" "20 16"; do set -- $cfg; echo "-- secp256k1 2^$1 x $2"; python tools/ab_many.py secp256k1 $1 --count $2 ecfft_amd/libecfft_hip.so $V/bw4.so $V/bw8.so 2>&1 | tail -3; done
echo "-- m31 2^24 x 4"; python tools/ab_many.py m31 24 --count 4 ecfft_amd/libecfft_hip.so $V/bw4.so 2>&1 | tail -2
echo "== batched EXTEND, ms per vector pair"
for cfg in "20 4" "19 8" "22 4"; do set -- $cfg; echo "-- secp256k1 e = 2^$1 x $2"; python tools/ab_many.py secp256k1 $1 --count $2 --extend ecfft_amd/libecfft_hip.so $V/bw4.so $V/bw8.so 2>&1 | tail -3; done
} > $O/batch_ways_ab.txt 2>&1
cat $O/batch_ways_ab.txt
