#!/bin/bash
# round 6, lease F: batched EXTEND (ecfft_extend, count > 1) as two half-batches on two streams: interleaved A/B against the one-stream
# build (bs0), then the parity tests that cover it
O=gpurun_out/r06f2; rm -rf $O; mkdir -p $O
V=ecfft_amd/variants
{
echo "== batched EXTEND S0 -> S1 -> S0, ms per vector pair: bs0 = one stream, libecfft_hip = two half-batches on two streams"
for cfg in "19 2" "18 8" "16 32" "20 2" "22 2"; do set -- $cfg; echo "-- secp256k1 e = 2^$1 x $2"; python tools/ab_many.py secp256k1 $1 --count $2 --extend $V/bs0.so ecfft_amd/libecfft_hip.so 2>&1 | tail -2; done
echo "-- m31 e = 2^22 x 4"; python tools/ab_many.py m31 22 --count 4 --extend $V/bs0.so ecfft_amd/libecfft_hip.so 2>&1 | tail -2
} > $O/extend_split_ab.txt 2>&1
cat $O/extend_split_ab.txt
(time timeout 1200 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "batched or headline or golden or matches_oracle") > $O/parity.log 2>&1; tail -4 $O/parity.log
