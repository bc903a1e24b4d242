#!/bin/bash
# round 6, lease G: the two-halves schedule in the LATENCY regime — single transforms of 2^15 .. 2^18 as two concurrent halves on two streams
# (-DECFFT_SPLIT_MIN_LOG=18 / 17 / 16 / 15 against the shipped 19), interleaved
O=gpurun_out/r06g; rm -rf $O; mkdir -p $O
V=ecfft_amd/variants
{
for ln in 16 17 18 15 14; do echo "== secp256k1 2^$ln (shipped: concurrent halves from 2^19; smNN: from 2^NN)"; python tools/ab_many.py secp256k1 $ln ecfft_amd/libecfft_hip.so $V/sm18.so $V/sm17.so $V/sm16.so $V/sm15.so 2>&1 | tail -5; done
echo "== secp256k1 2^20"; python tools/ab_many.py secp256k1 20 ecfft_amd/libecfft_hip.so $V/sm18.so $V/sm16.so 2>&1 | tail -3
} > $O/split_min_log_ab.txt 2>&1
cat $O/split_min_log_ab.txt
