#!/bin/bash
# round 6, lease J: non-temporal tile-data stores / loads+stores across the latency regime (2^12 .. 2^19)
O=gpurun_out/r06j; rm -rf $O; mkdir -p $O
V=ecfft_amd/variants
{
for ln in 12 14 16 17 18 19; do echo "== secp256k1 2^$ln"; python tools/ab_many.py secp256k1 $ln ecfft_amd/libecfft_hip.so $V/nt1.so $V/nt2.so $V/nt3.so 2>&1 | tail -4; done
echo "== m31 2^24"; python tools/ab_many.py m31 24 ecfft_amd/libecfft_hip.so $V/nt3.so 2>&1 | tail -2
} > $O/nt_data_ab2.txt 2>&1
cat $O/nt_data_ab2.txt
