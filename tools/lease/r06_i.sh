#!/bin/bash
# round 6, lease I: the tile DATA of the fused passes with non-temporal loads (nt1), stores (nt2) or both (nt3) against the shipped build
O=gpurun_out/r06i; rm -rf $O; mkdir -p $O
V=ecfft_amd/variants
{
for ln in 20 18 22; do echo "== secp256k1 2^$ln"; python tools/ab_many.py secp256k1 $ln ecfft_amd/libecfft_hip.so $V/nt1.so $V/nt2.so $V/nt3.so 2>&1 | tail -4; done
echo "== secp256k1 2^20 x 8 (batched)"; python tools/ab_many.py secp256k1 20 --count 8 ecfft_amd/libecfft_hip.so $V/nt1.so $V/nt2.so $V/nt3.so 2>&1 | tail -4
} > $O/nt_data_ab.txt 2>&1
cat $O/nt_data_ab.txt
