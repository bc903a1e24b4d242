#!/bin/bash
# round 6, lease L: where did the -2 % of the compile-time non-temporal-store build (nt2) come from?  head = the committed sources (no data_st at
# all), nts0 = data_st with the flag never set, libecfft_hip = flag for launches <= 2^19, nt2 = every result store non-temporal at compile time
O=gpurun_out/r06l; rm -rf $O; mkdir -p $O
V=ecfft_amd/variants
{
for ln in 18 19 17 16; do echo "== secp256k1 2^$ln"; python tools/ab_many.py secp256k1 $ln $V/head.so $V/nts0.so ecfft_amd/libecfft_hip.so $V/nt2.so $V/nt3.so 2>&1 | tail -5; done
} > $O/nt_store_ab2.txt 2>&1
cat $O/nt_store_ab2.txt
