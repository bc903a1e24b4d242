#!/bin/bash
# round 6, lease M: the committed sources (head) against the data_st form with the flag never set (nts0) and set for launches <= 2^18 (nts18) at the
# sizes lease L did not cover
O=gpurun_out/r06m; rm -rf $O; mkdir -p $O
V=ecfft_amd/variants
{
for ln in 20 21 22 14 12; do echo "== secp256k1 2^$ln"; python tools/ab_many.py secp256k1 $ln $V/head.so $V/nts0.so $V/nts18.so 2>&1 | tail -3; done
echo "== secp256k1 2^20 x 8 (batched)"; python tools/ab_many.py secp256k1 20 --count 8 $V/head.so $V/nts0.so $V/nts18.so 2>&1 | tail -3
echo "== m31 2^24"; python tools/ab_many.py m31 24 $V/head.so $V/nts0.so 2>&1 | tail -2
} > $O/nt_store_ab3.txt 2>&1
cat $O/nt_store_ab3.txt
