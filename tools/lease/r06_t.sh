#!/bin/bash
# round 6, session 4: the largest sizes (secp256k1 2^24 = 64 GiB-class tables, M31 2^27) and a wider soak on the final library
O=gpurun_out/r06t; rm -rf $O; mkdir -p $O
timeout 1200 python tools/big_sizes_check.py secp256k1:22 secp256k1:24 m31:25 m31:27 > $O/big_sizes.txt 2>&1; echo "rc=$?" >> $O/big_sizes.txt
timeout 900 python tools/soak_gpu.py 16 > $O/soak16.txt 2>&1; echo "rc=$?" >> $O/soak16.txt
timeout 400 python tools/soak_mfma.py 240 > $O/soak_mfma.txt 2>&1; echo "rc=$?" >> $O/soak_mfma.txt
cat $O/big_sizes.txt $O/soak16.txt $O/soak_mfma.txt | grep -v amdgpu.ids
