#!/bin/bash
# round 6, session 4: odd batches take the two-stream schedule too (uneven parts): parity vs oracle + interleaved A/B against the even-only build
O=gpurun_out/r06r; rm -rf $O; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "batched" > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log; tail -4 $O/pytest.log
LIBS="ecfft_amd/variants/even_only.so ecfft_amd/libecfft_hip.so"
{
for c in 3 5 7; do echo "## secp256k1 2^20 x $c ENTER+EXIT"; python tools/ab_many.py secp256k1 20 --count $c $LIBS; done
echo "## secp256k1 2^19 x 3 ENTER+EXIT"; python tools/ab_many.py secp256k1 19 --count 3 $LIBS
echo "## secp256k1 2^20 x 8 ENTER+EXIT (unchanged path)"; python tools/ab_many.py secp256k1 20 --count 8 $LIBS
echo "## secp256k1 2^20 x 3 EXTEND pair"; python tools/ab_many.py secp256k1 20 --count 3 --extend $LIBS
echo "## m31 2^24 x 3 ENTER+EXIT"; python tools/ab_many.py m31 24 --count 3 $LIBS
} 2>&1 | grep -v amdgpu.ids > $O/odd_batch_ab.txt
cat $O/odd_batch_ab.txt
