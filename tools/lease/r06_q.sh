#!/bin/bash
# round 6, session 4: re-sweep of the block / tile / occupancy knobs on the round-6 kernels (the round-2 sweep predates the matrix-core phases)
O=gpurun_out/r06q; rm -rf $O; mkdir -p $O
V=ecfft_amd/variants
LIBS="ecfft_amd/libecfft_hip.so $V/row256.so $V/lds256.so $V/tile14.so $V/tile16.so $V/mw2.so $V/mw3.so"
{
echo "## secp256k1 2^20"; python tools/ab_many.py secp256k1 20 $LIBS
echo "## secp256k1 2^20 x 8"; python tools/ab_many.py secp256k1 20 --count 8 $LIBS
echo "## secp256k1 2^18"; python tools/ab_many.py secp256k1 18 $LIBS
} 2>&1 | grep -v amdgpu.ids > $O/knob_resweep_ab.txt
cat $O/knob_resweep_ab.txt
