#!/bin/bash
# round 6, session 4: the round-4 bounding experiment (no tile loads / stores; stage constants made up in registers; results garbage by design) repeated
# for the STEADY STATE (8 polynomials per call) — what could perfect tile prefetch / free constants buy where launches are deep?
O=gpurun_out/r06u; rm -rf $O; mkdir -p $O
V=ecfft_amd/variants
LIBS="ecfft_amd/libecfft_hip.so $V/notio.so $V/notab.so $V/noboth.so"
{
echo "## single transform"; python tools/bound_variants.py secp256k1 20 $LIBS
echo "## 8 polynomials per call (per polynomial)"; python tools/bound_variants.py secp256k1 20 --count 8 $LIBS
echo "## M31 2^24, 2 polynomials per call"; python tools/bound_variants.py m31 24 --count 2 $LIBS
} 2>&1 | grep -v amdgpu.ids > $O/bounds_batched.txt
cat $O/bounds_batched.txt
