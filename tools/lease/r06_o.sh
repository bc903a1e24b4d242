#!/bin/bash
# round 6, session 4: backend scheduler flags (never swept before): interleaved A/B of whole-library builds
O=gpurun_out/r06o; rm -rf $O; mkdir -p $O
V=ecfft_amd/variants
LIBS="ecfft_amd/libecfft_hip.so $V/cf_ilp.so $V/cf_memcl.so $V/cf_nopost.so $V/cf_trackers.so $V/cf_relaxed.so"
{
echo "## secp256k1 2^20"; python tools/ab_many.py secp256k1 20 $LIBS
echo "## secp256k1 2^20 x 8"; python tools/ab_many.py secp256k1 20 --count 8 $LIBS
echo "## secp256k1 2^16"; python tools/ab_many.py secp256k1 16 $LIBS
echo "## secp256k1 2^18"; python tools/ab_many.py secp256k1 18 $LIBS
echo "## m31 2^24"; python tools/ab_many.py m31 24 $LIBS
} > $O/sched_flags_ab.txt 2>&1
cat $O/sched_flags_ab.txt
