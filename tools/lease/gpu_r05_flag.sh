#!/bin/bash
# round 5, first call: the flag-chained pass micro-benchmark + same-box latency-regime baseline
O=gpurun_out/r05flag; rm -rf $O; mkdir -p $O
for G in 256 128 512; do timeout 120 tools/ubench/flag_wait $G 200 >> $O/flag_wait.txt 2>&1; echo "rc=$?" >> $O/flag_wait.txt; done
cat $O/flag_wait.txt
SIZES=12,16,17 timeout 300 python tools/small_sizes.py secp256k1 > $O/small_sizes.txt 2>&1; cat $O/small_sizes.txt
