#!/bin/bash
# round 6, lease B: (1) real-RCCL self-exchange floor at world = 1; (2) per-class counters of the BATCHED mode (8 polynomials of 2^20 per
# call) on the shipped library (two half-batches on two streams) and on the one-stream build (bs0), and of the single transform on the
# barrier build (wl0) next to the shipped one — the counters behind the wave-local A/B
O=gpurun_out/r06b; rm -rf $O; mkdir -p $O
(time timeout 300 python tools/rccl_self_exchange.py $O/rccl_self_exchange.txt) > $O/rccl.log 2>&1; cat $O/rccl_self_exchange.txt; tail -4 $O/rccl.log
ECFFT_PROF_BATCH=8 bash tools/prof_counters.sh r06_b8 secp256k1 20 2 > /dev/null 2>&1; cp gpurun_out/ctr_r06_b8/counters.json $O/counters_secp256k1_20_b8.json; cp gpurun_out/ctr_r06_b8/kernel_stats.md $O/kernel_stats_secp256k1_20_b8.md
ECFFT_PROF_BATCH=8 ECFFT_LIB=$PWD/ecfft_amd/variants/bs0.so bash tools/prof_counters.sh r06_b8_one secp256k1 20 2 > /dev/null 2>&1; cp gpurun_out/ctr_r06_b8_one/counters.json $O/counters_secp256k1_20_b8_onestream.json
bash tools/prof_counters.sh r06_20 secp256k1 20 3 > /dev/null 2>&1; cp gpurun_out/ctr_r06_20/counters.json $O/counters_secp256k1_20.json; cp gpurun_out/ctr_r06_20/kernel_stats.md $O/kernel_stats_secp256k1_20.md
ECFFT_LIB=$PWD/ecfft_amd/variants/wl0.so bash tools/prof_counters.sh r06_20_wl0 secp256k1 20 3 > /dev/null 2>&1; cp gpurun_out/ctr_r06_20_wl0/counters.json $O/counters_secp256k1_20_wl0.json
for f in counters_secp256k1_20_b8 counters_secp256k1_20_b8_onestream counters_secp256k1_20 counters_secp256k1_20_wl0; do echo "== $f"; python tools/class_breakdown.py $O/$f.json $([[ $f == *b8* ]] && echo 8 || echo 1); done > $O/class_breakdown.txt 2>&1
cat $O/class_breakdown.txt
ls gpurun_out/ctr_r06_b8/
