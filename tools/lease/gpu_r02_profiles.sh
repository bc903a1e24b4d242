#!/bin/bash
# round-2 evidence for profiles/r02: counters of both workloads, bench lines, microbenchmarks, latency tables
mkdir -p gpurun_out/p
tools/prof_counters.sh secp20 secp256k1 20 3 > /dev/null 2>&1
tools/prof_counters.sh m31_24 m31 24 3 > /dev/null 2>&1
cp gpurun_out/ctr_secp20/counters.json profiles/r02/counters_secp256k1_20.json
cp gpurun_out/ctr_m31_24/counters.json profiles/r02/counters_m31_24.json
python bench.py > gpurun_out/p/bench_default.json 2> gpurun_out/p/bench_default.err
python bench.py --field m31 --log-n 24 --cpu-log-n 0 > gpurun_out/p/bench_m31_2e24.json 2> gpurun_out/p/bench_m31.err
tools/ubench/clock > gpurun_out/p/clock_ubench.txt 2>&1
tools/ubench/sweep > gpurun_out/p/sweep_ubench.txt 2>&1
python tools/small_sizes.py secp256k1 > gpurun_out/p/small_sizes.txt 2>&1
SIZES=11,12,16,18,20,22 python tools/small_sizes.py m31 >> gpurun_out/p/small_sizes.txt 2>&1
g++ -O2 -std=c++17 -Iinclude examples/bench_fftree.cpp -Lecfft_amd -lecfft_hip -Wl,-rpath,$PWD/ecfft_amd -Wl,--allow-shlib-undefined -o /tmp/bench_fftree && /tmp/bench_fftree > gpurun_out/p/bench_fftree.txt 2>&1
python bench.py --mode extend-split --log-n 22 --steps 10 --warmup 2 2>/dev/null | grep "^{" > gpurun_out/p/bench_extend_split_2e22_world1.json
tail -c 600 gpurun_out/p/bench_default.json
