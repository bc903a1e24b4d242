#!/bin/bash
# round-4 evidence batch -> gpurun_out/r04f/ (copied into profiles/r04/ afterwards): PMC counters of both bench workloads + of the
# latency-regime workload (configs[1], 2^16), kernel-trace + PMC passes of the bench command itself, the bench lines, small sizes
# with the A/B switches, criterion shape, EXTEND 2^22, ubenches, split projection, shard emulation
O=gpurun_out/r04f; rm -rf $O; mkdir -p $O
bash tools/prof_counters.sh r04_secp secp256k1 20 3 > /dev/null 2>&1; cp gpurun_out/ctr_r04_secp/counters.json $O/counters_secp256k1_20.json; cp gpurun_out/ctr_r04_secp/kernel_stats.md $O/kernel_stats_secp256k1_20.md
bash tools/prof_counters.sh r04_m31 m31 24 3 > /dev/null 2>&1; cp gpurun_out/ctr_r04_m31/counters.json $O/counters_m31_24.json; cp gpurun_out/ctr_r04_m31/kernel_stats.md $O/kernel_stats_m31_24.md
bash tools/prof_counters.sh r04_secp16 secp256k1 16 5 > /dev/null 2>&1; cp gpurun_out/ctr_r04_secp16/counters.json $O/counters_secp256k1_16.json; cp gpurun_out/ctr_r04_secp16/kernel_stats.md $O/kernel_stats_secp256k1_16.md
# bench.py reads the newest counters under profiles/: make this run see its own
mkdir -p profiles/r04; cp $O/counters_secp256k1_20.json $O/counters_m31_24.json $O/counters_secp256k1_16.json profiles/r04/
bash tools/profile_gpu.sh r04_bench k_stages_lds 380 > /dev/null 2>&1; mkdir -p $O/bench_trace; cp gpurun_out/prof_r04_bench/{command.txt,kernel_stats.md,kernel_hot.json,bench_line.json,pmc_FETCH_SIZE_hot.json,pmc_WRITE_SIZE_hot.json,pmc_SQ_WAVES_hot.json} $O/bench_trace/ 2>/dev/null
python bench.py 2>/dev/null | grep "^{" > $O/bench_default.json
ECFFT_NO_MFMA=1 python bench.py --cpu-log-n 0 2>/dev/null | grep "^{" > $O/bench_default_no_mfma.json
python bench.py --field m31 --log-n 24 --cpu-log-n 0 2>/dev/null | grep "^{" > $O/bench_m31_2e24.json
python bench.py --log-n 16 --cpu-log-n 0 --steps 50 --warmup 5 2>/dev/null | grep "^{" > $O/bench_secp_2e16.json
python bench.py --mode extend-split --log-n 22 --steps 10 --warmup 2 2>/dev/null | grep "^{" > $O/bench_extend_split_2e22_world1.json
python tools/small_sizes.py secp256k1 > $O/small_sizes.txt 2>&1
SIZES=11,12,16,18,20,22 python tools/small_sizes.py m31 >> $O/small_sizes.txt 2>&1
ECFFT_NO_MFMA=1 python tools/small_sizes.py secp256k1 > $O/small_sizes_no_mfma.txt 2>&1
ECFFT_NO_MFMA=1 ECFFT_NO_ROW256=1 ECFFT_NO_COL256=1 ECFFT_SMALL_MIN_LOGC=2 python tools/small_sizes.py secp256k1 > $O/small_sizes_round3_kernels.txt 2>&1
g++ -O2 -std=c++17 -Iinclude examples/bench_fftree.cpp -Lecfft_amd -lecfft_hip -Wl,-rpath,$PWD/ecfft_amd -Wl,--allow-shlib-undefined -o /tmp/bench_fftree && /tmp/bench_fftree > $O/bench_fftree.txt 2>&1
python tools/cold_build.py > $O/build_times.txt 2>&1
(cd tools/ubench && ./mfma_mul16_da2 1024 50 && ./mfma_mul16_stamps 256 10 | grep -A5 "^stamps") > $O/ubench_mfma_mul16.txt 2>&1
python tools/big_sizes_check.py secp256k1:22 m31:25 > $O/big_sizes.txt 2>&1
python tools/shard_emulate.py secp256k1 22 8 > $O/shard_emulate.txt 2>&1
python tools/split_project.py 20 25 48 2>&1 | grep -v amdgpu.ids > $O/split_projection.txt
bash tools/trace_case.sh r04f_16 secp256k1 16 both 3 > /dev/null 2>&1; cp gpurun_out/trace_r04f_16/dispatches.txt $O/dispatches_secp_2e16.txt
ls -la $O; python tools/bench_classes.py < $O/bench_default.json; python tools/bench_classes.py < $O/bench_m31_2e24.json; cat $O/small_sizes.txt | head -9
