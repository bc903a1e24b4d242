#!/bin/bash
# round-6 evidence batch -> gpurun_out/r06f/ (copied into profiles/r06/ afterwards): the whole GPU suite + smoke on the final tree, PMC
# counters of both bench workloads + of the latency-regime workload (configs[1], 2^16) — they carry the hash of the kernel sources and
# must be regenerated after the last kernel change —, kernel-trace + PMC passes of the bench command itself (tools/profile_gpu.sh), the
# bench lines, small sizes, criterion shape, build times
O=gpurun_out/r06f; rm -rf $O; mkdir -p $O
(time timeout 2400 python -m pytest tests -m gpu -x -q --durations=6) > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log; tail -14 $O/pytest.log | head -9
(time timeout 600 python -c "import __graft_entry__ as g; g.smoke()") > $O/smoke.log 2>&1; echo "smoke rc=$?" >> $O/smoke.log; tail -5 $O/smoke.log
bash tools/prof_counters.sh r06_secp secp256k1 20 3 > /dev/null 2>&1; cp gpurun_out/ctr_r06_secp/counters.json $O/counters_secp256k1_20.json; cp gpurun_out/ctr_r06_secp/kernel_stats.md $O/kernel_stats_secp256k1_20.md
bash tools/prof_counters.sh r06_m31 m31 24 3 > /dev/null 2>&1; cp gpurun_out/ctr_r06_m31/counters.json $O/counters_m31_24.json; cp gpurun_out/ctr_r06_m31/kernel_stats.md $O/kernel_stats_m31_24.md
bash tools/prof_counters.sh r06_secp16 secp256k1 16 5 > /dev/null 2>&1; cp gpurun_out/ctr_r06_secp16/counters.json $O/counters_secp256k1_16.json; cp gpurun_out/ctr_r06_secp16/kernel_stats.md $O/kernel_stats_secp256k1_16.md
# bench.py reads the newest counters under profiles/: make this run see its own
mkdir -p profiles/r06; cp $O/counters_secp256k1_20.json $O/counters_m31_24.json $O/counters_secp256k1_16.json profiles/r06/
# dominant kernel CLASS of the bench command (every k_stages_lds<...> instantiation, as in counters_*.json): 95 launches per step x (3 + 1) steps
bash tools/profile_gpu.sh r06_bench "k_stages_lds<" 380 > /dev/null 2>&1; mkdir -p $O/bench_trace; cp gpurun_out/prof_r06_bench/{command.txt,kernel_stats.md,kernel_hot.json,bench_line.json,pmc_FETCH_SIZE_hot.json,pmc_WRITE_SIZE_hot.json,pmc_SQ_WAVES_hot.json} $O/bench_trace/ 2>/dev/null
python bench.py 2>/dev/null | grep "^{" > $O/bench_default.json
python bench.py --steps 20 --warmup 5 2>/dev/null | grep "^{" > $O/bench_default_rerun.json
python bench.py --field m31 --log-n 24 --cpu-log-n 0 2>/dev/null | grep "^{" > $O/bench_m31_2e24.json
python bench.py --log-n 16 --cpu-log-n 0 --steps 50 --warmup 5 2>/dev/null | grep "^{" > $O/bench_secp_2e16.json
python bench.py --mode extend-split --log-n 22 --steps 10 --warmup 2 2>/dev/null | grep "^{" > $O/bench_extend_split_2e22_world1.json
python tools/small_sizes.py secp256k1 > $O/small_sizes.txt 2>&1
SIZES=11,12,16,18,20,22 python tools/small_sizes.py m31 >> $O/small_sizes.txt 2>&1
g++ -O2 -std=c++17 -Iinclude examples/bench_fftree.cpp -Lecfft_amd -lecfft_hip -Wl,-rpath,$PWD/ecfft_amd -Wl,--allow-shlib-undefined -o /tmp/bench_fftree && /tmp/bench_fftree > $O/bench_fftree.txt 2>&1
python tools/cold_build.py > $O/build_times.txt 2>&1
ls -la $O $O/bench_trace; cat $O/bench_trace/kernel_hot.json; python tools/bench_classes.py < $O/bench_default.json; head -9 $O/small_sizes.txt
