#!/bin/bash
# round-2 GPU session 1: full GPU test suite, baseline bench lines, clock ubench, M31 counters
mkdir -p gpurun_out/s1
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/s1/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/s1/pytest.log
tools/ubench/clock > gpurun_out/s1/clock.txt 2>&1
timeout 600 python bench.py > gpurun_out/s1/bench_secp.json 2> gpurun_out/s1/bench_secp.err
timeout 600 python bench.py --field m31 --log-n 24 --cpu-log-n 0 > gpurun_out/s1/bench_m31.json 2> gpurun_out/s1/bench_m31.err
timeout 900 tools/profile_gpu.sh r02_m31_base k_stages_lds 1 --field m31 --log-n 24 > gpurun_out/s1/prof_m31.log 2>&1
tail -3 gpurun_out/s1/pytest.log; cat gpurun_out/s1/clock.txt
