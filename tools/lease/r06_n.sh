#!/bin/bash
# round 6, lease N: examples/sharded_extend (real RCCL, world = 1, plain C++ process) timed out once in the suite — flaky or real?  Ten runs with a
# 60 s limit each (NCCL_DEBUG=INFO on a run that hangs), then the whole GPU suite without -x
O=gpurun_out/r06n; rm -rf $O; mkdir -p $O
g++ -O2 -std=c++17 -Iinclude examples/sharded_extend.cpp -Lecfft_amd -lecfft_hip -Wl,-rpath,$PWD/ecfft_amd -Wl,--allow-shlib-undefined -o /tmp/sharded_extend 2> $O/build.log
for i in 1 2 3 4 5 6 7 8 9 10; do
  s=$(date +%s.%N); timeout 60 /tmp/sharded_extend > $O/run_$i.out 2> $O/run_$i.err; rc=$?; e=$(date +%s.%N)
  echo "run $i rc=$rc $(echo "$e - $s" | bc) s  $(tail -1 $O/run_$i.out)" | tee -a $O/runs.txt
  if [ $rc -eq 124 ]; then NCCL_DEBUG=INFO timeout 60 /tmp/sharded_extend > $O/hang_$i.out 2> $O/hang_$i.err; echo "  with NCCL_DEBUG=INFO rc=$?" | tee -a $O/runs.txt; tail -5 $O/hang_$i.out $O/hang_$i.err | cut -c1-200 | tee -a $O/runs.txt; fi
done
(time timeout 2400 python -m pytest tests -m gpu -q --durations=6) > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log; tail -12 $O/pytest.log
