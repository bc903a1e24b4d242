#!/bin/bash
# round-4 baseline: full GPU tests, the bench line, small sizes, per-dispatch timelines of the latency regime (2^16 / 2^17)
O=gpurun_out/r04base; rm -rf $O; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log; tail -5 $O/pytest.log
python bench.py 2>/dev/null | grep "^{" > $O/bench_default.json; python tools/bench_classes.py < $O/bench_default.json
python tools/small_sizes.py secp256k1 > $O/small_sizes.txt 2>&1; cat $O/small_sizes.txt
bash tools/trace_case.sh r04_16 secp256k1 16 both 3 > /dev/null 2>&1; cp gpurun_out/trace_r04_16/dispatches.txt $O/dispatches_2e16.txt
bash tools/trace_case.sh r04_17 secp256k1 17 both 3 > /dev/null 2>&1; cp gpurun_out/trace_r04_17/dispatches.txt $O/dispatches_2e17.txt
tail -3 $O/dispatches_2e16.txt $O/dispatches_2e17.txt
