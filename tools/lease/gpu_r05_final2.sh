#!/bin/bash
# round 5, last call: the whole GPU suite + smoke on the final tree, then the evidence batch of tools/gpu_r05_final.sh (counters carry the
# hash of the kernels they were taken from: they must be regenerated after the last kernel change)
O=gpurun_out/r05suite2; rm -rf $O; mkdir -p $O
(time timeout 2400 python -m pytest tests -m gpu -x -q --durations=6) > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log; tail -14 $O/pytest.log | head -9
(time timeout 600 python -c "import __graft_entry__ as g; g.smoke()") > $O/smoke.log 2>&1; echo "smoke rc=$?" >> $O/smoke.log; tail -5 $O/smoke.log
bash tools/gpu_r05_final.sh 2>&1 | tail -22
