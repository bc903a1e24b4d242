#!/bin/bash
# round 5: the whole GPU suite + smoke on a fresh box
O=gpurun_out/r05suite; rm -rf $O; mkdir -p $O
(time timeout 2400 python -m pytest tests -m gpu -x -q --durations=12) > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log; tail -30 $O/pytest.log
(time timeout 600 python -c "import __graft_entry__ as g; g.smoke()") > $O/smoke.log 2>&1; echo "smoke rc=$?" >> $O/smoke.log; tail -8 $O/smoke.log
