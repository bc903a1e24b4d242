#!/bin/bash
# round 6, session 4: the whole GPU suite + smoke + the default bench line on the tree as rebuilt in a fresh container
O=gpurun_out/r06p; rm -rf $O; mkdir -p $O
(time timeout 2400 python -m pytest tests -m gpu -x -q --durations=6) > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log; tail -14 $O/pytest.log
(time timeout 600 python -c "import __graft_entry__ as g; g.smoke()") > $O/smoke.log 2>&1; echo "smoke rc=$?" >> $O/smoke.log; tail -5 $O/smoke.log
(time python bench.py) 2> $O/bench.err | grep "^{" > $O/bench_default.json; tail -4 $O/bench.err
python tools/bench_classes.py < $O/bench_default.json
