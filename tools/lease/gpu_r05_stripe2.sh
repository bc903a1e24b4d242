#!/bin/bash
# round 5: the whole GPU suite on the striping build, then the projections with the shipped 4 MiB threshold
O=gpurun_out/r05stripe2; rm -rf $O; mkdir -p $O
(time timeout 2400 python -m pytest tests -m gpu -x -q) > $O/pytest.log 2>&1; tail -6 $O/pytest.log
WORLDS=8 timeout 900 python tools/split_project.py 22 25 48 2>&1 | grep -v amdgpu.ids > $O/split_projection_2e22_striped.txt; tail -9 $O/split_projection_2e22_striped.txt | cut -c1-200
timeout 900 python tools/split_project.py 20 25 48 2>&1 | grep -v amdgpu.ids > $O/split_projection.txt; tail -16 $O/split_projection.txt | cut -c1-200
