#!/bin/bash
# round 5: link striping of the pairwise exchanges — split projection with / without it (per-link billing of the projection transport)
O=gpurun_out/r05stripe; rm -rf $O; mkdir -p $O
WORLDS=8 ECFFT_NO_STRIPE=1 timeout 900 python tools/split_project.py 22 25 48 2>&1 | grep -v amdgpu.ids > $O/split_projection_2e22_plain.txt; tail -9 $O/split_projection_2e22_plain.txt | cut -c1-200
WORLDS=8 timeout 900 python tools/split_project.py 22 25 48 2>&1 | grep -v amdgpu.ids > $O/split_projection_2e22_striped.txt; tail -9 $O/split_projection_2e22_striped.txt | cut -c1-200
WORLDS=4,8 ECFFT_NO_STRIPE=1 timeout 900 python tools/split_project.py 20 25 48 2>&1 | grep -v amdgpu.ids > $O/split_projection_2e20_plain.txt; tail -14 $O/split_projection_2e20_plain.txt | cut -c1-200
WORLDS=4,8 timeout 900 python tools/split_project.py 20 25 48 2>&1 | grep -v amdgpu.ids > $O/split_projection_2e20_striped.txt; tail -14 $O/split_projection_2e20_striped.txt | cut -c1-200
