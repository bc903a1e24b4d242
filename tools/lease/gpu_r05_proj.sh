#!/bin/bash
# round 5: split projection — the default path at 2^20, and the all-gather EXIT form at 2^22 against the split levels
O=gpurun_out/r05proj; rm -rf $O; mkdir -p $O
timeout 900 python tools/split_project.py 20 25 48 > $O/split_projection_2e20.txt 2>&1; tail -12 $O/split_projection_2e20.txt
WORLDS=8 timeout 900 python tools/split_project.py 22 25 48 > $O/split_projection_2e22_default.txt 2>&1; tail -9 $O/split_projection_2e22_default.txt
WORLDS=8 ECFFT_SPLIT_GATHER_MAX_LOG=22 timeout 900 python tools/split_project.py 22 25 48 > $O/split_projection_2e22_gather.txt 2>&1; tail -9 $O/split_projection_2e22_gather.txt
