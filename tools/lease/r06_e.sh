#!/bin/bash
# round 6, lease E: the refactored wire-format reader (wire_parse.h) on the GPU (serialize + crate-pin consumer tests), then the two soaks on
# the final build: every size 2^0..2^15, every algorithm, both fields against the oracle; matrix-core against all-VALU over random shapes
O=gpurun_out/r06e; rm -rf $O; mkdir -p $O
(time timeout 900 python -m pytest tests/test_serialize.py tests/test_rust_pin.py tests/test_abi_host.py -x -q) > $O/serialize.log 2>&1; tail -4 $O/serialize.log
(time timeout 1500 python tools/soak_gpu.py 15) > $O/soak_gpu.txt 2>&1; tail -4 $O/soak_gpu.txt
(time timeout 600 python tools/soak_mfma.py 120) > $O/soak_mfma.txt 2>&1; tail -4 $O/soak_mfma.txt
