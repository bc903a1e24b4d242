#!/bin/bash
# round 3: instruction-rate and MFMA-multiply microbenchmarks on the GPU box -> gpurun_out/r03/
mkdir -p gpurun_out/r03
cd tools/ubench
CLOCK_R3_ONLY=1 timeout 300 ./clock > ../../gpurun_out/r03/clock_r3.txt 2>&1
for v in u4w4 u8w4 u4w2 u16w2; do
  echo "== mfma_mul_$v" >> ../../gpurun_out/r03/mfma_mul.txt
  timeout 300 ./mfma_mul_$v 2048 50 >> ../../gpurun_out/r03/mfma_mul.txt 2>&1
done
cat ../../gpurun_out/r03/clock_r3.txt ../../gpurun_out/r03/mfma_mul.txt
