#!/bin/bash
# round 6, lease D: interleaved A/B of the register-resident wave-local sweeps of the row kernel (-DECFFT_ROW_REGS=1: distance 64 / 32 / 16
# without LDS in between, v_permlane32_swap / v_permlane16_swap) against the shipped build, then parity of the variant at the sizes
# that run the row kernel (2^19 / 2^20 vs the oracle need the background job: the quick check here is the round trip + A/B equality)
O=gpurun_out/r06d; rm -rf $O; mkdir -p $O
V=ecfft_amd/variants
{
for ln in 20 18 22; do echo "== secp256k1 2^$ln  (libecfft_hip = LDS between the wave-local sweeps, rr1 = registers)"; python tools/ab_many.py secp256k1 $ln ecfft_amd/libecfft_hip.so $V/rr1.so 2>&1 | tail -2; done
echo "== secp256k1 2^20 x 8 (batched)"; python tools/ab_many.py secp256k1 20 --count 8 ecfft_amd/libecfft_hip.so $V/rr1.so 2>&1 | tail -2
} > $O/row_regs_ab.txt 2>&1
cat $O/row_regs_ab.txt
ECFFT_LIB=$PWD/$V/rr1.so timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "headline or golden or matches_oracle or batched" > $O/parity_rr1.log 2>&1; tail -4 $O/parity_rr1.log
