#!/bin/bash
# Run ON THE GPU BOX (through gpurun): rocprofv3 kernel-trace stats + PMC passes of the bench command; keeps only
# the text summaries (the rocpd databases exceed gpurun's 64 MiB return limit).
# Usage: tools/profile_gpu.sh <tag> <dominant kernel, FULL instantiation e.g. "k_stages_lds<ecfft::Secp256k1, 10>"> <hot-path launches of it = (steps+warmup)*launches_per_step> [bench args...]
# The traced command must dispatch NOTHING of that kernel family after the timed steps: bench.py's latency_regime section (2^16 / 2^17
# transforms after the timed loop, small <..., 0> instantiations) is switched off for it, and the kernel is matched by its full
# instantiation — round 4's kernel_hot.json averaged the small launches instead of the timed ones.
set -u
export ECFFT_BENCH_NO_LATENCY=1
TAG=$1; KERN=$2; LAST=$3; shift 3
OUT=gpurun_out/prof_$TAG
rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp
CMD="python bench.py --steps 3 --warmup 1 --no-profile --cpu-log-n 0 --batch 0 $*"
echo "$CMD" > $OUT/command.txt
timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/trace -o trace -- $CMD > $OUT/trace.log 2>&1
python tools/prof_summary.py $OUT/trace/trace_results.db "rocprofv3 --kernel-trace --stats -- $CMD" > $OUT/kernel_stats.md
python tools/prof_summary.py $OUT/trace/trace_results.db x --kernel "$KERN" --last $LAST > $OUT/kernel_hot.json
rm -rf $OUT/trace
for P in FETCH_SIZE WRITE_SIZE "SQ_WAVES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT"; do
  N=$(echo $P | cut -d' ' -f1)
  timeout 300 rocprofv3 --kernel-trace --pmc $P -d $OUT/pmc_$N -o pmc -- $CMD > $OUT/pmc_$N.log 2>&1
  python tools/pmc_summary.py $OUT/pmc_$N/pmc_results.db > $OUT/pmc_$N.md 2>&1
  python tools/pmc_summary.py $OUT/pmc_$N/pmc_results.db --kernel "$KERN" --last $LAST > $OUT/pmc_${N}_hot.json 2>&1
  rm -rf $OUT/pmc_$N
done
grep -h '"metric"' $OUT/trace.log | tail -1 > $OUT/bench_line.json
ls -la $OUT
