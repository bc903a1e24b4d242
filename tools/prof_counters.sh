#!/bin/bash
# Run ON THE GPU BOX: per-class rocprofv3 counters of one ENTER+EXIT workload -> gpurun_out/ctr_TAG/{counters.json,kernel_stats.md}
# usage: tools/prof_counters.sh TAG FIELD LOG_N [REPS]
set -u
TAG=$1; FIELD=$2; LOGN=$3; REPS=${4:-3}
OUT=gpurun_out/ctr_$TAG
rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp
python tools/prof_case.py $FIELD $LOGN both $REPS --count $OUT/launches.json > $OUT/count.log 2>&1
CMD="python tools/prof_case.py $FIELD $LOGN both $REPS"
echo "$CMD" > $OUT/command.txt
timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/trace -o t -- $CMD > $OUT/trace.log 2>&1
python tools/prof_summary.py $OUT/trace/t_results.db "rocprofv3 --kernel-trace --stats -- $CMD" > $OUT/kernel_stats.md 2>&1
i=0
for P in "SQ_WAVES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SALU" \
         "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM" \
         "FETCH_SIZE" "WRITE_SIZE" "GRBM_GUI_ACTIVE GRBM_COUNT" \
         "SQ_INSTS_VALU_MFMA_I8 SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_BUSY_CU_CYCLES SQ_INSTS_MFMA"; do
  i=$((i+1))
  timeout 600 rocprofv3 --kernel-trace --pmc $P -d $OUT/pmc$i -o p -- $CMD > $OUT/pmc$i.log 2>&1
done
python tools/counters_json.py $OUT $FIELD $LOGN $REPS > $OUT/counters.log 2>&1
rm -rf $OUT/trace $OUT/pmc*/
head -c 3000 $OUT/counters.json
