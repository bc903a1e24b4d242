#!/bin/bash
# usage: tools/prof_case.sh TAG MIN_GRID FIELD LOG_N OP [REPS]   -> gpurun_out/case_TAG/summary.txt
set -u
TAG=$1; MING=$2; shift 2
OUT=gpurun_out/case_$TAG
rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp
CMD="python tools/prof_case.py $*"
timeout 300 rocprofv3 --kernel-trace -d $OUT/trace -o t -- $CMD > $OUT/trace.log 2>&1
DBS=""
i=0
for P in "SQ_WAVES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY" \
         "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS" \
         "FETCH_SIZE" "WRITE_SIZE" "GRBM_GUI_ACTIVE GRBM_COUNT"; do
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $P -d $OUT/pmc$i -o p -- $CMD > $OUT/pmc$i.log 2>&1
  DBS="$DBS $OUT/pmc$i/p_results.db"
done
python tools/pmc_by_grid.py --min-grid $MING $OUT/trace/t_results.db $DBS > $OUT/summary.txt 2>&1
rm -rf $OUT/trace $OUT/pmc*/
cat $OUT/summary.txt
