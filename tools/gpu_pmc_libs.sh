#!/bin/bash
# PMC comparison of library builds on one workload.  Usage: bash tools/gpu_pmc_libs.sh <tag> <3|boss> lib1.so lib2.so ...
# One rocprofv3 --pmc pass per counter set and build (counters only, no trace domains); condensed by tools/summarize_pmc_libs.py
TAG=$1; WHICH=$2; shift 2
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for LIB in "$@"; do
  export VICTOR_HIP_LIB=$R/$LIB
  B=$(basename $LIB .so)
  I=0
  for C in "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_LDS_UNALIGNED_STALL SQ_INSTS_LDS SQ_INSTS_VALU_TRANS_F64" \
           "GRBM_GUI_ACTIVE SQ_INST_LEVEL_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA SQ_INSTS_SALU SQ_INSTS_SMEM SQ_THREAD_CYCLES_VALU SQ_BUSY_CU_CYCLES"; do
    I=$((I+1))
    timeout -k 10 120 rocprofv3 --pmc $C --output-format csv -d $OUT/${B}_set$I -o pmc -- python3 $R/tools/gpu_loop.py $WHICH 3 > $OUT/${B}_set$I.log 2>&1 || echo "pmc set $I failed for $B"
  done
done
cd $R
python tools/summarize_pmc_libs.py $OUT > $OUT/summary.txt 2>&1
cat $OUT/summary.txt
