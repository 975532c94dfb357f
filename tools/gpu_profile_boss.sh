#!/bin/bash
# BOSS CMASS (cells kernel + per-point likelihood kernel): rocprofv3 kernel stats and PMC passes.  Usage: bash tools/gpu_profile_boss.sh <tag>
set -e -o pipefail
TAG=${1:-r01_boss}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o trace -- python3 $R/tools/gpu_boss_loop.py 8 > $OUT/trace.log 2>&1
for C in FETCH_SIZE WRITE_SIZE "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_ANY" "GRBM_GUI_ACTIVE GRBM_COUNT"; do
  N=$(echo $C | tr ' ' '_' | cut -c1-40)
  rocprofv3 --pmc $C --output-format csv -d $OUT/pmc_$N -o pmc -- python3 $R/tools/gpu_boss_loop.py 4 > $OUT/pmc_$N.log 2>&1 || echo "pmc $C failed"
done
cd $R
python tools/summarize_prof.py $OUT > $OUT/summary.txt 2>&1 || true
cat $OUT/summary.txt
