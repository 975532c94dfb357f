#!/bin/bash
# One RSD model of the BOSS CMASS configuration under rocprofv3: kernel stats, then PMC passes (each its own run; the program
# directly after `--`).  Usage: bash tools/gpu_profile_model.sh <rsd_model> <tag> [batch]
# Integrand points per launch for the per-point instruction counts: batch x 30 s x 100 mu x 50 v (dispersion, streaming),
# batch x 30 s x 100 mu (kaiser, euclid_special: no velocity integral).
set -e -o pipefail
RSD=${1:-dispersion}
TAG=${2:-r04_$RSD}
BATCH=${3:-16384}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o trace -- python3 $R/tools/gpu_model_loop.py $RSD 8 $BATCH > $OUT/trace.log 2>&1
for C in FETCH_SIZE WRITE_SIZE "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_ANY" "GRBM_GUI_ACTIVE GRBM_COUNT"; do
  N=$(echo $C | tr ' ' '_' | cut -c1-40)
  rocprofv3 --pmc $C --output-format csv -d $OUT/pmc_$N -o pmc -- python3 $R/tools/gpu_model_loop.py $RSD 4 $BATCH > $OUT/pmc_$N.log 2>&1 || echo "pmc $C failed"
done
cd $R
case $RSD in kaiser|euclid_special) PTS=$((BATCH * 30 * 100));; *) PTS=$((BATCH * 30 * 100 * 50));; esac
python tools/summarize_prof.py $OUT --points $PTS --label "BOSS CMASS, rsd_model=$RSD, batch $BATCH" > $OUT/summary.txt 2>&1 || true
cat $OUT/summary.txt
