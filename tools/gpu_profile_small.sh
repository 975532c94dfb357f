#!/bin/bash
# Small-batch profile (K1 / K2 / gap split).  Usage: bash tools/gpu_profile_small.sh <tag>
set -e -o pipefail
TAG=${1:-r02_small}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for CASE in "3 1 api" "3 1 resident" "3 64 resident" "3 1024 resident" "boss 1 api" "boss 1 resident" "boss 64 resident"; do
  set -- $CASE
  N=c$1_b$2_$3
  python3 $R/tools/small_batch_loop.py $1 $2 $3 400 > $OUT/$N.wall.json 2> $OUT/$N.err
  cat $OUT/$N.wall.json
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/$N -o trace -- python3 $R/tools/small_batch_loop.py $1 $2 $3 200 > $OUT/$N.log 2>&1
done
python3 $R/tools/summarize_small_batch.py $OUT | tee $OUT/summary.txt
