#!/bin/bash
# Run on the GPU box (via gpurun): bench + rocprofv3 kernel stats + PMC passes.  Usage: bash tools/gpu_profile.sh <tag>
set -e -o pipefail
TAG=${1:-r01}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
python bench.py > $OUT/bench_n1.json 2> $OUT/bench_n1.err
tail -c 3000 $OUT/bench_n1.json
python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29517 \
    bench.py --gpus 1 --steps 5 --warmup 2 > $OUT/bench_torchrun1.json 2> $OUT/bench_torchrun1.err || { tail -20 $OUT/bench_torchrun1.err; exit 1; }
tail -c 1500 $OUT/bench_torchrun1.json
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o trace -- python3 $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-boss > $OUT/trace.log 2>&1
for C in FETCH_SIZE WRITE_SIZE "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_ANY SQ_INSTS_VALU_MFMA_F64" "GRBM_GUI_ACTIVE GRBM_COUNT"; do
  N=$(echo $C | tr ' ' '_' | cut -c1-40)
  rocprofv3 --pmc $C --output-format csv -d $OUT/pmc_$N -o pmc -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-boss > $OUT/pmc_$N.log 2>&1 || echo "pmc $C failed"
done
cd $R
find $OUT -name "*.csv" | head -40
python tools/summarize_prof.py $OUT > $OUT/summary.txt 2>&1 || true
cat $OUT/summary.txt
