#!/bin/bash
# bench.py as two launched ranks SHARING the one GPU of a gpurun box, through the RCCL stand-in of tests/rccl_double (development
# switch): the N > 1 communicator branch with the BASELINE config 4 / 5 legs.  Says nothing about RCCL's or xGMI's speed - the
# record shows that the legs run and check themselves.  Usage: bash tools/gpu_two_rank_double.sh <tag> [ranks, default 2, at most 5]
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$1
mkdir -p $OUT
cd $R
gcc -shared -fPIC -O2 -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include tests/rccl_double/rccl_double.c -L/opt/rocm/lib -lamdhip64 \
    -Wl,-rpath,/opt/rocm/lib -o $OUT/librccl_double.so || exit 1
N=${2:-2}
PORT=$((29600 + RANDOM % 200))
for RANK in $(seq 0 $((N - 1))); do
  VICTOR_HIP_DEV=1 VICTOR_HIP_RCCL_LIB=$OUT/librccl_double.so VICTOR_HIP_RCCL_SHARED_DEVICE_OK=1 RCCL_DOUBLE_DIR=$OUT \
  RANK=$RANK LOCAL_RANK=$RANK WORLD_SIZE=$N MASTER_ADDR=127.0.0.1 MASTER_PORT=$PORT \
    timeout -k 5 300 python tools/with_watchdog.py 280 bench.py --gpus $N --steps 5 --warmup 2 --batch 16384 --no-cpu-baseline \
    > $OUT/bench_two_ranks_double_rank$RANK.json 2> $OUT/bench_two_ranks_double_rank$RANK.err &
done
wait
tail -c 4000 $OUT/bench_two_ranks_double_rank0.json; for RANK in $(seq 0 $((N - 1))); do tail -3 $OUT/bench_two_ranks_double_rank$RANK.err; done
rm -f $OUT/librccl_double.so
