#!/bin/bash
# Two launched ranks of examples/run_walkers.py and of bench.py on one GPU, each under a 100 s watchdog; logs per rank.
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$1
mkdir -p $OUT
cd $R
for WHAT in walkers bench; do
  PORT=$((29600 + RANDOM % 200))
  for RANK in 0 1; do
    if [ $WHAT = walkers ]; then ARGS="examples/run_walkers.py --steps 20"; else ARGS="bench.py --gpus 2 --steps 2 --warmup 1 --batch 4096 --no-cpu-baseline --no-boss"; fi
    RANK=$RANK LOCAL_RANK=$RANK WORLD_SIZE=2 MASTER_ADDR=127.0.0.1 MASTER_PORT=$PORT timeout -k 5 130 python tools/with_watchdog.py 100 $ARGS > $OUT/${WHAT}_rank$RANK.out 2> $OUT/${WHAT}_rank$RANK.err &
  done
  wait
  echo "== $WHAT"; tail -c 600 $OUT/${WHAT}_rank0.out; tail -5 $OUT/${WHAT}_rank0.err; tail -5 $OUT/${WHAT}_rank1.err
done
echo "== one process, two contexts"
timeout -k 5 130 python tools/with_watchdog.py 100 bench.py --gpus 2 --steps 2 --warmup 1 --batch 4096 --no-cpu-baseline --no-boss > $OUT/bench_group.out 2> $OUT/bench_group.err; tail -c 600 $OUT/bench_group.out; tail -5 $OUT/bench_group.err
timeout -k 5 130 python tools/with_watchdog.py 100 examples/run_walkers.py --steps 20 --gpus 2 > $OUT/walkers_group.out 2> $OUT/walkers_group.err; tail -c 600 $OUT/walkers_group.out; tail -5 $OUT/walkers_group.err
