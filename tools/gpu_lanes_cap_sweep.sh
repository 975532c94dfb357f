for CAP in 8 16 32 64 128 1000; do
  VICTOR_HIP_DEV=1 VICTOR_HIP_LANES_CAP=$CAP python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-boss 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('cap $CAP: K1 %.3f ms  %.0f evals/s' % (d['kernels_ms']['theory'], d['value']))"
done
