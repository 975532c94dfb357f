#!/bin/bash
# A/B of the two theory-kernel mappings on the bench workload
set -e -o pipefail
python -m pytest tests -m gpu -x -q 2>&1 | tail -5
for M in point lanes point lanes; do
  VICTOR_HIP_DEV=1 VICTOR_HIP_MAPPING=$M python bench.py --steps 10 --warmup 2 --no-cpu-baseline | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$M value %.0f evals/s  K1 %.3f ms  K2 %.3f ms' % (d['value'], d['kernels_ms']['theory'], d['kernels_ms']['likelihood']))"
done
