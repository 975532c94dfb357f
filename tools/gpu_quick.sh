#!/bin/bash
# quick GPU check: parity tests then a short bench line
set -e -o pipefail
mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q 2>&1 | tail -15
python bench.py --steps 10 --warmup 2 --no-cpu-baseline | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('value %.0f evals/s  ms/step %.3f  K1 %.3f ms  K2 %.3f ms  frac %.3f' % (d['value'], d['ms_per_step'], d['kernels_ms']['theory'], d['kernels_ms']['likelihood'], d['roofline']['frac']))"
