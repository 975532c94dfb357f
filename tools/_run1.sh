set -o pipefail
mkdir -p gpurun_out
timeout -k 10 1000 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 > gpurun_out/r02b_t2.log; cat gpurun_out/r02b_t2.log
