#!/bin/bash
# 20 more runs per family (fp32, fp16x3) and 10 of fp16x3w on one pair (default 0 = (5,6); 1 = (4,6)), 10,000 steps: two processes share
# the GPU.  usage: bash tools/gpu_equiv_pool.sh [pair] [tag]
PAIR=${1:-0}; TAG=${2:-}
mkdir -p gpurun_out
python tools/exp_equivalence_long.py --steps 10000 --pairs $PAIR --twin-range 5:15 --precisions fp32,fp16x3,fp16x3w --out gpurun_out/r06_equivalence_10k${TAG}_b.json > gpurun_out/r06_equiv${TAG}_b.out 2>&1 &
python tools/exp_equivalence_long.py --steps 10000 --pairs $PAIR --twin-range 15:25 --precisions fp32,fp16x3 --out gpurun_out/r06_equivalence_10k${TAG}_c.json > gpurun_out/r06_equiv${TAG}_c.out 2>&1 &
wait
tail -n 3 gpurun_out/r06_equiv${TAG}_b.out gpurun_out/r06_equiv${TAG}_c.out | cut -c1-200
