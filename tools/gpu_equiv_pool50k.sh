#!/bin/bash
# 8 more runs per family (fp32, fp16x3) on pair (5,6) to 50,000 steps: two processes share the GPU
mkdir -p gpurun_out
python tools/exp_equivalence_long.py --steps 50000 --pairs 0 --twin-range 3:7 --precisions fp32,fp16x3 --out gpurun_out/r06_equivalence_50k_b.json > gpurun_out/r06_equiv50_b.out 2>&1 &
python tools/exp_equivalence_long.py --steps 50000 --pairs 0 --twin-range 7:11 --precisions fp32,fp16x3 --out gpurun_out/r06_equivalence_50k_c.json > gpurun_out/r06_equiv50_c.out 2>&1 &
wait
tail -n 2 gpurun_out/r06_equiv50_b.out gpurun_out/r06_equiv50_c.out | cut -c1-200
