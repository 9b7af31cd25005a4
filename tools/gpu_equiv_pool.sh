#!/bin/bash
# 20 more runs per family (fp32, fp16x3) and 10 of fp16x3w on pair (5,6), 10,000 steps: two processes share the GPU (a 1024-ray step
# leaves most of it idle)
mkdir -p gpurun_out
python tools/exp_equivalence_long.py --steps 10000 --pairs 0 --twin-range 5:15 --precisions fp32,fp16x3,fp16x3w --out gpurun_out/r06_equivalence_10k_b.json > gpurun_out/r06_equiv_b.out 2>&1 &
python tools/exp_equivalence_long.py --steps 10000 --pairs 0 --twin-range 15:25 --precisions fp32,fp16x3 --out gpurun_out/r06_equivalence_10k_c.json > gpurun_out/r06_equiv_c.out 2>&1 &
wait
tail -3 gpurun_out/r06_equiv_b.out gpurun_out/r06_equiv_c.out | cut -c1-200
