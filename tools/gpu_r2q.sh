#!/bin/bash
# the N > 1 bench path on a 1-GPU box: two ranks share cuda:0, gloo carries the gradient all-reduce (RCCL refuses two ranks
# on one device).  Exercises sharding, broadcast, all-reduce of the flat buckets, barriers, max-over-ranks timing, JSON.
mkdir -p gpurun_out
NERF_ALLOW_SHARED_GPU=1 timeout 600 python bench.py --gpus 2 --backend gloo --steps 4 --warmup 2 --single-datapath --no-gate > gpurun_out/r2q_bench_2rank.json 2> gpurun_out/r2q_bench_2rank.err; echo "2-rank weak rc=$?"
cut -c1-700 gpurun_out/r2q_bench_2rank.json; tail -3 gpurun_out/r2q_bench_2rank.err
NERF_ALLOW_SHARED_GPU=1 timeout 600 python bench.py --gpus 2 --backend gloo --steps 2 --warmup 1 --single-datapath --no-gate --strong > gpurun_out/r2q_bench_2rank_strong.json 2> gpurun_out/r2q_bench_2rank_strong.err; echo "2-rank strong rc=$?"
cut -c1-400 gpurun_out/r2q_bench_2rank_strong.json
NERF_ALLOW_SHARED_GPU=1 timeout 600 python bench.py --gpus 2 --backend gloo --steps 2 --warmup 1 --mode render_only --no-gate > gpurun_out/r2q_bench_2rank_render.json 2> gpurun_out/r2q_bench_2rank_render.err; echo "2-rank render_only rc=$?"
cut -c1-400 gpurun_out/r2q_bench_2rank_render.json
