#!/bin/bash
# scratch driver for the experiment of the moment (gpurun)
mkdir -p gpurun_out
bash tools/profile.sh bf16x3 > gpurun_out/prof_train.log 2>&1; tail -3 gpurun_out/prof_train.log
TAG=bf16x3_render_only bash tools/profile.sh bf16x3 --mode render_only --steps 2 --warmup 1 > gpurun_out/prof_render.log 2>&1; tail -3 gpurun_out/prof_render.log
