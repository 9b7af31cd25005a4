#!/bin/bash
# scratch driver for the experiment of the moment (gpurun)
mkdir -p gpurun_out
bash tools/profile.sh fp32 > gpurun_out/prof_fp32.log 2>&1; tail -2 gpurun_out/prof_fp32.log
bash tools/profile.sh mixed > gpurun_out/prof_mixed.log 2>&1; tail -2 gpurun_out/prof_mixed.log
