#!/bin/bash
mkdir -p gpurun_out
bash tools/profile.sh bf16x3 --mode render_only > gpurun_out/r3l_profile_render.log 2>&1; echo "profile rc=$?"
cp gpurun_out/prof_bf16x3/kernel_stats.csv gpurun_out/r3l_render_kernel_stats.csv
cp gpurun_out/prof_bf16x3/pmc_summary.csv gpurun_out/r3l_render_pmc_summary.csv
head -6 gpurun_out/r3l_render_kernel_stats.csv | cut -c1-150
