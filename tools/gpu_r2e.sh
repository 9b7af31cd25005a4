#!/bin/bash
mkdir -p gpurun_out
{
for rep in 1 2; do
for lib in libexp_head.so libnerf_hip.so; do
  echo "== $lib (rep $rep)"; timeout 300 python tools/exp_fwd3.py $lib --bwd 2>&1 | grep -v amdgpu.ids
done; done
} > gpurun_out/r2e_exp.log 2>&1
cat gpurun_out/r2e_exp.log
