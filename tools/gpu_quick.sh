#!/bin/bash
# Fast iteration: full GPU test suite (parallel) + bench line.
mkdir -p gpurun_out
{
  timeout 1200 python -m pytest tests -q -m gpu --tb=short -p no:cacheprovider -n 6 2>&1 | grep -vE "^\s*$|amdgpu.ids" | cut -c1-600 | tail -40
  timeout 600 python bench.py --no-cpu-baseline 2>&1 | tail -2
} > gpurun_out/gpu_quick.log 2>&1
tail -60 gpurun_out/gpu_quick.log
