#!/bin/bash
# One gpurun call: GPU parity tests, contract bench, rocprofv3 kernel stats.  Output lands in gpurun_out/.
mkdir -p gpurun_out
R=${GRAFT_REPO_ROOT:-$(pwd)}
{
  echo "== pytest -m gpu"
  timeout 1500 python -m pytest tests -q -m gpu --tb=short -p no:cacheprovider -n 8 -rP 2>&1 | grep -vE "^\s*$|amdgpu.ids|Captured|^-+$" | cut -c1-1500 | tail -150
  echo "== bench"
  timeout 900 python bench.py 2>&1 | tail -1 | tee gpurun_out/bench_default.json | cut -c1-600
  echo "== smoke"
  timeout 300 python __graft_entry__.py smoke 2>&1 | tail -3
} > gpurun_out/gpu_check.log 2>&1
tail -60 gpurun_out/gpu_check.log
