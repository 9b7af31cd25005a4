#!/bin/bash
# One gpurun call: GPU parity tests, contract bench, rocprofv3 kernel stats.  Output lands in gpurun_out/.
mkdir -p gpurun_out
R=${GRAFT_REPO_ROOT:-$(pwd)}
{
  echo "== pytest -m gpu"
  timeout 1500 python -m pytest tests -q -m gpu --tb=short -p no:cacheprovider -n 4 -rP 2>&1 | grep -vE "^\s*$|amdgpu.ids|Captured|^-+$" | cut -c1-1500 | tail -150
  echo "== bench"
  timeout 900 python bench.py 2>&1 | tail -5
  echo "== smoke"
  timeout 300 python __graft_entry__.py smoke 2>&1 | tail -3
  echo "== rocprofv3"
  cd /tmp && export TMPDIR=/tmp
  timeout 900 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof -o bench -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline 2>&1 | tail -5
  cd $R; find gpurun_out/prof -name "*stats*" | head; for f in $(find gpurun_out/prof -name "*kernel_stats*.csv"); do head -20 $f; done
} > gpurun_out/gpu_check.log 2>&1
tail -230 gpurun_out/gpu_check.log
