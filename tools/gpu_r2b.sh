#!/bin/bash
# round 2, run b: kernel timing probes (variants built by tools/build_variants.py) + full GPU suite on the new default
mkdir -p gpurun_out
{
for lib in libnerf_hip.so libexp_oldwg.so libexp_nomask.so libexp_norows.so libexp_dgrad_nostore.so; do
  echo "== $lib"; timeout 300 python tools/exp_fwd3.py $lib --bwd 2>&1 | grep -v amdgpu.ids
done
} > gpurun_out/r2b_exp.log 2>&1
cat gpurun_out/r2b_exp.log
timeout 1200 python -m pytest tests -m gpu -q -x -p no:cacheprovider > gpurun_out/r2b_tests.log 2>&1; echo "pytest rc=$?"
tail -4 gpurun_out/r2b_tests.log
grep -n "held-out\|first-step" gpurun_out/r2b_tests.log
