#!/bin/bash
mkdir -p gpurun_out
{
for lib in libexp_head.so libnerf_hip.so libexp_head.so libnerf_hip.so; do
  echo "== $lib"; timeout 300 python tools/exp_fwd3.py $lib --bwd 2>&1 | grep -v amdgpu.ids | grep -E "wgrad3|^=="
done
} > gpurun_out/r2p_exp.log 2>&1
cat gpurun_out/r2p_exp.log
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -p no:cacheprovider -k "backward or golden or ragged or large_chunks" > gpurun_out/r2p_tests.log 2>&1; echo "pytest rc=$?"
tail -3 gpurun_out/r2p_tests.log
