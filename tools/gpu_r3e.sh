#!/bin/bash
mkdir -p gpurun_out
for v in - libexp_wg1_39.so -; do python tools/exp_wgrad.py $v --bf16 2>&1 | grep -v amdgpu; done
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -p no:cacheprovider -k "golden or gate or bf16x3 or pack or cache" > gpurun_out/r3e_tests.log 2>&1; echo "pytest rc=$?"
tail -2 gpurun_out/r3e_tests.log
