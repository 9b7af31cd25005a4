#!/bin/bash
mkdir -p gpurun_out
python tools/exp_wgrad.py - --bf16 2>&1 | grep -v amdgpu
python tools/exp_wgrad.py - 2>&1 | grep -v amdgpu
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -p no:cacheprovider -k "backward or operand or golden" > gpurun_out/r2z_tests.log 2>&1; echo "pytest rc=$?"
tail -2 gpurun_out/r2z_tests.log
