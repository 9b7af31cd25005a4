#!/bin/bash
mkdir -p gpurun_out
python tools/exp_wgrad.py - --bf16 2>&1 | grep -v amdgpu
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -p no:cacheprovider -k "backward or operand or golden or mixed" > gpurun_out/r3c_tests.log 2>&1; echo "pytest rc=$?"
tail -2 gpurun_out/r3c_tests.log
python tools/exp_wgrad.py - --bf16 2>&1 | grep -v amdgpu
