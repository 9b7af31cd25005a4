#!/bin/bash
mkdir -p gpurun_out
python tools/exp_fwd3.py - --bwd 2>&1 | grep -v amdgpu | grep "fwd16\|dgrad3\|\["
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -p no:cacheprovider -k "forward or backward or operand or golden or mixed" > gpurun_out/r3j_tests.log 2>&1; echo "pytest rc=$?"
tail -2 gpurun_out/r3j_tests.log
