#!/bin/bash
mkdir -p gpurun_out
STEPS=1500 python tools/exp_converge.py 2>&1 | grep -v amdgpu | tee gpurun_out/r3g_converge.log
