#!/bin/bash
mkdir -p gpurun_out/r04
timeout 900 python -m pytest tests/test_gpu_fp16x3.py tests/test_gpu_parity.py -m gpu -q -s -k "reduced or (gate_psnr and fp16_fp8c)" 2>&1 | grep -E "^E  |reduced class|fp16_fp8c|passed|failed" | cut -c1-400 | tee gpurun_out/r04/t_red.log
timeout 300 python tools/exp_reduced.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r04/reduced_timing.txt
