#!/bin/bash
mkdir -p gpurun_out/r04
timeout 900 python -m pytest tests/test_gpu_fp16x3.py tests/test_gpu_parity.py -m gpu -q -k "reduced or guard or gate or ring_forward or fused or infer" 2>&1 | tail -5
timeout 300 python tools/exp_reduced.py 2>&1 | tail -5
