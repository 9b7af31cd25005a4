#!/bin/bash
mkdir -p gpurun_out/r04
timeout 900 python -m pytest tests/test_gpu_fp16x3.py tests/test_gpu_round3.py -m gpu -q 2>&1 | tail -4
timeout 300 python tools/exp_reduced.py 2>&1 | tail -14
