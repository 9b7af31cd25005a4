#!/bin/bash
mkdir -p gpurun_out/r04
timeout 1200 python -m pytest tests/test_gpu_fp16x3.py tests/test_gpu_golden_cfg.py -m gpu -q -s 2>&1 | grep -E "^E  |datapath:|backward vs fp64|upstream gradient|passed|failed|cfg4" | cut -c1-700 > gpurun_out/r04/t_fail.log; cat gpurun_out/r04/t_fail.log
