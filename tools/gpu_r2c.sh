#!/bin/bash
mkdir -p gpurun_out
python tools/probe/wr_probe.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r2c_wr_probe.log; cat gpurun_out/r2c_wr_probe.log
timeout 900 python -m pytest tests/test_train_loop_gpu.py -m gpu -q -s -p no:cacheprovider 2>&1 | grep -E "first-step|blank|passed|failed|Error" | cut -c1-300
