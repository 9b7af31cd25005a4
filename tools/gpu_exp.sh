#!/bin/bash
# scratch driver for the experiment of the moment (gpurun)
mkdir -p gpurun_out
for R in 4 8 16; do
echo "--- $R rays per workgroup"; NERF_FUSED_RAYS=$R timeout 200 python tools/exp_fused_infer.py 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/fused.log | tail -8
NERF_FUSED_RAYS=$R timeout 300 python -m pytest tests/test_gpu_round3.py -q -x -k "one_launch" 2>&1 | tail -2
done
