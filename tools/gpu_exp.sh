#!/bin/bash
# scratch driver for the experiment of the moment (gpurun)
mkdir -p gpurun_out
timeout 500 python tools/exp_ring_variants.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/ringv6.log | tail -12
