#!/bin/bash
mkdir -p gpurun_out
timeout 300 python tools/probe/ack_probe.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r2m_ack_probe.log
