#!/bin/bash
mkdir -p gpurun_out/r04
timeout 300 python tools/exp_power.py > gpurun_out/r04/power.txt 2>&1; tail -40 gpurun_out/r04/power.txt
