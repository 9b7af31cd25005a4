#!/bin/bash
# scratch driver for the experiment of the moment (gpurun)
mkdir -p gpurun_out
timeout 300 python -m pytest tests/test_gpu_round3.py -q -x -k "img2mse" 2>&1 | tail -4
timeout 900 python -m pytest tests -q -x -m gpu -k "golden or train or smoke or two_ranks or one_call" 2>&1 | tail -4
for i in 1 2; do
timeout 300 python bench.py --no-cpu-baseline --no-eager-baseline --single-datapath --no-configs --no-gate --steps 20 > gpurun_out/q.json 2> gpurun_out/q.err; tail -c 200 gpurun_out/q.err | grep -v amdgpu
python -c "
import json; d=json.loads([l for l in open('gpurun_out/q.json') if l.startswith('{')][-1]); print(round(d['value']), round(d['ms_per_step'],3))"
done
