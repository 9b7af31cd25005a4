#!/bin/bash
# scratch driver for the experiment of the moment (gpurun)
mkdir -p gpurun_out
for i in 1 2; do
timeout 300 python bench.py --no-cpu-baseline --no-eager-baseline --single-datapath --no-configs --no-gate --steps 20 > gpurun_out/q.json 2> gpurun_out/q.err; tail -c 300 gpurun_out/q.err
python -c "
import json; d=json.loads([l for l in open('gpurun_out/q.json') if l.startswith('{')][-1]); print(d['value'], d['ms_per_step'], d['inference']['one_launch_rays_per_s'], d['inference']['chain_of_launches_rays_per_s'])"
done
