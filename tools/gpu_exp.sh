#!/bin/bash
# scratch driver for the experiment of the moment (gpurun)
mkdir -p gpurun_out
for c in lego fern lego fern; do
timeout 300 python bench.py --config $c --no-cpu-baseline --no-eager-baseline --single-datapath --no-configs --no-gate --steps 20 > gpurun_out/q.json 2> gpurun_out/q.err; tail -c 200 gpurun_out/q.err | grep -v amdgpu
python -c "
import json; d=json.loads([l for l in open('gpurun_out/q.json') if l.startswith('{')][-1]); print('$c', round(d['value']), round(d['ms_per_step'],3), {k:round(v['avg_ms'],4) for k,v in d['kernels'].items()})"
done
