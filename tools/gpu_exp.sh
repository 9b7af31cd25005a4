#!/bin/bash
mkdir -p gpurun_out/r04
for p in fp16x3 fp16x3; do
timeout 300 python bench.py --no-cpu-baseline --no-eager-baseline --single-datapath --no-configs --no-gate --steps 20 --precision $p > gpurun_out/r04/q_$p.json 2> gpurun_out/r04/q_$p.err
python -c "
import json; d=json.loads([l for l in open('gpurun_out/r04/q_$p.json') if l.startswith('{')][-1]); print('$p', round(d['value']), round(d['ms_per_step'],3), round(d['inference_rays_per_s'])); print({k: round(v['avg_ms'],4) for k,v in d['kernels'].items()})"
done
