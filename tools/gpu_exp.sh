#!/bin/bash
mkdir -p gpurun_out
timeout 900 python bench.py > gpurun_out/check_bench.json 2> gpurun_out/check_bench.err
tail -c 300 gpurun_out/check_bench.err
python - <<'PY'
import json
d=json.loads([l for l in open('gpurun_out/check_bench.json') if l.startswith('{')][-1])
print('value', round(d['value']), round(d['ms_per_step'],3), {k:round(v,2) for k,v in d['speedup_vs_rocm_eager'].items()})
print('eager', {k:v for k,v in d['rocm_eager_baseline'].items() if k in ('train_rays_per_s','infer_rays_per_s','render_only')})
print('reduced', round(d['reduced_inference']['rays_per_s']), d['reduced_inference'].get('render_only'), 'render_only', d['configs']['render_only']['value'], 'errors', d.get('errors'))
PY
