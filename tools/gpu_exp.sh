#!/bin/bash
mkdir -p gpurun_out
timeout 900 python bench.py > gpurun_out/check_bench.json 2> gpurun_out/check_bench.err
tail -c 300 gpurun_out/check_bench.err
python - <<'PY'
import json
d=json.loads([l for l in open('gpurun_out/check_bench.json') if l.startswith('{')][-1])
print('value', round(d['value']), round(d['ms_per_step'],3), 'eager', round(d['rocm_eager_baseline']['train_rays_per_s']), round(d['rocm_eager_baseline']['infer_rays_per_s']), {k:round(v,2) for k,v in d['speedup_vs_rocm_eager'].items()})
print('reduced', round(d['reduced_inference']['rays_per_s']), d['reduced_inference'].get('render_only'), 'infer', round(d['inference_rays_per_s']), 'render_only', d['configs']['render_only']['value'], d['configs']['render_only']['s_per_frame'], 'errors', d.get('errors'))
print('f32', round(d['other_datapath']['value']))
PY
