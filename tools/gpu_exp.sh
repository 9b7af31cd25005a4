#!/bin/bash
mkdir -p gpurun_out/r04
timeout 900 python -m pytest tests/test_gpu_fp16x3.py tests/test_gpu_dense.py -m gpu -q 2>&1 | tail -5
timeout 600 python bench.py --no-cpu-baseline > gpurun_out/r04/full.json 2> gpurun_out/r04/full.err
python - <<'PY'
import json
d=json.loads([l for l in open('gpurun_out/r04/full.json') if l.startswith('{')][-1])
print('value', round(d['value']), round(d['ms_per_step'],3), 'infer', round(d['inference_rays_per_s']))
print({k:(round(v['value']), v.get('ms_per_step')) for k,v in d['configs'].items()})
print('errors', d.get('errors'))
print({k: round(v['avg_ms'],4) for k,v in d['kernels'].items()})
PY
tail -5 gpurun_out/r04/full.err
