#!/bin/bash
# round-end evidence (gpurun): gpu_check.sh with the rocprofv3 passes, then the training-equivalence table (bench.py --long) and a
# 60-s sustained leg; everything lands under gpurun_out/ and is copied into profiles/ by hand
bash tools/gpu_check.sh profile
timeout 1800 python bench.py --long --no-cpu-baseline --no-eager-baseline --no-configs --single-datapath --sustained-s 0 > gpurun_out/long.json 2> gpurun_out/long.err
python - <<'PY'
import json
d = json.loads([l for l in open('gpurun_out/long.json') if l.startswith('{')][-1])
t = d['precision_gate']['training']
print('long:', {k: (v['psnr_db_per_seed'], round(v['max_abs_diff_to_fp32_db'], 3)) for k, v in t['datapaths'].items()}, t.get('families'))
PY
timeout 400 python bench.py --no-cpu-baseline --no-eager-baseline --single-datapath --no-training-gate --sustained-s 60 > gpurun_out/sustained60.json 2> gpurun_out/sustained60.err
python - <<'PY'
import json
d = json.loads([l for l in open('gpurun_out/sustained60.json') if l.startswith('{')][-1])
s = d['sustained']['train']
print('sustained 60 s:', round(s['rays_per_s']), s['rays_per_s_by_second'][:3], '...', s['rays_per_s_by_second'][-3:], d['power'])
PY
