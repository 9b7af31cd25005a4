#!/bin/bash
bash tools/profile.sh bf16x3 2>&1 | tail -14
timeout 600 python bench.py > gpurun_out/r2o_bench_lego.json 2> gpurun_out/r2o_bench_lego.err; echo "bench lego rc=$?"
timeout 600 python bench.py --config fern --no-cpu-baseline > gpurun_out/r2o_bench_fern.json 2> gpurun_out/r2o_bench_fern.err; echo "bench fern rc=$?"
timeout 600 python bench.py --mode render_only --steps 5 --warmup 1 --no-cpu-baseline > gpurun_out/r2o_bench_render.json 2> gpurun_out/r2o_bench_render.err; echo "bench render_only rc=$?"
python - <<'PY'
import json
for f in ("lego","fern","render"):
    d=json.loads(open(f'gpurun_out/r2o_bench_{f}.json').read())
    print(f, round(d['value']), round(d['ms_per_step'],3), d.get('inference_rays_per_s'), d.get('speedup_vs_rocm_eager'), (d.get('other_datapath') or {}).get('value'), (d.get('mixed_precision_training') or {}).get('value'), (d.get('cpu_baseline') or {}).get('value'), (d.get('rocm_eager_baseline') or {}).get('train_rays_per_s'), (d.get('rocm_eager_baseline') or {}).get('infer_rays_per_s'))
    print('   gate', {k: (round(v,5) if isinstance(v,float) else v) for k,v in (d.get('precision_gate') or {}).items() if k!='what'})
    print('   roof', {k: v for k,v in d['roofline'].items() if k in ('kernel','bound','frac','achieved','traffic','whole_step_mfma_frac')} if d.get('roofline') else None)
    print('   kern', {k:(round(v['avg_ms'],3), round(v['mfma_frac'],3), round(v['hbm_frac'],3)) for k,v in d['kernels'].items()})
PY
