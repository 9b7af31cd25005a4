#!/bin/bash
mkdir -p gpurun_out
timeout 1200 python -m pytest tests -m gpu -q -x -p no:cacheprovider > gpurun_out/r2h_tests.log 2>&1; echo "pytest rc=$?"
tail -6 gpurun_out/r2h_tests.log
timeout 600 python bench.py --no-cpu-baseline > gpurun_out/r2h_bench_lego.json 2> gpurun_out/r2h_bench_lego.err; echo "bench rc=$?"
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r2h_bench_lego.json').read())
print(d['value'], d['ms_per_step'], d['inference_rays_per_s'], d['speedup_vs_rocm_eager'])
print({k:(round(v['avg_ms'],3), round(v['mfma_frac'],3), round(v['hbm_frac'],3)) for k,v in d['kernels'].items()})
PY
timeout 600 python bench.py --strong --steps 3 --warmup 1 --no-cpu-baseline --single-datapath --no-eager-baseline --no-gate > gpurun_out/r2h_bench_strong.json 2> gpurun_out/r2h_bench_strong.err; echo "bench strong rc=$?"; cut -c1-400 gpurun_out/r2h_bench_strong.json; tail -3 gpurun_out/r2h_bench_strong.err
