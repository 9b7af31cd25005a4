#!/bin/bash
mkdir -p gpurun_out
timeout 1200 python -m pytest tests -m gpu -q -x -p no:cacheprovider > gpurun_out/r2n_tests.log 2>&1; echo "pytest rc=$?"
tail -6 gpurun_out/r2n_tests.log
timeout 300 python tools/exp_fwd3.py libnerf_hip.so --bwd 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r2n_exp.log
timeout 600 python bench.py --no-cpu-baseline > gpurun_out/r2n_bench_lego.json 2> gpurun_out/r2n_bench_lego.err; echo "bench rc=$?"
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r2n_bench_lego.json').read())
print(d['value'], d['ms_per_step'], d['inference_rays_per_s'], d['speedup_vs_rocm_eager'], d['precision_gate']['psnr_delta_db'], d['precision_gate']['psnr_vs_ref_db'])
print({k:(round(v['avg_ms'],3), round(v['mfma_frac'],3), round(v['hbm_frac'],3)) for k,v in d['kernels'].items()})
print('mixed', d['mixed_precision_training']['value'], 'fp32', d['other_datapath']['value'])
PY
