#!/bin/bash
# scratch driver for the experiment of the moment (gpurun)
mkdir -p gpurun_out/r04
timeout 900 python -m pytest tests/test_gpu_fp16x3.py -q -x -s 2>&1 | grep -v "^$" | tail -40 > gpurun_out/r04/t_fp16.log
tail -25 gpurun_out/r04/t_fp16.log
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_golden_cfg.py -q -s -k "fp16x3" 2>&1 | grep -E "fp16x3|passed|failed|Error|assert" | tail -60 > gpurun_out/r04/t_gold.log
tail -40 gpurun_out/r04/t_gold.log
for p in bf16x3 fp16x3 bf16x3 fp16x3; do
timeout 300 python bench.py --no-cpu-baseline --no-eager-baseline --single-datapath --no-configs --no-gate --steps 20 --precision $p > gpurun_out/r04/q_$p.json 2> gpurun_out/r04/q_$p.err
python -c "
import json; d=json.loads([l for l in open('gpurun_out/r04/q_$p.json') if l.startswith('{')][-1]); print('$p', round(d['value']), round(d['ms_per_step'],3), round(d['inference_rays_per_s']), {k: round(v['avg_ms'],3) for k,v in d['kernels'].items()})"
done
