#!/bin/bash
mkdir -p gpurun_out/r04
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py -m gpu -q -k "fp32 or backward or golden or grad or fuzz" 2>&1 | tail -4
timeout 300 python bench.py --no-cpu-baseline --no-eager-baseline --single-datapath --no-configs --no-gate --steps 10 --precision fp32 > gpurun_out/r04/q_fp32.json 2> gpurun_out/r04/q_fp32.err
python -c "
import json; d=json.loads([l for l in open('gpurun_out/r04/q_fp32.json') if l.startswith('{')][-1]); print('fp32', round(d['value']), round(d['ms_per_step'],3)); print({k: round(v['avg_ms'],4) for k,v in d['kernels'].items()})"
