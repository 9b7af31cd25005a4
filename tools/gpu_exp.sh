#!/bin/bash
# scratch driver for the experiment of the moment (gpurun)
mkdir -p gpurun_out
timeout 900 python -m pytest tests -q -x -m gpu -k "backward or golden or operand or one_call or train or two_ranks" 2>&1 | tail -5
timeout 300 python bench.py --no-cpu-baseline --no-eager-baseline --single-datapath --no-configs --steps 20 > gpurun_out/q.json 2> gpurun_out/q.err; tail -c 300 gpurun_out/q.err
python -c "
import json; d=json.loads([l for l in open('gpurun_out/q.json') if l.startswith('{')][-1]); print(d['value'], d['ms_per_step']); print({k:round(v['avg_ms'],4) for k,v in d['kernels'].items()})"
