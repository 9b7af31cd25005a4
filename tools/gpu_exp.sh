#!/bin/bash
mkdir -p gpurun_out/r04
timeout 2400 python -m pytest tests -m gpu -q 2>&1 | tail -60 > gpurun_out/check_tests.log; tail -25 gpurun_out/check_tests.log
timeout 200 python __graft_entry__.py smoke 2>&1 | tail -4
timeout 900 python bench.py > gpurun_out/check_bench.json 2> gpurun_out/check_bench.err; tail -c 300 gpurun_out/check_bench.err; python -c "
import json
d=json.loads([l for l in open('gpurun_out/check_bench.json') if l.startswith('{')][-1])
print(d['value'], d['ms_per_step'], d.get('sustained_rays_per_s'), d.get('errors'), d['speedup_vs_rocm_eager'])
print(d['precision_gate'].get('gradient'), d['roofline'].get('traffic'))
"
