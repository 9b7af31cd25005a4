#!/bin/bash
mkdir -p gpurun_out
python tools/exp_fwd3.py - --bwd 2>&1 | grep -v amdgpu | grep "fwd16\|dgrad3\|\["
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -p no:cacheprovider -k "backward or operand or golden or mixed or forward or empty or ragged or boundary" > gpurun_out/r3b_tests.log 2>&1; echo "pytest rc=$?"
tail -2 gpurun_out/r3b_tests.log
python bench.py --steps 20 --warmup 5 --single-datapath --no-cpu-baseline --no-eager-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print(d['value'], d['ms_per_step'])
for k,v in d['kernels'].items(): print('   ',k,round(v['avg_ms'],3))"
