#!/bin/bash
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -p no:cacheprovider -k "not bf16 and not mixed and not operand" > gpurun_out/r3i_tests.log 2>&1; echo "pytest rc=$?"
grep -v amdgpu gpurun_out/r3i_tests.log | tail -3
python bench.py --precision fp32 --steps 8 --warmup 3 --single-datapath --no-cpu-baseline --no-eager-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print(d['value'], d['ms_per_step'], d['precision_gate']['psnr_delta_db'], d['precision_gate']['psnr_vs_ref_db'])
for k,v in d['kernels'].items(): print('   ',k,round(v['avg_ms'],3), round(v['mfma_frac'],3), round(v['hbm_frac'],3))"
