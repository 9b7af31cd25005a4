#!/bin/bash
mkdir -p gpurun_out
L=gpurun_out/r2s.log; : > $L
python tools/exp_wgrad.py - >> $L 2>&1
for c in 64 96; do NERF_WGRAD_CHUNKS=$c python tools/exp_wgrad.py - >> $L 2>&1; done
for v in wg_anti wg_anti_prio wg_prio_mfma wg_prio_stage; do python tools/exp_wgrad.py libexp_$v.so >> $L 2>&1; done
python tools/exp_wgrad.py - >> $L 2>&1
cat $L
timeout 600 python -m pytest tests/test_two_ranks_gpu.py -m gpu -q -x -p no:cacheprovider > gpurun_out/r2s_tests.log 2>&1; echo "pytest rc=$?"
tail -5 gpurun_out/r2s_tests.log
