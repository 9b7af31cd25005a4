#!/bin/bash
# round-6 evidence run (gpurun): the whole -m gpu suite, smoke(), the default bench line, rocprofv3 stats + PMC of the headline
# datapath, the two-word datapath's line, the CPU baseline on the metric's own shape
mkdir -p gpurun_out
timeout 3000 python -m pytest tests -m gpu -q 2>&1 | tail -40 > gpurun_out/r06_gpu_tests.log; tail -5 gpurun_out/r06_gpu_tests.log
timeout 200 python __graft_entry__.py smoke > gpurun_out/r06_smoke.log 2>&1; tail -2 gpurun_out/r06_smoke.log
timeout 900 python bench.py > gpurun_out/r06_bench_default.json 2> gpurun_out/r06_bench_default.err; head -c 700 gpurun_out/r06_bench_default.json; echo
timeout 600 python bench.py --precision fp16x3w --no-cpu-baseline --single-datapath --no-configs --sustained-s 3 --no-training-gate > gpurun_out/r06_bench_fp16x3w.json 2> gpurun_out/r06_bench_fp16x3w.err; head -c 400 gpurun_out/r06_bench_fp16x3w.json; echo
bash tools/profile.sh fp16x3 > gpurun_out/r06_profile.log 2>&1; tail -3 gpurun_out/r06_profile.log
TAG=fp16x3w bash tools/profile.sh fp16x3w > gpurun_out/r06_profile_w.log 2>&1; tail -3 gpurun_out/r06_profile_w.log
timeout 600 python bench.py --cpu-full > gpurun_out/r06_cpu_full_shape.json 2> gpurun_out/r06_cpu_full.err; cat gpurun_out/r06_cpu_full_shape.json
