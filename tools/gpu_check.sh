#!/bin/bash
# full GPU verification (gpurun): the -m gpu test suite, smoke(), the default bench line, and (with "profile") the rocprofv3
# kernel stats + PMC passes of tools/profile.sh for the headline datapath
mkdir -p gpurun_out
timeout 2400 python -m pytest tests -m gpu -q 2>&1 | tail -40 > gpurun_out/check_tests.log
timeout 200 python __graft_entry__.py smoke > gpurun_out/check_smoke.log 2>&1
timeout 900 python bench.py > gpurun_out/check_bench.json 2> gpurun_out/check_bench.err
tail -6 gpurun_out/check_tests.log; tail -3 gpurun_out/check_smoke.log; tail -c 600 gpurun_out/check_bench.err; head -c 1500 gpurun_out/check_bench.json
if [ "$1" == "profile" ]; then bash tools/profile.sh fp16x3 > gpurun_out/check_profile.log 2>&1; tail -5 gpurun_out/check_profile.log; fi
