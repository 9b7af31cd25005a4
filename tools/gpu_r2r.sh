#!/bin/bash
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_two_ranks_gpu.py -m gpu -q -x -p no:cacheprovider > gpurun_out/r2r_tests.log 2>&1; echo "pytest rc=$?"
tail -15 gpurun_out/r2r_tests.log
