mkdir -p gpurun_out
timeout 2400 python -m pytest tests -m gpu -q -x --deselect tests/test_gpu_digests.py 2>&1 | tail -25 > gpurun_out/r05e_tests.log
NERF_WRITE_DIGESTS=gpurun_out/kernel_digests.json timeout 600 python -m pytest tests/test_gpu_digests.py -m gpu -q 2>&1 | tail -5 > gpurun_out/r05e_digests.log
timeout 900 python -m pytest tests/test_gpu_golden_cfg.py tests/test_gpu_parity.py -m gpu -q -s -k "reduced_inference_class" 2>&1 | grep -E "fp16_fp8c \{|passed|failed" > gpurun_out/r05e_reduced.log
timeout 900 python bench.py > gpurun_out/r05e_bench.json 2> gpurun_out/r05e_bench.err
tail -8 gpurun_out/r05e_tests.log; tail -3 gpurun_out/r05e_digests.log; cat gpurun_out/r05e_reduced.log | cut -c1-700; tail -c 300 gpurun_out/r05e_bench.err
