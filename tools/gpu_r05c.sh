mkdir -p gpurun_out
NERF_WRITE_DIGESTS=gpurun_out/kernel_digests.json timeout 600 python -m pytest tests/test_gpu_digests.py -m gpu -q 2>&1 | tail -5 > gpurun_out/r05c_digests.log
timeout 900 python -m pytest tests/test_rccl_one_rank_gpu.py tests/test_gpu_golden_cfg.py tests/test_gpu_parity.py -m gpu -q -s -k "rccl or reduced_inference_class" 2>&1 | tail -60 > gpurun_out/r05c_new_tests.log
timeout 900 python bench.py > gpurun_out/r05c_bench.json 2> gpurun_out/r05c_bench.err
tail -3 gpurun_out/r05c_digests.log; tail -30 gpurun_out/r05c_new_tests.log; tail -c 400 gpurun_out/r05c_bench.err
