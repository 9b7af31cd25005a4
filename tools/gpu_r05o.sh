mkdir -p gpurun_out
NERF_WRITE_DIGESTS=gpurun_out/kernel_digests.json timeout 600 python -m pytest tests/test_gpu_digests.py -m gpu -q 2>&1 | tail -3 > gpurun_out/r05o_digests.log
timeout 2400 python -m pytest tests -m gpu -q --deselect tests/test_gpu_digests.py 2>&1 | tail -6 > gpurun_out/r05o_tests.log
timeout 900 python bench.py > gpurun_out/r05o_bench.json 2> gpurun_out/r05o_bench.err
tail -2 gpurun_out/r05o_digests.log; tail -4 gpurun_out/r05o_tests.log; tail -c 200 gpurun_out/r05o_bench.err
