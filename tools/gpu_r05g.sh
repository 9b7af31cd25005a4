mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_fp16x3.py -m gpu -q -x 2>&1 | tail -3 > gpurun_out/r05g_tests.log
timeout 900 python -m pytest tests/test_gpu_golden_cfg.py tests/test_gpu_parity.py -m gpu -q -s -k "golden and fp16x3" 2>&1 | grep -E "fp16x3 \{|passed|failed" > gpurun_out/r05g_golden_fp16x3.log
timeout 900 python bench.py > gpurun_out/r05g_bench.json 2> gpurun_out/r05g_bench.err
timeout 900 python bench.py --long --no-cpu-baseline --no-eager-baseline --no-configs --single-datapath --sustained-s 0 > gpurun_out/r05g_long.json 2> gpurun_out/r05g_long.err
tail -3 gpurun_out/r05g_tests.log; cut -c1-400 gpurun_out/r05g_golden_fp16x3.log; tail -c 300 gpurun_out/r05g_bench.err; tail -c 300 gpurun_out/r05g_long.err
