mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_golden_cfg.py tests/test_gpu_parity.py -m gpu -q -s -k "golden" 2>&1 | grep -E "^\.?\w+ (fp32|fp16x3|bf16x3) \{|passed|failed" > gpurun_out/r05h_golden.log
timeout 900 python bench.py > gpurun_out/r05h_bench.json 2> gpurun_out/r05h_bench.err
timeout 1500 python bench.py --long --no-cpu-baseline --no-eager-baseline --no-configs --single-datapath --sustained-s 0 > gpurun_out/r05h_long.json 2> gpurun_out/r05h_long.err
tail -2 gpurun_out/r05h_golden.log; tail -c 300 gpurun_out/r05h_bench.err; tail -c 300 gpurun_out/r05h_long.err
