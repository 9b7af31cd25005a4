mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_golden_cfg.py tests/test_gpu_parity.py -m gpu -q -s -k "reduced_inference_class" 2>&1 | grep -E "fp16_fp8c \{|passed|failed" > gpurun_out/r05d_reduced.log
timeout 600 python bench.py --no-cpu-baseline --no-eager-baseline --single-datapath --no-training-gate --sustained-s 0 > gpurun_out/r05d_bench.json 2> gpurun_out/r05d_bench.err
timeout 900 python tools/exp_pairs.py > gpurun_out/exp_pairs.log 2>&1
cat gpurun_out/r05d_reduced.log; tail -c 300 gpurun_out/r05d_bench.err; cat gpurun_out/exp_pairs.log | tail -30
