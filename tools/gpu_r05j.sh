mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "subchunks or subset_of_the_network or fused_adam" 2>&1 | tail -8 > gpurun_out/r05j_tests.log
timeout 300 python bench.py --gpus 1 --backend nccl --force-group --steps 5 --warmup 2 --single-datapath --no-gate --no-cpu-baseline --no-eager-baseline --no-configs --sustained-s 0 > gpurun_out/r05j_forcegroup.json 2> gpurun_out/r05j_forcegroup.err
timeout 900 python bench.py > gpurun_out/r05j_bench.json 2> gpurun_out/r05j_bench.err
timeout 400 python bench.py --no-cpu-baseline --no-eager-baseline --single-datapath --no-training-gate --no-configs --sustained-s 60 > gpurun_out/r05j_sustained60.json 2> gpurun_out/r05j_sustained60.err
timeout 1800 python bench.py --long --no-cpu-baseline --no-eager-baseline --no-configs --single-datapath --sustained-s 0 > gpurun_out/r05j_long.json 2> gpurun_out/r05j_long.err
tail -4 gpurun_out/r05j_tests.log; tail -c 300 gpurun_out/r05j_forcegroup.err; tail -c 200 gpurun_out/r05j_bench.err; tail -c 200 gpurun_out/r05j_long.err
