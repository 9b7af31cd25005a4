mkdir -p gpurun_out
timeout 2400 python -m pytest tests -m gpu -q -x 2>&1 | tail -8 > gpurun_out/r05k_tests.log
timeout 200 python __graft_entry__.py smoke > gpurun_out/r05k_smoke.log 2>&1
timeout 900 python bench.py > gpurun_out/r05k_bench.json 2> gpurun_out/r05k_bench.err
timeout 400 python bench.py --no-cpu-baseline --no-eager-baseline --single-datapath --no-training-gate --sustained-s 60 > gpurun_out/r05k_sustained60.json 2> gpurun_out/r05k_sustained60.err
tail -3 gpurun_out/r05k_tests.log; tail -3 gpurun_out/r05k_smoke.log; tail -c 200 gpurun_out/r05k_bench.err; tail -c 200 gpurun_out/r05k_sustained60.err
