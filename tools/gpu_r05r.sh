mkdir -p gpurun_out
bash tools/profile.sh fp16x3 > gpurun_out/r05r_profile.log 2>&1
timeout 900 python bench.py > gpurun_out/r05r_bench.json 2> gpurun_out/r05r_bench.err
tail -3 gpurun_out/r05r_profile.log; tail -c 200 gpurun_out/r05r_bench.err
