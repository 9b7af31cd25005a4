mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_fp16x3.py tests/test_gpu_round3.py -m gpu -q -x 2>&1 | tail -5 > gpurun_out/r05f_tests.log
R=$(pwd)
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -f csv -d $R/gpurun_out/r05f_prof -o bench -- python $R/bench.py --steps 12 --warmup 3 --no-cpu-baseline --no-eager-baseline --no-gate --single-datapath --no-configs --sustained-s 0 > $R/gpurun_out/r05f_prof.log 2>&1
cd $R
T=$(find gpurun_out/r05f_prof -name "*kernel_trace.csv" | head -1)
python tools/gaps.py $T > gpurun_out/r05f_gaps.txt 2>&1
tail -3 gpurun_out/r05f_tests.log; head -60 gpurun_out/r05f_gaps.txt
