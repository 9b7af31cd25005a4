mkdir -p gpurun_out
{ timeout 900 python -m pytest tests/test_gpu_parity.py -x -q 2>&1 | tail -6
  timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>&1 | tail -1 > gpurun_out/b.json; cut -c1-200 gpurun_out/b.json
} > gpurun_out/exp.log 2>&1
cat gpurun_out/exp.log
