mkdir -p gpurun_out
{ timeout 900 python -m pytest tests/test_gpu_parity.py -x -q 2>&1 | tail -8
  timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --single-datapath --mode infer 2>&1 | tail -1 | cut -c1-200
} > gpurun_out/exp.log 2>&1
cat gpurun_out/exp.log
