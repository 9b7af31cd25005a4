mkdir -p gpurun_out
{ timeout 200 python tools/exp_fwd3.py - 2>&1 | grep -v amdgpu
  timeout 900 python -m pytest tests/test_gpu_parity.py -x -q 2>&1 | tail -6
  timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --single-datapath --mode infer 2>&1 | tail -1 | cut -c1-200
} > gpurun_out/exp.log 2>&1
cat gpurun_out/exp.log
