mkdir -p gpurun_out
{ timeout 300 python -m pytest tests/test_gpu_parity.py -x -q -k "fused_adam" 2>&1 | tail -4
  timeout 400 python tools/exp_converge.py 2>&1 | grep -v amdgpu
} > gpurun_out/exp.log 2>&1
cat gpurun_out/exp.log
