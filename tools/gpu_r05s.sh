mkdir -p gpurun_out
V=nerf-pytorch_amd/build/variants
timeout 1200 python -m pytest tests -m gpu -q -x -k "(grad or backward or bwd or golden or one_call or fuzz) and not digest" 2>&1 | tail -5 > gpurun_out/r05s_tests.log
for i in 1 2 3; do
  for lib in nerf-pytorch_amd/libnerf_hip.so $V/libnerf_hip_nobal.so; do
    NERF_HIP_LIB=$lib python tools/time_kernels.py --only wgrad_gemm,wgrad_reduce 2>/dev/null | tail -1
  done
done > gpurun_out/r05s_balance.log
tail -3 gpurun_out/r05s_tests.log; cat gpurun_out/r05s_balance.log
