mkdir -p gpurun_out
V=nerf-pytorch_amd/build/variants
for i in 1 2 3; do
  for lib in nerf-pytorch_amd/libnerf_hip.so $V/libnerf_hip_nomerge.so; do
    NERF_HIP_LIB=$lib python tools/time_kernels.py --only wgrad_gemm,wgrad_reduce 2>/dev/null | tail -1
  done
done > gpurun_out/r05n_merge.log
timeout 1200 python -m pytest tests -m gpu -q -k "(grad or backward or bwd or golden or train or one_call) and not digest" 2>&1 | tail -8 > gpurun_out/r05n_tests.log
cat gpurun_out/r05n_merge.log; tail -6 gpurun_out/r05n_tests.log
