mkdir -p gpurun_out
V=nerf-pytorch_amd/build/variants
for i in 1 2 3; do
  for lib in nerf-pytorch_amd/libnerf_hip.so $V/libnerf_hip_novj.so $V/libnerf_hip_vj25.so; do
    NERF_HIP_LIB=$lib python tools/time_kernels.py --only wgrad_gemm,wgrad_reduce 2>/dev/null | tail -1
  done
done > gpurun_out/r05m_vjobs.log
timeout 1200 python -m pytest tests -m gpu -q -x -k "grad or backward or bwd or golden or digest or train or one_call" 2>&1 | tail -6 > gpurun_out/r05m_tests.log
cat gpurun_out/r05m_vjobs.log; tail -4 gpurun_out/r05m_tests.log
