mkdir -p gpurun_out
V=nerf-pytorch_amd/build/variants
for i in 1 2 3; do
  for lib in nerf-pytorch_amd/libnerf_hip.so $V/libnerf_hip_s5.so; do
    NERF_HIP_LIB=$lib python tools/time_kernels.py --only wgrad_gemm 2>&1 | tail -1
  done
done > gpurun_out/r05u_stages.log
cat gpurun_out/r05u_stages.log
