mkdir -p gpurun_out
V=nerf-pytorch_amd/build/variants
for i in 1 2; do
  for lib in $V/libnerf_hip_baluni.so $V/libnerf_hip_nobal.so; do
    NERF_HIP_LIB=$lib python tools/time_kernels.py --only wgrad_gemm 2>/dev/null | tail -1
  done
done > gpurun_out/r05t.log
cat gpurun_out/r05t.log
