#!/bin/bash
# round 6, VERDICT r5 item 2: the save path's ingredients on ONE box, builds alternated (boxes differ by 2-4 %).  Build first (CPU):
#   for k in 1 2 3; do tools/build_variant.sh abl$k -DNERF_ABL_SAVE=$k; done
mkdir -p gpurun_out
V=nerf-pytorch_amd/build/variants
for i in 1 2; do
  for lib in nerf-pytorch_amd/libnerf_hip.so $V/libnerf_hip_abl1.so $V/libnerf_hip_abl2.so $V/libnerf_hip_abl3.so; do
    NERF_HIP_LIB=$lib timeout 120 python tools/exp_save_ablation.py "$@" 2>&1 | tail -1
  done
done | tee gpurun_out/r06_save_ablation.log
