mkdir -p gpurun_out
{ timeout 300 python tools/bisect.py 2>&1 | grep -v amdgpu.ids | cut -c1-1500; timeout 300 python tools/bisect.py libnerf_hip_v_nowg256.so 2>&1 | grep -v amdgpu.ids | cut -c1-600; } > gpurun_out/bisect.log 2>&1
tail -30 gpurun_out/bisect.log
