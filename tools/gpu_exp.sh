mkdir -p gpurun_out
timeout 300 python tools/exp_traj.py 2>&1 | grep -v amdgpu > gpurun_out/exp.log
cat gpurun_out/exp.log
