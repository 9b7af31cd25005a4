mkdir -p gpurun_out
timeout 300 python tools/exp_render_path.py 2>&1 | grep -v amdgpu > gpurun_out/exp.log
cat gpurun_out/exp.log
