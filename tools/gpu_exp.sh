mkdir -p gpurun_out
{ for v in - libexp_sched.so; do timeout 200 python tools/exp_fwd3.py $v --bwd 2>&1 | grep -v amdgpu; done; } > gpurun_out/exp.log 2>&1
cat gpurun_out/exp.log
