mkdir -p gpurun_out
V=nerf-pytorch_amd/build/variants
for i in 1 2 3; do
  for lib in nerf-pytorch_amd/libnerf_hip.so $V/libnerf_hip_nomerge.so; do
    NERF_HIP_LIB=$lib python bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-eager-baseline --no-gate --single-datapath --no-configs --sustained-s 0 2>/dev/null | python -c "
import sys,json
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('$lib', round(d['value']), round(d['ms_per_step'],4), {k:round(v['avg_ms'],4) for k,v in d['kernels'].items()})"
  done
done > gpurun_out/r05p_step_ab.log
cat gpurun_out/r05p_step_ab.log
