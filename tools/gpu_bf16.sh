mkdir -p gpurun_out
{
  timeout 900 python -m pytest tests -q -m gpu --tb=short -p no:cacheprovider -k "bf16x3" -rP 2>&1 | grep -vE "^\s*$|amdgpu.ids|^-+$|Captured" | cut -c1-1200 | tail -40
  timeout 600 python bench.py --no-cpu-baseline --precision bf16x3 2>&1 | tail -2
} > gpurun_out/gpu_bf16.log 2>&1
tail -80 gpurun_out/gpu_bf16.log
