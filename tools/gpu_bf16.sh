mkdir -p gpurun_out
{
  timeout 600 python -m pytest tests -q -m gpu --tb=short -p no:cacheprovider -k "bf16x3 or field_backward" -rP 2>&1 | grep -vE "^\s*$|amdgpu.ids|^-+$|Captured" | cut -c1-400 | tail -40
  timeout 600 python tools/quick_bench.py 4096 2>&1 | tail -30
} > gpurun_out/gpu_bf16.log 2>&1
tail -80 gpurun_out/gpu_bf16.log
