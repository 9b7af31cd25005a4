#!/bin/bash
# One gpurun call: environment probe, GPU parity tests, quick timing.  Output lands in gpurun_out/.
mkdir -p gpurun_out
{
  echo "== env"; nproc; rocm-smi --showproductname 2>/dev/null | head -8; ls /root/reference 2>&1 | head -2
  python - <<'PY'
import torch, os
print("torch", torch.__version__, "hip", torch.version.hip, "gpus", torch.cuda.device_count(), torch.cuda.get_device_name(0), "cpus", os.cpu_count())
PY
  echo "== pytest -m gpu"
  timeout 1500 python -m pytest tests -q -m gpu -x --tb=short -p no:cacheprovider 2>&1 | tail -60
  echo "== quick bench"
  timeout 600 python tools/quick_bench.py 4096 2>&1 | tail -40
} > gpurun_out/gpu_check.log 2>&1
tail -100 gpurun_out/gpu_check.log
