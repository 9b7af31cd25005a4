mkdir -p gpurun_out
{ timeout 600 python -m pytest tests -q -m gpu --tb=line -p no:cacheprovider -n 4 -k "field_forward or field_backward or golden_lego_det" 2>&1 | grep -E "passed|failed|Error" | tail -5
  timeout 200 python tools/exp_fwd3.py - --bwd 2>&1 | grep -v amdgpu; } > gpurun_out/exp.log 2>&1
cat gpurun_out/exp.log
