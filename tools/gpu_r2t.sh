#!/bin/bash
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -p no:cacheprovider -k "bf16 or mixed or operand or golden or gate or workspace or subchunks" -s > gpurun_out/r2t_tests.log 2>&1; echo "pytest rc=$?"
grep -v "amdgpu.ids" gpurun_out/r2t_tests.log | grep -i "operand\|passed\|failed\|error\|bwd max" | tail -20
python bench.py --steps 20 --warmup 5 --single-datapath --no-cpu-baseline --no-eager-baseline > gpurun_out/r2t_bench_bf16.json 2> gpurun_out/r2t_bench_bf16.err; echo "bench rc=$?"
NERF_WGRAD_OPERANDS=fp32 python bench.py --steps 20 --warmup 5 --single-datapath --no-cpu-baseline --no-eager-baseline > gpurun_out/r2t_bench_fp32ops.json 2> gpurun_out/r2t_bench_fp32ops.err; echo "bench rc=$?"
python - <<'PY'
import json
for f in ("r2t_bench_bf16", "r2t_bench_fp32ops"):
    try:
        d = json.loads([l for l in open(f"gpurun_out/{f}.json") if l.startswith("{")][-1])
        print(f, d["value"], d["ms_per_step"], d["roofline"]["kernel"], d["roofline"]["frac"], d["roofline"].get("whole_step_mfma_frac"), d.get("precision_gate"))
        for k, v in d["kernels"].items():
            print("   ", k, round(v["avg_ms"], 3), round(v["mfma_frac"], 3), round(v["hbm_frac"], 3))
    except Exception as e:
        print(f, "ERR", e)
PY
