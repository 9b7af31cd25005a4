#!/bin/bash
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -q -x -p no:cacheprovider > gpurun_out/r3d_tests.log 2>&1; echo "pytest rc=$?"
grep -v "amdgpu.ids" gpurun_out/r3d_tests.log | tail -2
python bench.py > gpurun_out/r3d_bench_lego.json 2> gpurun_out/r3d_bench_lego.err; echo "bench lego rc=$?"
python bench.py --config fern --no-cpu-baseline > gpurun_out/r3d_bench_fern.json 2> gpurun_out/r3d_bench_fern.err; echo "bench fern rc=$?"
bash tools/profile.sh bf16x3 > gpurun_out/r3d_profile.log 2>&1; echo "profile rc=$?"
python - <<'PY'
import json
for f in ("lego", "fern"):
    try:
        d = json.loads([l for l in open(f"gpurun_out/r3d_bench_{f}.json") if l.startswith("{")][-1])
        print(f, round(d["value"]), round(d["ms_per_step"], 3), d["roofline"]["kernel"], round(d["roofline"]["frac"], 3), d["roofline"].get("whole_step_mfma_frac"), d["roofline"].get("traffic"), d.get("speedup_vs_rocm_eager"), (d.get("cpu_baseline") or {}).get("value"))
        print("   other", {k: (v.get("value") if isinstance(v, dict) and "value" in v else v) for k, v in d.items() if k.startswith(("other", "inference", "rocm", "mixed"))})
        for k, v in d["kernels"].items(): print("     ", k, round(v["avg_ms"], 3), round(v["mfma_frac"], 3), round(v["hbm_frac"], 3))
    except Exception as e:
        print(f, "ERR", e)
PY
