#!/bin/bash
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -q -x -p no:cacheprovider > gpurun_out/r3f_tests.log 2>&1; echo "pytest rc=$?"
grep -v "amdgpu.ids" gpurun_out/r3f_tests.log | tail -2
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | grep -v amdgpu | tail -1
python bench.py > gpurun_out/r3f_bench_lego.json 2> gpurun_out/r3f_bench_lego.err; echo "bench lego rc=$?"
python bench.py --config fern --no-cpu-baseline > gpurun_out/r3f_bench_fern.json 2> gpurun_out/r3f_bench_fern.err; echo "bench fern rc=$?"
python bench.py --mode render_only --steps 5 --warmup 1 --no-cpu-baseline > gpurun_out/r3f_bench_render.json 2> gpurun_out/r3f_bench_render.err; echo "bench render rc=$?"
bash tools/profile.sh bf16x3 > gpurun_out/r3f_profile_bf16x3.log 2>&1; echo "profile bf16x3 rc=$?"
bash tools/profile.sh fp32 > gpurun_out/r3f_profile_fp32.log 2>&1; echo "profile fp32 rc=$?"
bash tools/profile.sh mixed > gpurun_out/r3f_profile_mixed.log 2>&1; echo "profile mixed rc=$?"
python - <<'PY'
import json
for f in ("lego", "fern", "render"):
    try:
        d = json.loads([l for l in open(f"gpurun_out/r3f_bench_{f}.json") if l.startswith("{")][-1])
        print(f, round(d["value"]), round(d["ms_per_step"], 3), d["roofline"]["kernel"], round(d["roofline"]["frac"], 3), d["roofline"].get("whole_step_mfma_frac"), d["roofline"].get("traffic"), d.get("speedup_vs_rocm_eager"), (d.get("cpu_baseline") or {}).get("value"))
        print("   other", {k: (v.get("value") if isinstance(v, dict) and "value" in v else v) for k, v in d.items() if k.startswith(("other", "inference", "rocm", "mixed"))})
        for k, v in d["kernels"].items(): print("     ", k, round(v["avg_ms"], 3), round(v["mfma_frac"], 3), round(v["hbm_frac"], 3))
    except Exception as e:
        print(f, "ERR", e)
PY
