#!/bin/bash
# round 2, first GPU pass: full -m gpu suite (no -x, -s for the calibration reports), smoke, bench lines
mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r2a_smoke.log 2>&1; echo "smoke rc=$?"
timeout 1200 python -m pytest tests -m gpu -q -s -p no:cacheprovider > gpurun_out/r2a_tests.log 2>&1; echo "pytest rc=$?"
tail -5 gpurun_out/r2a_tests.log
timeout 600 python bench.py > gpurun_out/r2a_bench_lego.json 2> gpurun_out/r2a_bench_lego.err; echo "bench lego rc=$?"
timeout 600 python bench.py --config fern --no-cpu-baseline > gpurun_out/r2a_bench_fern.json 2> gpurun_out/r2a_bench_fern.err; echo "bench fern rc=$?"
timeout 600 python bench.py --mode render_only --steps 4 --warmup 1 --no-cpu-baseline > gpurun_out/r2a_bench_render.json 2> gpurun_out/r2a_bench_render.err; echo "bench render_only rc=$?"
python bench.py --gpus 2 --steps 1 > gpurun_out/r2a_bench_2gpu.log 2>&1; echo "bench --gpus 2 on a 1-GPU box rc=$? (expected 2)"
cut -c1-1500 gpurun_out/r2a_bench_lego.json
