mkdir -p gpurun_out
{
  timeout 900 python -m pytest tests/test_gpu_parity.py -x -q 2>&1 | tail -4
  timeout 400 python bench.py --steps 20 --warmup 5 2>&1 | tail -1 > gpurun_out/bench_default.json; cut -c1-250 gpurun_out/bench_default.json
  python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
} > gpurun_out/gpu_final.log 2>&1
cat gpurun_out/gpu_final.log
bash tools/profile.sh > /dev/null 2>&1
grep -E "^\"(void )?nerf::" gpurun_out/profile.log | cut -c1-150 | head -12
