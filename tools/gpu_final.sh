mkdir -p gpurun_out
{
  timeout 900 python -m pytest tests/test_gpu_parity.py -x -q 2>&1 | tail -4
  timeout 400 python bench.py --steps 20 --warmup 5 2>&1 | tail -1 > gpurun_out/bench_default.json; cut -c1-250 gpurun_out/bench_default.json
  python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
} > gpurun_out/gpu_final.log 2>&1
cat gpurun_out/gpu_final.log
bash tools/profile.sh > /dev/null 2>&1
grep -E "^\"(void )?nerf::" gpurun_out/profile.log | cut -c1-150 | head -8
cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --kernel-trace --stats -f csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_stats_mixed -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 4 --warmup 2 --no-cpu-baseline --single-datapath --precision mixed > /dev/null 2>&1
cd $GRAFT_REPO_ROOT && head -8 gpurun_out/prof_stats_mixed/bench_kernel_stats.csv | cut -c1-150
