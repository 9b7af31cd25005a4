mkdir -p gpurun_out
bash tools/profile.sh > /dev/null 2>&1
grep -E "^\"(void )?nerf::" gpurun_out/profile.log | cut -c1-150 | head -6
cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --kernel-trace --stats -f csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_stats_mixed -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 4 --warmup 2 --no-cpu-baseline --single-datapath --precision mixed > /dev/null 2>&1
cd $GRAFT_REPO_ROOT && head -6 gpurun_out/prof_stats_mixed/bench_kernel_stats.csv | cut -c1-150
