mkdir -p gpurun_out
{
  echo "== modes"
  timeout 300 python bench.py --mode infer --steps 10 --warmup 3 --no-cpu-baseline --single-datapath 2>&1 | tail -1 | cut -c1-330
  timeout 300 python bench.py --precision fp32 --steps 6 --warmup 2 --no-cpu-baseline --single-datapath 2>&1 | tail -1 | cut -c1-330
} > gpurun_out/gpu_final.log 2>&1
cat gpurun_out/gpu_final.log
bash tools/profile.sh > /dev/null 2>&1
grep -E "^\"(void )?nerf::" gpurun_out/profile.log | cut -c1-150 | head -12
