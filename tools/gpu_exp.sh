mkdir -p gpurun_out
{ timeout 100 python -m pytest tests -x -q -m gpu 2>&1 | tail -3
  timeout 60 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --single-datapath 2>&1 | tail -1 > gpurun_out/b.json; cut -c1-200 gpurun_out/b.json
} > gpurun_out/exp.log 2>&1
cat gpurun_out/exp.log
