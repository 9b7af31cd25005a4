mkdir -p gpurun_out
{ timeout 600 python -m pytest tests -x -q -m gpu 2>&1 | tail -3
  python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
  timeout 400 python bench.py 2>&1 | tail -1 > gpurun_out/bench_default.json; cut -c1-300 gpurun_out/bench_default.json
} > gpurun_out/exp.log 2>&1
cat gpurun_out/exp.log
