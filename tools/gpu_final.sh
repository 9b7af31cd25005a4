mkdir -p gpurun_out
{
  echo "== torchrun world=1"
  timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --steps 5 --warmup 2 --no-cpu-baseline --single-datapath 2>&1 | tail -1 | cut -c1-400
  echo "== default bench"
  timeout 900 python bench.py 2>&1 | tail -1 > gpurun_out/bench_default.json; cut -c1-300 gpurun_out/bench_default.json
  echo "== smoke"; timeout 300 python __graft_entry__.py smoke 2>&1 | tail -1
} > gpurun_out/gpu_final.log 2>&1
tail -20 gpurun_out/gpu_final.log
