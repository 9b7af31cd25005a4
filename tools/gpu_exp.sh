#!/bin/bash
# scratch driver for the experiment of the moment (gpurun)
mkdir -p gpurun_out
NERF_ALLOW_SHARED_GPU=1 timeout 600 python bench.py --gpus 2 --backend gloo --steps 6 --warmup 2 > gpurun_out/two.json 2> gpurun_out/two.err; tail -c 400 gpurun_out/two.err
python -c "
import json; d=json.loads([l for l in open('gpurun_out/two.json') if l.startswith('{')][-1]); print(d['value'], d['n_gpus'], d['ms_per_step'], d.get('multi_gpu'), d.get('inference'), d.get('errors'))"
NERF_ALLOW_SHARED_GPU=1 timeout 600 python bench.py --gpus 2 --backend gloo --mode render_only --steps 2 --warmup 1 > gpurun_out/two_r.json 2> gpurun_out/two_r.err; tail -c 300 gpurun_out/two_r.err
python -c "
import json; d=json.loads([l for l in open('gpurun_out/two_r.json') if l.startswith('{')][-1]); print(d['value'], d['n_gpus'], d.get('errors'))"
