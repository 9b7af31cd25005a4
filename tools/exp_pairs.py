"""Search for CONVERGING (teacher, student) pairs for bench.convergence_table (round 5): 500 fused-Adam steps of
1024 rays per candidate on the headline datapath; a pair qualifies when the held-out PSNR ends >= 35 dB.  Candidates: ("scene", t, s)
students from another scene's weights and ("near", t, eps) students from the teacher's weights perturbed by a relative eps.
    python tools/exp_pairs.py > gpurun_out/exp_pairs.log"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402

dev = torch.device("cuda", 0)
cands = [("scene", 5, 6)] + [("near", t, eps) for t in (0, 1, 2, 3, 7) for eps in (0.02, 0.05, 0.1)] + [("scene", t, s) for t, s in ((5, 2), (5, 9), (2, 3), (3, 4), (9, 5), (4, 6))]
for c in cands:
    try:
        t = bench.convergence_table(dev, 500, seeds=(0,), which=("fp32", "fp16x3"), pairs=(c,), checkpoints=(1, 100, 250))
        d = t["datapaths"]
        print(json.dumps({"pair": c, "fp32": d["fp32"]["psnr_db_per_seed_at_checkpoints"][0], "fp16x3": d["fp16x3"]["psnr_db_per_seed_at_checkpoints"][0]}), flush=True)
    except Exception as e:      # noqa: BLE001
        print(json.dumps({"pair": c, "error": str(e)[:200]}), flush=True)
