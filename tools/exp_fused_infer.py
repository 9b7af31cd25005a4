"""render_rays without gradients: the chain of six launches against the one-launch kernel (csrc/render_fused.hip), wall time per
batch for several ray counts (split-bf16 datapath, 64 + 128 samples)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import workloads as wl
import nerf_pytorch_amd as npa

hb = npa.hip_backend
dev = torch.device("cuda", 0)
Pc, Pf = wl.scene_params()
kw = dict(D=8, W=256, input_ch=63, output_ch=5, skips=[4], input_ch_views=27, use_viewdirs=True)
nc, nf = npa.NeRF(**kw).to(dev), npa.NeRF(**kw).to(dev)
nc.load_state_dict(Pc)
nf.load_state_dict(Pf)
npa.set_precision("fp16x3")
for n in (1024, 4096, 5000, 32768):
    rays = wl.synthetic_rays(n, seed=1).to(dev)
    res = {}
    for rep in range(2):
        for one in (False, True):
            hb.INFER_ONE_LAUNCH = one
            with torch.no_grad():
                for _ in range(2):
                    npa.render_rays(rays, nc, None, 64, N_importance=128, network_fine=nf, white_bkgd=True)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                k = 10 if n <= 5000 else 3
                e0.record()
                for _ in range(k):
                    npa.render_rays(rays, nc, None, 64, N_importance=128, network_fine=nf, white_bkgd=True)
                e1.record()
                torch.cuda.synchronize()
            res.setdefault(one, []).append(e0.elapsed_time(e1) / k)
    a, b = min(res[False]), min(res[True])
    print(f"n_rays {n:6d}: chain of launches {a:8.3f} ms ({n / a:7.1f} k rays/s)   one launch {b:8.3f} ms ({n / b:7.1f} k rays/s)   ratio {b / a:.3f}", flush=True)
