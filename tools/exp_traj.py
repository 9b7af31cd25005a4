import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import torch
import nerf_oracle as orc
import nerf_pytorch_amd as npa
dev = torch.device("cuda", 0)
Pc, Pf = orc.scene_params(seed=2)
kw = dict(D=8, W=256, input_ch=63, output_ch=5, skips=[4], input_ch_views=27, use_viewdirs=True)
rays = orc.synthetic_rays(256, seed=41).to(dev)
target = torch.rand(256, 3, generator=torch.Generator().manual_seed(9)).to(dev)
for lr in (5e-4, 1e-4):
    for prec in ("fp32", "bf16x3", "mixed"):
        nc, nf = npa.NeRF(**kw).to(dev), npa.NeRF(**kw).to(dev)
        nc.load_state_dict(Pc); nf.load_state_dict(Pf)
        opt = torch.optim.Adam(list(nc.parameters()) + list(nf.parameters()), lr=lr)
        npa.set_precision(prec)
        out_l = []
        for step in range(40):
            opt.zero_grad()
            out = npa.render_rays(rays, nc, None, 64, N_importance=128, network_fine=nf, white_bkgd=True)
            loss = npa.img2mse(out["rgb_map"], target) + npa.img2mse(out["rgb0"], target)
            loss.backward(); opt.step(); out_l.append(round(loss.item(), 5))
        print(lr, prec, out_l, flush=True)
