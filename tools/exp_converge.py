"""Training-equivalence evidence: the same student trained for N Adam steps in each datapath (same data order, same
init), PSNR against a teacher scene on held-out rays.  Prints one line per datapath."""
import os, sys, math
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import torch
import nerf_oracle as orc
import nerf_pytorch_amd as npa
dev = torch.device("cuda", 0)
STEPS, NB = int(os.environ.get("STEPS", 300)), 1024
kw = dict(D=8, W=256, input_ch=63, output_ch=5, skips=[4], input_ch_views=27, use_viewdirs=True)
Tc, Tf = orc.scene_params(seed=5)                       # teacher
Sc, Sf = orc.scene_params(seed=6)                       # student init (a different scene)
def net(P):
    m = npa.NeRF(**kw).to(dev); m.load_state_dict(P); return m
tc, tf = net(Tc), net(Tf)
pool = orc.synthetic_rays(NB * 16, seed=77).to(dev)
held = orc.synthetic_rays(2048, seed=78).to(dev)
rk = dict(N_samples=64, N_importance=128, white_bkgd=True, raw_noise_std=0.)
npa.set_precision("fp32")
with torch.no_grad():
    tgt_pool = torch.cat([npa.render_rays(pool[i:i + 4096], tc, None, network_fine=tf, perturb=0., **rk)["rgb_map"] for i in range(0, pool.shape[0], 4096)])
    tgt_held = npa.render_rays(held, tc, None, network_fine=tf, perturb=0., **rk)["rgb_map"]
def psnr(nc, nf):
    npa.set_precision("fp32")
    with torch.no_grad():
        out = npa.render_rays(held, nc, None, network_fine=nf, perturb=0., **rk)["rgb_map"]
    return -10 * math.log10(float(((out - tgt_held) ** 2).mean()))
for prec, operands in (("fp32", None), ("bf16x3", "bf16"), ("bf16x3", "fp32"), ("mixed", None)):
    if operands is not None:        # storage of the weight-gradient GEMM's operands on the split-bf16 datapath
        npa.hip_backend.WGRAD_OPERANDS = operands
    torch.manual_seed(0)
    nc, nf = net(Sc), net(Sf)
    opt = npa.FlatAdam(list(nc.parameters()) + list(nf.parameters()), lr=5e-4, betas=(0.9, 0.999))
    p0 = psnr(nc, nf)
    g = torch.Generator(device="cpu").manual_seed(1)
    curve = []
    for step in range(STEPS):
        idx = torch.randint(0, pool.shape[0], (NB,), generator=g).to(dev)
        npa.set_precision(prec)
        opt.zero_grad()
        out = npa.render_rays(pool[idx], nc, None, network_fine=nf, perturb=1.0, **rk)
        loss = npa.img2mse(out["rgb_map"], tgt_pool[idx]) + npa.img2mse(out["rgb0"], tgt_pool[idx])
        loss.backward(); opt.step()
        if (step + 1) % (STEPS // 5) == 0:
            curve.append(round(psnr(nc, nf), 3))
    print(f"{prec:7s} {'(' + operands + ' operands)' if operands else '':16s} held-out PSNR: start {p0:.3f} dB -> {curve}", flush=True)
npa.hip_backend.WGRAD_OPERANDS = "bf16"
