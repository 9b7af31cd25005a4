"""cfg5-like inference: render_path over 800x800 spiral poses (chunk 32768), frames/s with overlapped output."""
import os, sys, time, math, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np, torch
import nerf_oracle as orc
import nerf_pytorch_amd as npa
dev = torch.device("cuda", 0)
kw = dict(D=8, W=256, input_ch=63, output_ch=5, skips=[4], input_ch_views=27, use_viewdirs=True)
Pc, Pf = orc.scene_params()
nc, nf = npa.NeRF(**kw).to(dev), npa.NeRF(**kw).to(dev)
nc.load_state_dict(Pc); nf.load_state_dict(Pf)
H = W = 800; focal = 1111.0
K = np.array([[focal, 0, 0.5 * W], [0, focal, 0.5 * H], [0, 0, 1]])
def pose(theta):        # camera on a circle of radius 4 looking at the origin
    c, s = math.cos(theta), math.sin(theta)
    return torch.tensor([[c, 0, s, 4 * s], [0, 1, 0, 0], [-s, 0, c, 4 * c], [0, 0, 0, 1.0]])
poses = torch.stack([pose(0.15 * i) for i in range(int(os.environ.get("FRAMES", 6)))]).to(dev)
rk = dict(network_fn=nc, network_query_fn=None, N_samples=64, N_importance=128, network_fine=nf, perturb=0., white_bkgd=True,
          raw_noise_std=0., ndc=False, near=2., far=6., use_viewdirs=True)
npa.set_precision(os.environ.get("PREC", "fp16x3"))
with torch.no_grad(), tempfile.TemporaryDirectory() as d:
    npa.render_path(poses[:1], (H, W, focal), K, 32768, rk)          # warm-up
    for savedir in (None, d):
        torch.cuda.synchronize(); t = time.perf_counter()
        rgbs, disps = npa.render_path(poses, (H, W, focal), K, 32768, rk, savedir=savedir)
        torch.cuda.synchronize(); el = time.perf_counter() - t
        n = len(poses)
        print(f"savedir={'yes' if savedir else 'no '}: {n} frames of {H}x{W} in {el:.2f} s = {el / n:.3f} s/frame = {n * H * W / el / 1e6:.3f} M rays/s; rgb range [{rgbs.min():.3f}, {rgbs.max():.3f}]", flush=True)
