"""Where does the bf16x3 datapath go wrong in tests/test_gpu_golden_cfg.py::test_golden_cfg4 (rgb0 off by 0.42, bf16x3 only)?"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np, torch
import nerf_oracle as orc
import nerf_pytorch_amd as npa
hb = npa.hip_backend
dev = torch.device("cuda", 0)
Pc, Pf = orc.scene_params()
kw = dict(D=8, W=256, input_ch=63, output_ch=5, skips=[4], input_ch_views=27, use_viewdirs=True)
nc, nf = npa.NeRF(**kw).to(dev), npa.NeRF(**kw).to(dev)
nc.load_state_dict(Pc); nf.load_state_dict(Pf)
gold = np.load(os.path.join(ROOT, "tests", "golden", "lego_cfg4_forward.npz"))
n = 32768
batch = orc.lego_batch(n, seed=19)
torch.manual_seed(2024)
rnd = {"t_rand": torch.rand(n, 64), "u": torch.rand(n, 128)}
K = orc.intrinsics(dict(orc.LEGO, H=800, W=800, focal=1111.0))
args = dict(ndc=False, near=2.0, far=6.0, use_viewdirs=True, network_fn=nc, network_query_fn=None, N_samples=64, N_importance=128,
            network_fine=nf, perturb=1.0, white_bkgd=True, raw_noise_std=0.0)
for prec in ("bf16x3", "fp16x3", "bf16x3"):
    for variant in ("render cpu-rnd", "render dev-rnd", "no_grad"):
        npa.set_precision(prec)
        r = rnd if variant == "render cpu-rnd" else {k: v.to(dev) for k, v in rnd.items()}
        with torch.set_grad_enabled(variant != "no_grad"):
            rgb, disp, acc, ex = npa.render(800, 800, K, chunk=1024 * 32, rays=batch.to(dev), randoms=r, **args)
        e0 = (ex["rgb0"].detach().cpu() - torch.tensor(gold["rgb0"])).abs().amax(-1)
        e1 = (rgb.detach().cpu() - torch.tensor(gold["rgb_map"])).abs().amax(-1)
        bad = (e0 > 1e-3).nonzero().flatten()
        print(prec, variant, "rgb0 max", float(e0.max()), "bad", bad.numel(), bad[:8].tolist(), bad[-4:].tolist() if bad.numel() else "", "rgb p95", float(torch.quantile(e1, 0.95)), flush=True)
        del rgb, disp, acc, ex
npa.set_precision("fp32")
