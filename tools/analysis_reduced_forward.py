"""CPU, oracle: the north-star gate (PSNR delta vs the reference's image at a teacher target) if every linear layer of the forward
saw its input rounded to fp16 / tf32 / bf16 (weights exact) -- the accuracy class a 2-MFMA-per-product forward would have.
Result (round 3): lego-like 1.7e-4 dB / 75 dB vs the reference's image (passes the 0.01 dB bar), fern-like NDC 1.24 dB / 42 dB
(fails it by two orders of magnitude: hierarchical sampling amplifies the coarse pass's rounding there).  Usage: python tools/analysis_reduced_forward.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'oracle'))
import numpy as np, torch
import nerf_oracle as orc, workloads as wl
torch.set_num_threads(32)
def rnd(x, mode):
    if mode == "fp16": return x.half().float()
    if mode == "bf16": return x.bfloat16().float()
    if mode == "tf32": # 10-bit mantissa RNE
        xi = x.view(torch.int32); r = ((xi + 0x0FFF + ((xi >> 13) & 1)) & ~0x1FFF); return r.view(torch.float32)
    return x
def lin(x, W, b, mode):
    return torch.nn.functional.linear(rnd(x, mode), W, b)
def mlp(P, feats, mode):
    xyz, dirs = feats[:, :63], feats[:, 63:]
    h = xyz
    for i in range(8):
        h = torch.relu(lin(h, P[f"pts_linears.{i}.weight"], P[f"pts_linears.{i}.bias"], mode))
        if i == 4: h = torch.cat([xyz, h], -1)
    sigma = torch.nn.functional.linear(h, P["alpha_linear.weight"], P["alpha_linear.bias"])
    feat = lin(h, P["feature_linear.weight"], P["feature_linear.bias"], mode)
    hv = torch.relu(lin(torch.cat([feat, dirs], -1), P["views_linears.0.weight"], P["views_linears.0.bias"], mode))
    rgb = torch.nn.functional.linear(hv, P["rgb_linear.weight"], P["rgb_linear.bias"])
    return torch.cat([rgb, sigma], -1)
import types
for cfgname in ("lego", "fern"):
    gold = np.load(os.path.join(ROOT, "tests", "golden", f"gate_{cfgname}.npz"))
    cfg = wl.LEGO if cfgname == "lego" else wl.FERN
    batch = wl.lego_batch(1024, seed=31) if cfgname == "lego" else wl.fern_batch(1024, seed=32)
    Pc, Pf = wl.scene_params()
    flat = orc.assemble_render_rays(cfg["H"], cfg["W"], wl.intrinsics(cfg), batch[0], batch[1], cfg["ndc"], cfg["near"], cfg["far"])
    for mode in ("fp32", "fp16", "tf32", "bf16"):
        orig = orc.field_mlp
        orc.field_mlp = lambda P, feats, mode=mode, **kw: mlp(P, feats, mode)
        try:
            with torch.no_grad():
                out = orc.trace_rays(flat, Pc, Pf, 64, 128, perturb=0., white_bkgd=cfg["white_bkgd"], raw_noise_std=0.)
        finally:
            orc.field_mlp = orig
        g = wl.precision_gate(out["rgb_map"], torch.tensor(gold["rgb_ref"]), torch.tensor(gold["target"]))
        print(cfgname, mode, {k: (round(v, 6) if isinstance(v, float) else v) for k, v in g.items()}, flush=True)
