"""CPU analysis (fp64): how far is the weight gradient dW = delta^T x when ONLY the stored GEMM operands are rounded to
bf16 (the dgrad chain itself exact)?  Compared with rounding nothing (reference) on a realistic loss (teacher-scene
target).  Not product code."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np, torch
import nerf_oracle as orc
import workloads as wl
torch.set_num_threads(8)

def bf16(t):
    return t.float().to(torch.bfloat16).double()

def run(n_rays):
    cfg = wl.LEGO
    batch = wl.lego_batch(n_rays, seed=5)
    rays = orc.assemble_render_rays(cfg["H"], cfg["W"], wl.intrinsics(cfg), batch[0], batch[1], cfg["ndc"], cfg["near"], cfg["far"])
    Pc, Pf = wl.scene_params()
    Tc, Tf = wl.teacher_params()
    with torch.no_grad():
        target = orc.trace_rays(rays, Tc, Tf, 64, 128, perturb=0.0, white_bkgd=True)["rgb_map"].double()
    P = {k: v.double().requires_grad_(True) for k, v in Pc.items()}
    r = rays.double()
    o, d, near, far, vd = r[:, 0:3], r[:, 3:6], r[:, 6:7], r[:, 7:8], r[:, 8:11]
    t = torch.linspace(0, 1, 64, dtype=torch.float64)
    z = near * (1 - t) + far * t
    pts = o[:, None] + d[:, None] * z[..., None]
    feats = torch.cat([orc.posenc(pts.reshape(-1, 3), 10), orc.posenc(vd[:, None].expand(pts.shape).reshape(-1, 3), 4)], -1)
    raw, hidden, feat, hv = orc.field_mlp(P, feats, return_hidden=True)
    for h in hidden + [feat, hv]:
        h.retain_grad()
    rgb = orc.composite(raw.reshape(n_rays, 64, 4), z, d, None, True)[0]
    loss = ((rgb - target) ** 2).mean()
    loss.backward()
    xyz = feats[:, :63]
    # (delta_l, input_l) pairs of the 256-wide trunk layers 1..7 (pre-activation delta = grad(h_l) * [h_l > 0])
    tot_e = tot_r = dot = 0.0
    out = []
    for l in range(1, 8):
        delta = hidden[l].grad * (hidden[l] > 0)
        x = hidden[l - 1].detach()
        if l == 5:
            x = torch.cat([xyz, x], -1)
        exact = delta.T @ x
        assert float((exact - P[f"pts_linears.{l}.weight"].grad).abs().max()) < 1e-12
        approx = bf16(delta).T @ bf16(x)
        e = float((approx - exact).norm()); n = float(exact.norm())
        out.append(e / n)
        tot_e += e * e; tot_r += n * n
        dot += float((approx * exact).sum()); 
    return out, (tot_e / tot_r) ** 0.5

for n in (128, 512, 2048):
    per, total = run(n)
    print(n * 64, "points: rel L2 error per layer", " ".join("%.2e" % v for v in per), "| trunk total %.2e" % total, flush=True)
