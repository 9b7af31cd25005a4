import os, sys, ctypes
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import torch, numpy as np
import nerf_oracle as orc
import nerf_pytorch_amd as npa
hb = npa.hip_backend
if len(sys.argv) > 1:
    npa.build.LIB_PATH = os.path.join(ROOT, "nerf-pytorch_amd", sys.argv[1]); hb._LIB = None
dev = torch.device("cuda", 0)
Pc, Pf = orc.scene_params()
kw = dict(D=8, W=256, input_ch=63, output_ch=5, skips=[4], input_ch_views=27, use_viewdirs=True)
nf = npa.NeRF(**kw).to(dev); nf.load_state_dict(Pf)
for (n_rays, S) in [(48, 64), (11, 192), (70, 20)]:
    torch.manual_seed(0)
    rays = orc.synthetic_rays(n_rays, seed=S + 1)
    z = torch.sort(torch.rand(n_rays, S) * 4.0 + 2.0, -1)[0]
    d_raw = torch.randn(n_rays, S, 4)
    P = n_rays * S
    packed = nf.packed_params()
    # poison allocations so unwritten rows show up
    junk = torch.full((hb.lib().nerf_delta_floats(n_rays, S) + 64,), 1e6, device=dev); del junk
    raw, act = hb.field_fwd(packed, rays.to(dev), z.to(dev), save_act=True)
    L = hb.lib()
    delta = torch.full((L.nerf_delta_floats(n_rays, S),), float("nan"), device=dev)
    hb._check(L.nerf_field_dgrad(packed.data_ptr(), act.data_ptr(), d_raw.to(dev).data_ptr(), n_rays, S, delta.data_ptr(), torch.cuda.current_stream().cuda_stream), "dgrad")
    torch.cuda.synchronize()
    # reference deltas via autograd hooks (fp64)
    P64 = {k: v.double() for k, v in Pf.items()}
    pts = (rays[:, None, 0:3] + rays[:, None, 3:6] * z[..., None]).reshape(-1, 3).double()
    dirs = rays[:, None, 8:11].expand(n_rays, S, 3).reshape(-1, 3).double()
    feats = torch.cat([orc.posenc(pts, 10), orc.posenc(dirs, 4)], -1)
    lin = torch.nn.functional.linear
    xyz, dd = feats[:, :63], feats[:, 63:]
    leaves = {k: v.clone().requires_grad_(True) for k, v in P64.items()}
    hh = xyz; pres = []
    for i in range(8):
        a = lin(hh, leaves[f"pts_linears.{i}.weight"], leaves[f"pts_linears.{i}.bias"]); a.retain_grad(); pres.append(a)
        hh = torch.relu(a)
        if i == 4: hh = torch.cat([xyz, hh], -1)
    sigma = lin(hh, leaves["alpha_linear.weight"], leaves["alpha_linear.bias"])
    ft = lin(hh, leaves["feature_linear.weight"], leaves["feature_linear.bias"]); ft.retain_grad()
    av = lin(torch.cat([ft, dd], -1), leaves["views_linears.0.weight"], leaves["views_linears.0.bias"]); av.retain_grad()
    rgb = lin(torch.relu(av), leaves["rgb_linear.weight"], leaves["rgb_linear.bias"])
    out = torch.cat([rgb, sigma], -1)
    (out * d_raw.reshape(-1, 4).double()).sum().backward()
    dl = delta.cpu()
    res = {}
    for i in range(8):
        got = dl[i * P * 256:(i + 1) * P * 256].view(P, 256).double()
        ref = pres[i].grad
        bad_rows = ((got - ref).abs().max(-1)[0] > 1e-4 * ref.abs().max()).nonzero().flatten()
        res[f"h{i}"] = (float((got - ref).abs().max() / ref.abs().max()), int(torch.isnan(got).sum()), bad_rows[:8].tolist(), len(bad_rows))
    got = dl[8 * P * 256:9 * P * 256].view(P, 256).double(); res["feat"] = float((got - ft.grad).abs().max() / ft.grad.abs().max())
    got = dl[9 * P * 256:9 * P * 256 + P * 128].view(P, 128).double(); res["hv"] = float((got - av.grad).abs().max() / av.grad.abs().max())
    # masks
    al_mask_off = 8*P*256 + P*256 + P*128 + P*64 + n_rays*32 + P*32; al_mask_off = (al_mask_off + 3) & ~3
    m = act.cpu()[al_mask_off:al_mask_off + 9*P*8].view(torch.int32).view(9, P, 4, 2)
    mres = {}
    for i in range(8):
        refm = (pres[i].detach() > 0)
        bits = torch.zeros(P, 256, dtype=torch.bool)
        for q in range(4):
            for reg in range(64):
                nb, r = reg // 4, reg % 4
                word = m[i, :, q, 0 if reg < 32 else 1]
                bits[:, 16*nb + 4*q + r] = ((word >> (reg % 32)) & 1).bool()
        mres[f"m{i}"] = int((bits != refm).sum())
    print(sys.argv[1:] , (n_rays, S), "delta err:", {k: (f"{v[0]:.1e}", v[1], v[3], v[2]) if isinstance(v, tuple) else f"{v:.1e}" for k, v in res.items()}, "mask mismatches:", mres, flush=True)
