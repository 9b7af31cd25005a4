"""Pin oracle/nerf_oracle.py against the real reference (build container only).

Imports /root/reference/run_nerf.py with ``imageio`` and ``cv2`` stubbed (they
are used only by loaders / image writers), runs the reference functions and the
oracle restatement on identical inputs on CPU, and asserts bit-identical fp32
results function by function.  /root/reference does not exist on the GPU box,
so this script is run here (``python oracle/pin_against_reference.py``) and by
``tests/test_oracle_golden.py`` when the reference is present.
"""
import os
import sys
import types

import numpy as np
import torch

REF = os.environ.get("NERF_REFERENCE", "/root/reference")
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import nerf_oracle as orc  # noqa: E402


def load_reference():
    for name in ("imageio", "cv2"):
        if name not in sys.modules:
            sys.modules[name] = types.ModuleType(name)
    if REF not in sys.path:
        sys.path.insert(0, REF)
    import run_nerf  # noqa
    import run_nerf_helpers  # noqa
    return run_nerf, run_nerf_helpers


def reference_networks(helpers, P):
    net = helpers.NeRF(D=8, W=256, input_ch=63, output_ch=5, skips=[4], input_ch_views=27, use_viewdirs=True)
    net.load_state_dict({k: v.clone() for k, v in P.items()})
    return net


def same(a, b, what):
    a = a.detach()
    b = b.detach()
    eq = torch.equal(a, b) or bool(((a == b) | (torch.isnan(a) & torch.isnan(b))).all())
    md = float((a - b).abs().nan_to_num(0).max()) if a.numel() else 0.0
    print(f"  {what:28s} bit-identical={eq}  max|d|={md:.3e}")
    assert eq, f"{what}: oracle differs from reference (max|d|={md})"


def main(n_rays=96):
    torch.manual_seed(0)
    run_nerf, helpers = load_reference()
    Pc = orc.make_params(11, gain=1.0, sigma_bias=0.5)
    Pf = orc.make_params(12, gain=1.0, sigma_bias=0.5)
    net_c, net_f = reference_networks(helpers, Pc), reference_networks(helpers, Pf)
    embed_fn, ch = helpers.get_embedder(10, 0)
    embeddirs_fn, chv = helpers.get_embedder(4, 0)
    assert (ch, chv) == (63, 27)
    qfn = lambda inputs, viewdirs, network_fn: run_nerf.run_network(
        inputs, viewdirs, network_fn, embed_fn=embed_fn, embeddirs_fn=embeddirs_fn, netchunk=1024 * 64)

    rays = orc.synthetic_rays(n_rays, seed=3)
    print("posenc / field_mlp")
    x = torch.randn(257, 3) * 3.0
    same(orc.posenc(x, 10), embed_fn(x), "posenc L=10")
    same(orc.posenc(x, 4), embeddirs_fn(x), "posenc L=4")
    feats = torch.cat([embed_fn(x), embeddirs_fn(x / x.norm(dim=-1, keepdim=True))], -1)
    same(orc.field_mlp(Pc, feats), net_c(feats), "field_mlp")

    print("composite / inverse_cdf")
    raw = torch.randn(n_rays, 64, 4) * 3.0
    z = torch.sort(torch.rand(n_rays, 64) * 4.0 + 2.0, -1)[0]
    for wb in (False, True):
        got = orc.composite(raw, z, rays[:, 3:6], None, wb)
        ref = run_nerf.raw2outputs(raw, z, rays[:, 3:6], 0.0, wb)
        for g, r, nm in zip(got, ref, ("rgb", "disp", "acc", "weights", "depth")):
            same(g, r, f"composite wb={wb} {nm}")
    w = got[3]
    zmid = 0.5 * (z[..., 1:] + z[..., :-1])
    same(orc.inverse_cdf(zmid, w[..., 1:-1], 128, None), helpers.sample_pdf(zmid, w[..., 1:-1], 128, det=True),
         "inverse_cdf det")
    # random u: reference draws torch.rand inside; replay the same stream
    torch.manual_seed(5)
    ref_s = helpers.sample_pdf(zmid, w[..., 1:-1], 128, det=False)
    torch.manual_seed(5)
    u = torch.rand(n_rays, 128)
    same(orc.inverse_cdf(zmid, w[..., 1:-1], 128, u), ref_s, "inverse_cdf random u")

    print("trace_rays (render_rays), deterministic test-time configuration")
    kw = dict(network_fn=net_c, network_query_fn=qfn, N_samples=64, retraw=True, lindisp=False, perturb=0.0,
              N_importance=128, network_fine=net_f, white_bkgd=True, raw_noise_std=0.0)
    ref = run_nerf.render_rays(rays, **kw)
    got = orc.trace_rays(rays, Pc, Pf, 64, 128, perturb=0.0, white_bkgd=True, retraw=True)
    for k in ref:
        same(got[k], ref[k], f"trace det {k}")

    print("trace_rays, training configuration (perturb=1, raw_noise_std=1, lindisp) with replayed RNG")
    kw.update(perturb=1.0, raw_noise_std=1.0, lindisp=True, white_bkgd=False)
    torch.manual_seed(9)
    ref = run_nerf.render_rays(rays, **kw)
    torch.manual_seed(9)      # replay the order of draws of run_nerf.py:371,:285, helpers:208, :285
    t_rand = torch.rand(n_rays, 64)
    noise_c = torch.randn(n_rays, 64)
    u = torch.rand(n_rays, 128)
    noise_f = torch.randn(n_rays, 192)
    got = orc.trace_rays(rays, Pc, Pf, 64, 128, perturb=1.0, lindisp=True, white_bkgd=False, raw_noise_std=1.0,
                         retraw=True, t_rand=t_rand, u=u, noise_c=noise_c, noise_f=noise_f)
    for k in ref:
        same(got[k], ref[k], f"trace train {k}")

    print("gradients (loss of run_nerf.py:765-772)")
    target = torch.rand(n_rays, 3)
    kw.update(perturb=0.0, raw_noise_std=0.0, lindisp=False, white_bkgd=True)
    ref = run_nerf.render_rays(rays, **kw)
    loss = helpers.img2mse(ref["rgb_map"], target) + helpers.img2mse(ref["rgb0"], target)
    loss.backward()
    Pc_g = {k: v.clone().requires_grad_(True) for k, v in Pc.items()}
    Pf_g = {k: v.clone().requires_grad_(True) for k, v in Pf.items()}
    got = orc.trace_rays(rays, Pc_g, Pf_g, 64, 128, perturb=0.0, white_bkgd=True)
    l2 = orc.mse(got["rgb_map"], target) + orc.mse(got["rgb0"], target)
    l2.backward()
    same(l2, loss, "loss")
    for (k, p) in net_c.state_dict(keep_vars=True).items():
        same(Pc_g[k].grad, p.grad, f"grad coarse {k}")
    for (k, p) in net_f.state_dict(keep_vars=True).items():
        same(Pf_g[k].grad, p.grad, f"grad fine {k}")

    print("assemble_rays / pinhole_rays / ndc_warp via reference render() boundary")
    H, W, focal = 20, 24, 30.0
    K = np.array([[focal, 0, 0.5 * W], [0, focal, 0.5 * H], [0, 0, 1]])
    c2w = torch.tensor([[1.0, 0, 0, 0.1], [0, 0.8, -0.6, 0.2], [0, 0.6, 0.8, 4.0]])
    ro, rd = helpers.get_rays(H, W, K, c2w)
    go, gd = orc.pinhole_rays(H, W, K, c2w)
    same(go, ro, "pinhole rays_o")
    same(gd, rd, "pinhole rays_d")
    c2w_ff = torch.tensor([[1.0, 0, 0, 0.1], [0, 1.0, 0, 0.2], [0, 0, 1.0, 0.3]])
    ro2, rd2 = helpers.get_rays(H, W, K, c2w_ff)
    a, b = helpers.ndc_rays(H, W, focal, 1.0, ro2, rd2)
    c, d = orc.ndc_warp(H, W, focal, 1.0, ro2, rd2)
    same(c, a, "ndc rays_o")
    same(d, b, "ndc rays_d")
    kw_r = dict(kw)
    ref_list = run_nerf.render(H, W, K, chunk=128, rays=torch.stack([ro.reshape(-1, 3)[:64], rd.reshape(-1, 3)[:64]], 0),
                               ndc=False, near=2.0, far=6.0, use_viewdirs=True, **kw_r)
    flat = orc.assemble_rays(ro.reshape(-1, 3)[:64], rd.reshape(-1, 3)[:64], 2.0, 6.0)
    got = orc.trace_in_chunks(flat, 128, P_coarse=Pc, P_fine=Pf, n_coarse=64, n_fine=128, perturb=0.0,
                              white_bkgd=True, retraw=True)
    same(got["rgb_map"], ref_list[0], "render() rgb_map")
    same(got["disp_map"], ref_list[1], "render() disp_map")
    same(got["acc_map"], ref_list[2], "render() acc_map")
    same(got["raw"], ref_list[3]["raw"], "render() raw")
    print("render(ndc=True) on forward-facing rays (configs/fern.txt: near=0, far=1, raw_noise_std=1, no white_bkgd)")
    cfg = orc.FERN
    Kf = orc.intrinsics(cfg)
    batch = orc.fern_batch(48, seed=2)
    kw_f = dict(kw)
    kw_f.update(perturb=1.0, raw_noise_std=1.0, white_bkgd=False)
    torch.manual_seed(21)
    ref_list = run_nerf.render(cfg["H"], cfg["W"], Kf, chunk=1024, rays=batch, ndc=True, near=0.0, far=1.0,
                               use_viewdirs=True, **kw_f)
    torch.manual_seed(21)
    rnd = dict(t_rand=torch.rand(48, 64), noise_c=torch.randn(48, 64), u=torch.rand(48, 128), noise_f=torch.randn(48, 192))
    flat = orc.assemble_render_rays(cfg["H"], cfg["W"], Kf, batch[0], batch[1], True, 0.0, 1.0)
    got = orc.trace_rays(flat, Pc, Pf, 64, 128, perturb=1.0, white_bkgd=False, raw_noise_std=1.0, retraw=True, **rnd)
    same(got["rgb_map"], ref_list[0], "render(ndc) rgb_map")
    same(got["disp_map"], ref_list[1], "render(ndc) disp_map")
    same(got["acc_map"], ref_list[2], "render(ndc) acc_map")
    same(got["raw"], ref_list[3]["raw"], "render(ndc) raw")
    same(got["z_std"], ref_list[3]["z_std"], "render(ndc) z_std")
    print("other architectures (netdepth / netwidth / multires / i_embed = -1 / use_viewdirs = False -> output_linear)")
    ARCHS = {"no_viewdirs": orc.arch_of(use_viewdirs=False), "narrow_shallow": orc.arch_of(D=6, W=128, multires=6, multires_views=2),
             "identity_embedding": orc.arch_of(D=4, W=64, multires=-1, multires_views=-1, output_ch=4),
             "no_viewdirs_small": orc.arch_of(D=7, W=96, multires=3, use_viewdirs=False, output_ch=4)}
    for name, arch in ARCHS.items():
        A = {k: arch[k] for k in ("D", "W", "input_ch", "input_ch_views", "output_ch", "skips", "use_viewdirs")}
        nets, Ps = [], []
        for seed in (31, 32):
            P = orc.make_arch_params(arch, seed)
            net = helpers.NeRF(**A)
            assert [(k, tuple(v.shape)) for k, v in net.state_dict().items()] == orc.arch_param_shapes(arch), name
            net.load_state_dict({k: v.clone() for k, v in P.items()})
            nets.append(net)
            Ps.append(P)
        i_embed = -1 if arch["multires"] < 0 else 0
        e_fn, ch = helpers.get_embedder(arch["multires"], i_embed)
        ed_fn, chv = helpers.get_embedder(arch["multires_views"], i_embed) if arch["use_viewdirs"] else (None, 0)
        assert (ch, chv) == (arch["input_ch"], arch["input_ch_views"]), name
        qfn = lambda inputs, viewdirs, network_fn, e_fn=e_fn, ed_fn=ed_fn: run_nerf.run_network(
            inputs, viewdirs, network_fn, embed_fn=e_fn, embeddirs_fn=ed_fn, netchunk=1024 * 64)
        rr = rays if arch["use_viewdirs"] else rays[:, :8]          # render() appends view directions only with use_viewdirs
        torch.manual_seed(13)
        ref = run_nerf.render_rays(rr, network_fn=nets[0], network_query_fn=qfn, N_samples=24, retraw=True, lindisp=False, perturb=1.0,
                                   N_importance=40, network_fine=nets[1], white_bkgd=True, raw_noise_std=0.5)
        torch.manual_seed(13)
        rnd = dict(t_rand=torch.rand(n_rays, 24), noise_c=torch.randn(n_rays, 24), u=torch.rand(n_rays, 40), noise_f=torch.randn(n_rays, 64))
        Pg = [{k: v.clone().requires_grad_(True) for k, v in P.items()} for P in Ps]
        got = orc.trace_rays(rr, Pg[0], Pg[1], 24, 40, perturb=1.0, white_bkgd=True, raw_noise_std=0.5, retraw=True, arch=arch, **rnd)
        for k in ref:
            same(got[k], ref[k], f"{name} {k}")
        loss = helpers.img2mse(ref["rgb_map"], target) + helpers.img2mse(ref["rgb0"], target)
        loss.backward()
        (orc.mse(got["rgb_map"], target) + orc.mse(got["rgb0"], target)).backward()
        for net, P in zip(nets, Pg):
            for k, p in net.state_dict(keep_vars=True).items():
                if p.grad is not None or P[k].grad is not None:         # views_linears is unused without view directions
                    same(P[k].grad, p.grad, f"{name} grad {k}")
    print("ORACLE PINNED: every function bit-identical to the reference on CPU")


if __name__ == "__main__":
    main()
