"""GPU parity tests (-m gpu): every HIP kernel of the hot path, called through the C ABI
(ctypes binding), against the oracle on the same seeded inputs, and the end-to-end
render_rays against golden fixtures produced by the real reference.

Tolerances (fp32 datapath; the reference's own fp32-vs-fp64 noise floor is ~2e-7 on rgb,
~1e-5 on raw):
  exact elementwise stages (z_vals)                 bit-identical
  rgb / acc / weights                               |d| <= 1e-5
  raw                                               |d| <= 2e-4 * max(1, |raw|max/10)
  gradients                                         |d| <= 1e-3 * max|grad| per tensor (fp32 sum over ~50k points)
"""
import numpy as np
import pytest
import torch

import nerf_oracle as orc

pytestmark = pytest.mark.gpu

GOLD = __import__("os").path.join(__import__("os").path.dirname(__file__), "golden")


@pytest.fixture(scope="module")
def npa():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import nerf_pytorch_amd
    return nerf_pytorch_amd


@pytest.fixture(scope="module")
def dev():
    return torch.device("cuda", 0)


@pytest.fixture(scope="module")
def nets(npa, dev):
    Pc, Pf = orc.scene_params()
    kw = dict(D=8, W=256, input_ch=63, output_ch=5, skips=[4], input_ch_views=27, use_viewdirs=True)
    nc, nf = npa.NeRF(**kw).to(dev), npa.NeRF(**kw).to(dev)
    nc.load_state_dict(Pc)
    nf.load_state_dict(Pf)
    return nc, nf, Pc, Pf


def maxdiff(a, b):
    a, b = a.detach().cpu().double(), b.detach().cpu().double()
    both_nan = torch.isnan(a) & torch.isnan(b)
    return float((a - b).abs().masked_fill(both_nan, 0.0).nan_to_num(nan=float("inf")).max()) if a.numel() else 0.0


# ---------------------------------------------------------------- stage tests
def test_embed_matches_reference_layout(npa, dev):
    x = torch.randn(1000, 3) * 3.0
    for L in (10, 4):
        got = npa.hip_backend.embed(x.to(dev), L)
        ref = orc.posenc(x, L)
        assert got.shape == ref.shape
        assert maxdiff(got, ref) <= 2e-6, maxdiff(got, ref)
    fn, ch = npa.get_embedder(10, 0)
    assert ch == 63 and maxdiff(fn(x.to(dev)), orc.posenc(x, 10)) <= 2e-6


@pytest.mark.parametrize("lindisp", [False, True])
@pytest.mark.parametrize("perturb", [False, True])
@pytest.mark.parametrize("S", [64, 50])
def test_sample_coarse_bit_exact(npa, dev, lindisp, perturb, S):
    n = 333
    rays = orc.synthetic_rays(n, seed=2)
    rays[:, 6] = torch.rand(n) + 1.5          # per-ray near/far
    rays[:, 7] = rays[:, 6] + 3.0 + torch.rand(n)
    t_rand = torch.rand(n, S) if perturb else None
    t = torch.linspace(0., 1., S)
    z = npa.hip_backend.sample_coarse(rays.to(dev), t.to(dev), lindisp, t_rand.to(dev) if perturb else None)
    near, far = rays[:, 6:7], rays[:, 7:8]
    ref = near * (1. - t) + far * t if not lindisp else 1. / (1. / near * (1. - t) + 1. / far * t)
    if perturb:
        mids = .5 * (ref[..., 1:] + ref[..., :-1])
        upper = torch.cat([mids, ref[..., -1:]], -1)
        lower = torch.cat([ref[..., :1], mids], -1)
        ref = lower + (upper - lower) * t_rand
    d = maxdiff(z, ref)
    if lindisp:
        assert d <= 1e-6, d           # division rounding may differ by an ulp between CPU and GPU
    else:
        assert torch.equal(z.cpu(), ref), d


@pytest.mark.parametrize("S,white,noise", [(64, True, False), (192, False, True), (77, True, True), (1, False, False)])
def test_raw2outputs_forward(npa, dev, S, white, noise):
    n = 257
    g = torch.Generator().manual_seed(S)
    raw = torch.randn(n, S, 4, generator=g) * 4.0
    z = torch.sort(torch.rand(n, S, generator=g) * 4.0 + 2.0, -1)[0]
    rays_d = torch.randn(n, 3, generator=g)
    nz = torch.randn(n, S, generator=g) if noise else None
    raw[:3, :, 3] = -5.0                      # empty rays: acc == 0, disp == NaN (reference quirk)
    if noise:
        nz[:3] = 0.0
    got = npa.hip_backend.raw2outputs(raw.to(dev), z.to(dev), rays_d.to(dev), 3, nz.to(dev) if noise else None,
                                      0.7 if noise else 0.0, white)
    ref64 = orc.composite(raw.double(), z.double(), rays_d.double(), nz.double() * 0.7 if noise else None, white)
    ref32 = orc.composite(raw, z, rays_d, nz * 0.7 if noise else None, white)
    names = ("rgb", "disp", "acc", "weights", "depth")
    for nm, a, b64, b32 in zip(names, got, ref64, ref32):
        if nm == "disp":
            assert torch.isnan(a[:3]).all() and torch.isnan(b32[:3]).all()
            rel = ((a.cpu().double() - b64) / b64).abs()[3:].max().item()
            assert rel <= 1e-5, (nm, rel)
        else:
            assert maxdiff(a, b64) <= 1e-5 * max(1.0, float(b64.abs().max())), (nm, maxdiff(a, b64), maxdiff(b32, b64))


@pytest.mark.parametrize("S,white,noise,use_acc_disp", [(64, True, False, False), (192, False, True, True), (33, True, True, True)])
def test_raw2outputs_backward(npa, dev, S, white, noise, use_acc_disp):
    n = 130
    g = torch.Generator().manual_seed(100 + S)
    raw = torch.randn(n, S, 4, generator=g) * 3.0
    z = torch.sort(torch.rand(n, S, generator=g) * 4.0 + 2.0, -1)[0]
    rays_d = torch.randn(n, 3, generator=g)
    nz = torch.randn(n, S, generator=g) if noise else None
    d_rgb = torch.randn(n, 3, generator=g)
    d_acc = torch.randn(n, generator=g) if use_acc_disp else None
    d_disp = torch.randn(n, generator=g) if use_acc_disp else None
    raw64 = raw.double().requires_grad_(True)
    rgb, disp, acc, _, _ = orc.composite(raw64, z.double(), rays_d.double(), nz.double() * 0.5 if noise else None, white)
    loss = (rgb * d_rgb.double()).sum()
    if use_acc_disp:
        loss = loss + (acc * d_acc.double()).sum() + (disp * d_disp.double()).sum()
    loss.backward()
    c = lambda t: t.to(dev) if t is not None else None
    got = npa.hip_backend.raw2outputs_bwd(c(raw), c(z), c(rays_d), 3, c(nz), 0.5 if noise else 0.0, white,
                                          c(d_rgb), c(d_acc), c(d_disp))
    ref = raw64.grad
    scale = float(ref.abs().max())
    assert maxdiff(got, ref) <= 2e-5 * max(1.0, scale), (maxdiff(got, ref), scale)
    # autograd wrapper (nerf_pytorch_amd.raw2outputs) agrees
    rawg = raw.to(dev).requires_grad_(True)
    if not noise:
        o = npa.raw2outputs(rawg, z.to(dev), rays_d.to(dev), 0.0, white)
        l2 = (o[0] * d_rgb.to(dev)).sum()
        if use_acc_disp:
            l2 = l2 + (o[2] * d_acc.to(dev)).sum() + (o[1] * d_disp.to(dev)).sum()
        l2.backward()
        assert maxdiff(rawg.grad, ref) <= 2e-5 * max(1.0, scale)


@pytest.mark.parametrize("det", [True, False])
@pytest.mark.parametrize("Sc,Nf", [(64, 128), (64, 64), (40, 37)])
def test_sample_fine(npa, dev, det, Sc, Nf):
    n = 300
    g = torch.Generator().manual_seed(Sc * 1000 + Nf)
    z = torch.sort(torch.rand(n, Sc, generator=g) * 4.0 + 2.0, -1)[0]
    w = torch.rand(n, Sc, generator=g) ** 8            # peaked weights
    w[:5] = 0.0                                       # all-zero weights: uniform pdf from the 1e-5 floor
    u = None if det else torch.rand(n, Nf, generator=g)
    zmid = .5 * (z[..., 1:] + z[..., :-1])
    ref_s = orc.inverse_cdf(zmid.double(), w[..., 1:-1].double(), Nf, None if det else u.double())
    ref_all = torch.sort(torch.cat([z.double(), ref_s], -1), -1)[0]
    ref_std = torch.std(ref_s, dim=-1, unbiased=False)
    c = lambda t: t.to(dev) if t is not None else None
    z_all, z_std, z_s = npa.hip_backend.sample_fine(c(z), c(w), Nf, c(u), c(torch.linspace(0., 1., Nf)), want_samples=True)
    assert (z_all[:, 1:] >= z_all[:, :-1]).all()
    # fp32 cdf rounding moves samples inside (near-)empty bins: bound by a fraction of a bin width
    assert maxdiff(z_s, ref_s) <= 2e-3, maxdiff(z_s, ref_s)
    assert np.median((z_s.cpu().double() - ref_s).abs().numpy()) <= 1e-6
    assert maxdiff(z_all, ref_all) <= 2e-3
    assert maxdiff(z_std, ref_std) <= 1e-4
    # standalone sample_pdf entry point
    s2 = npa.hip_backend.sample_pdf(c(zmid.contiguous()), c(w[..., 1:-1].contiguous()), Nf, c(u), c(torch.linspace(0., 1., Nf)))
    assert torch.equal(s2, z_s)


@pytest.mark.parametrize("n_rays,S", [(64, 64), (37, 192), (5, 3), (130, 50)])
def test_field_forward(npa, dev, nets, n_rays, S):
    nc, nf, Pc, Pf = nets
    rays = orc.synthetic_rays(n_rays, seed=S)
    z = torch.sort(torch.rand(n_rays, S) * 4.0 + 2.0, -1)[0]
    raw, act = npa.hip_backend.field_fwd(nf.packed_params(), rays.to(dev), z.to(dev), save_act=False)
    pts = rays[:, None, 0:3] + rays[:, None, 3:6] * z[..., None]
    P64 = {k: v.double() for k, v in Pf.items()}
    ref64 = orc.query_field(P64, pts.double(), rays[:, 8:11].double())
    ref32 = orc.query_field(Pf, pts, rays[:, 8:11])
    scale = max(1.0, float(ref64.abs().max()) / 10)
    assert maxdiff(raw, ref64) <= 2e-4 * scale, (maxdiff(raw, ref64), maxdiff(ref32, ref64), scale)
    # saving activations must not change the result
    raw2, act = npa.hip_backend.field_fwd(nf.packed_params(), rays.to(dev), z.to(dev), save_act=True)
    assert torch.equal(raw, raw2)


def test_field_forward_hidden_activations(npa, dev, nets):
    """What the backward reads back (saved activations, encodings, ReLU bitmasks) is what the oracle computes."""
    nc, nf, Pc, Pf = nets
    n_rays, S = 19, 64
    P = n_rays * S
    rays = orc.synthetic_rays(n_rays, seed=3)
    z = torch.sort(torch.rand(n_rays, S) * 4.0 + 2.0, -1)[0]
    raw, act = npa.hip_backend.field_fwd(nc.packed_params(), rays.to(dev), z.to(dev), save_act=True)
    act = act.cpu()
    pts = (rays[:, None, 0:3] + rays[:, None, 3:6] * z[..., None]).reshape(-1, 3)
    dirs = rays[:, None, 8:11].expand(n_rays, S, 3).reshape(-1, 3)
    feats = torch.cat([orc.posenc(pts, 10), orc.posenc(dirs, 4)], -1)
    out, hidden, feat, hv = orc.field_mlp(Pc, feats, return_hidden=True)
    off = 0
    for l in range(8):
        got = act[off:off + P * 256].view(P, 256); off += P * 256
        assert maxdiff(got, hidden[l]) <= 1e-4 * max(1.0, float(hidden[l].abs().max())), (l, maxdiff(got, hidden[l]))
    got = act[off:off + P * 256].view(P, 256); off += P * 256
    assert maxdiff(got, feat) <= 1e-4 * max(1.0, float(feat.abs().max()))
    got = act[off:off + P * 128].view(P, 128); off += P * 128
    assert maxdiff(got, hv) <= 1e-4 * max(1.0, float(hv.abs().max()))
    got = act[off:off + P * 64].view(P, 64)[:, :63]; off += P * 64
    assert maxdiff(got, feats[:, :63]) <= 5e-6
    got = act[off:off + n_rays * 32].view(n_rays, 32)[:, :27]; off += n_rays * 32
    assert maxdiff(got, orc.posenc(rays[:, 8:11], 4)) <= 5e-6


@pytest.mark.parametrize("n_rays,S", [(48, 64), (11, 192), (70, 20)])
def test_field_backward(npa, dev, nets, n_rays, S):
    nc, nf, Pc, Pf = nets
    rays = orc.synthetic_rays(n_rays, seed=S + 1)
    z = torch.sort(torch.rand(n_rays, S) * 4.0 + 2.0, -1)[0]
    d_raw = torch.randn(n_rays, S, 4)
    raw, act = npa.hip_backend.field_fwd(nf.packed_params(), rays.to(dev), z.to(dev), save_act=True)
    grad = torch.full((595844,), float("nan"), device=dev)
    npa.hip_backend.field_bwd(nf.packed_params(), act, d_raw.to(dev), grad, accumulate=False)
    P64 = {k: v.double().requires_grad_(True) for k, v in Pf.items()}
    pts = rays[:, None, 0:3] + rays[:, None, 3:6] * z[..., None]
    ref = orc.query_field(P64, pts.double(), rays[:, 8:11].double())
    (ref * d_raw.double()).sum().backward()
    grad = grad.cpu()
    assert not torch.isnan(grad).any(), "wgrad left parts of the gradient vector unwritten"
    for nm, off, shape in npa.hip_backend.param_table():
        g = grad[off:off + int(np.prod(shape))].view(shape)
        r = P64[nm].grad
        tol = 1e-3 * max(float(r.abs().max()), 1e-6)
        assert maxdiff(g, r) <= tol, (nm, maxdiff(g, r), float(r.abs().max()))
    # accumulate=True adds
    npa.hip_backend.field_bwd(nf.packed_params(), act, d_raw.to(dev), grad_dev := grad.to(dev), accumulate=True)
    assert maxdiff(grad_dev, 2 * grad) <= 1e-3 * float(grad.abs().max())


# ---------------------------------------------------------------- end to end vs golden (reference-produced)
def _check_golden(npa, dev, nets, name, kw, seed):
    nc, nf, Pc, Pf = nets
    gold = np.load(f"{GOLD}/{name}.npz")
    rays = orc.synthetic_rays(256, seed=7)
    target = torch.tensor(np.random.RandomState(99).rand(256, 3), dtype=torch.float32)
    assert abs(float(rays.double().abs().sum()) - float(gold["rays_checksum"])) < 1e-6
    n_f = kw.get("N_importance", 128)
    randoms = None
    if seed is not None:          # replay the CPU generator stream the reference consumed
        torch.manual_seed(seed)
        randoms = {}
        if kw.get("perturb", 0.) > 0:
            randoms["t_rand"] = torch.rand(256, 64)
        if kw.get("raw_noise_std", 0.) > 0:
            randoms["noise_c"] = torch.randn(256, 64)
        if n_f > 0 and kw.get("perturb", 0.) > 0:
            randoms["u"] = torch.rand(256, n_f)
        if n_f > 0 and kw.get("raw_noise_std", 0.) > 0:
            randoms["noise_f"] = torch.randn(256, 64 + n_f)
    for m in (nc, nf):
        m.zero_grad()
    args = dict(N_samples=64, retraw=True, N_importance=128, network_fine=nf, perturb=0., white_bkgd=True, raw_noise_std=0.)
    args.update(kw)
    out = npa.render_rays(rays.to(dev), nc, None, randoms=randoms, **args)
    loss = npa.img2mse(out["rgb_map"], target.to(dev))
    if "rgb0" in out:
        loss = loss + npa.img2mse(out["rgb0"], target.to(dev))
    loss.backward()
    assert abs(loss.item() - float(gold["loss"])) <= 2e-6, (loss.item(), float(gold["loss"]))
    for k in ("rgb_map", "acc_map", "rgb0", "acc0"):
        if k in gold.files:
            assert maxdiff(out[k], torch.tensor(gold[k])) <= 1e-5, (k, maxdiff(out[k], torch.tensor(gold[k])))
    for k in ("disp_map", "disp0"):
        if k in gold.files:
            a, b = out[k].cpu().double(), torch.tensor(gold[k]).double()
            ok = ~(torch.isnan(a) & torch.isnan(b))
            assert (((a - b) / b).abs()[ok] <= 2e-5).all(), k
    if "z_std" in gold.files:
        assert maxdiff(out["z_std"], torch.tensor(gold["z_std"])) <= 1e-4
    graw = torch.tensor(gold["raw"])
    assert maxdiff(out["raw"][:, ::8], graw) <= 5e-4 * max(1.0, float(graw.abs().max()) / 10), maxdiff(out["raw"][:, ::8], graw)
    mse_vs_ref = float(((out["rgb_map"].cpu() - torch.tensor(gold["rgb_map"])) ** 2).mean())
    psnr_delta_bound = 10 * np.log10(1 + mse_vs_ref / max(float(gold["loss"]), 1e-12))
    assert psnr_delta_bound < 0.01, psnr_delta_bound      # north_star: PSNR delta < 0.01 dB
    for tag, net in (("c", nc), ("f", nf)):
        for nm, p in net.named_parameters():
            key = f"{tag}/{nm}/norm"
            if key not in gold.files:
                assert p.grad is None or float(p.grad.abs().max()) == 0.0
                continue
            g = p.grad.detach().cpu().reshape(-1)
            idx, val, norm = gold[f"{tag}/{nm}/idx"], gold[f"{tag}/{nm}/val"], float(gold[key])
            assert abs(float(g.double().norm()) - norm) <= 1e-3 * max(norm, 1e-9), (tag, nm)
            assert np.abs(g[idx].numpy() - val).max() <= 1e-3 * max(np.abs(val).max(), 1e-7), (tag, nm)


def test_golden_lego_det(npa, dev, nets):
    _check_golden(npa, dev, nets, "lego_det", {}, None)


def test_golden_lego_train(npa, dev, nets):
    _check_golden(npa, dev, nets, "lego_train", dict(perturb=1.0), 123)


def test_golden_fern_train(npa, dev, nets):
    _check_golden(npa, dev, nets, "fern_train", dict(perturb=1.0, raw_noise_std=1.0, white_bkgd=False, N_importance=64, lindisp=True), 321)


def test_golden_coarse_only(npa, dev, nets):
    _check_golden(npa, dev, nets, "lego_coarse_only", dict(perturb=1.0, N_importance=0, network_fine=None), 11)


# ---------------------------------------------------------------- boundary: render() / run_network / NeRF.forward
def test_render_boundary_and_chunking(npa, dev, nets):
    nc, nf, Pc, Pf = nets
    H, W, focal = 12, 16, 20.0
    K = np.array([[focal, 0, 0.5 * W], [0, focal, 0.5 * H], [0, 0, 1]])
    c2w = torch.tensor([[1.0, 0, 0, 0.1], [0, 0.8, -0.6, 0.2], [0, 0.6, 0.8, 4.0]])
    kw = dict(network_fn=nc, network_query_fn=None, N_samples=64, N_importance=128, network_fine=nf,
              perturb=0., white_bkgd=True, raw_noise_std=0., retraw=True)
    with torch.no_grad():
        rgb, disp, acc, extras = npa.render(H, W, K, chunk=50, c2w=c2w.to(dev), ndc=False, near=2., far=6.,
                                            use_viewdirs=True, **kw)
        rgb1, _, _, _ = npa.render(H, W, K, chunk=1 << 20, c2w=c2w.to(dev), ndc=False, near=2., far=6.,
                                   use_viewdirs=True, **kw)
    assert rgb.shape == (H, W, 3) and disp.shape == (H, W) and extras["raw"].shape == (H, W, 192, 4)
    assert set(extras) == {"raw", "rgb0", "disp0", "acc0", "z_std"}
    assert torch.equal(rgb, rgb1), "chunk must not affect results (run_nerf.py:78-79)"
    ro, rd = orc.pinhole_rays(H, W, K, c2w)
    flat = orc.assemble_rays(ro.reshape(-1, 3), rd.reshape(-1, 3), 2., 6.)
    ref = orc.trace_rays(flat, Pc, Pf, 64, 128, perturb=0., white_bkgd=True)
    assert maxdiff(rgb.reshape(-1, 3), ref["rgb_map"]) <= 1e-5
    # run_network / NeRF.forward on explicit points
    pts = torch.randn(7, 5, 3)
    vd = torch.nn.functional.normalize(torch.randn(7, 3), dim=-1)
    with torch.no_grad():
        got = npa.run_network(pts.to(dev), vd.to(dev), nf, None, None)
        emb = torch.cat([orc.posenc(pts.reshape(-1, 3), 10), orc.posenc(vd[:, None].expand(7, 5, 3).reshape(-1, 3), 4)], -1)
        got2 = nf(emb.to(dev))
    ref = orc.query_field(Pf, pts, vd)
    assert maxdiff(got, ref) <= 2e-4 * max(1.0, float(ref.abs().max()) / 10)
    assert maxdiff(got2.reshape(7, 5, 4), ref) <= 2e-4 * max(1.0, float(ref.abs().max()) / 10)


def test_training_step_moves_parameters_like_the_oracle(npa, dev):
    """Two Adam steps through the drop-in surface == two Adam steps of the oracle."""
    Pc, Pf = orc.scene_params(seed=1)
    kw = dict(D=8, W=256, input_ch=63, output_ch=5, skips=[4], input_ch_views=27, use_viewdirs=True)
    nc, nf = npa.NeRF(**kw).to(dev), npa.NeRF(**kw).to(dev)
    nc.load_state_dict(Pc); nf.load_state_dict(Pf)
    opt = torch.optim.Adam(list(nc.parameters()) + list(nf.parameters()), lr=5e-4, betas=(0.9, 0.999))
    Pc_o = {k: v.clone().requires_grad_(True) for k, v in Pc.items()}
    Pf_o = {k: v.clone().requires_grad_(True) for k, v in Pf.items()}
    opt_o = torch.optim.Adam(list(Pc_o.values()) + list(Pf_o.values()), lr=5e-4, betas=(0.9, 0.999))
    rays = orc.synthetic_rays(96, seed=21)
    target = torch.rand(96, 3)
    for step in range(2):
        opt.zero_grad()
        out = npa.render_rays(rays.to(dev), nc, None, 64, N_importance=128, network_fine=nf, white_bkgd=True)
        loss = npa.img2mse(out["rgb_map"], target.to(dev)) + npa.img2mse(out["rgb0"], target.to(dev))
        loss.backward()
        opt.step()
        opt_o.zero_grad()
        ref = orc.trace_rays(rays, Pc_o, Pf_o, 64, 128, white_bkgd=True)
        lo = orc.mse(ref["rgb_map"], target) + orc.mse(ref["rgb0"], target)
        lo.backward()
        opt_o.step()
        assert abs(loss.item() - lo.item()) <= 5e-6, (step, loss.item(), lo.item())
    assert nc._is_bound() and nf._is_bound()
