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


# The datapath of a boundary test: "fp32" = the exact anchor (the suite's default, tests/conftest.py), "fp16x3" = what a user gets
# without asking (render.DEFAULT_PRECISION).  BOUNDARY_TOL: per-ray / per-entry bounds of those tests on each (GOLD_TOL's classes).
BOUNDARY_DATAPATHS = ["fp32", "fp16x3"]
BOUNDARY_TOL = {"fp32": dict(coarse=1e-5, fine=1e-5, field=2e-4, grad=2e-3, loss=2e-4),
                "fp16x3": dict(coarse=3e-5, fine=3e-5, field=2e-4, grad=4e-3, loss=2e-4)}


@pytest.fixture
def datapath(npa, request):
    prev = npa.get_precision()
    npa.set_precision(request.param)
    yield request.param
    npa.set_precision(prev)


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
    assert torch.equal(z.cpu(), ref), (maxdiff(z, ref), float((z.cpu() != ref).float().mean()))


@pytest.mark.parametrize("S,white,noise", [(64, True, False), (192, False, True), (77, True, True), (2, False, False)])
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
            both_nan = torch.isnan(a.cpu()) & torch.isnan(b64)
            rel = ((a.cpu().double() - b64) / b64).abs().masked_fill(both_nan, 0.0).max().item()
            assert rel <= 1e-5, (nm, rel)
        else:
            assert maxdiff(a, b64) <= 1e-5 * max(1.0, float(b64.abs().max())), (nm, S, maxdiff(a, b64), maxdiff(b32, b64), float(b64.abs().max()))


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
    w[5:10] *= 0.1                                    # sum < 1: no entry below the denom guard
    u = None if det else torch.rand(n, Nf, generator=g)
    zmid = .5 * (z[..., 1:] + z[..., :-1])
    ref_s = orc.inverse_cdf(zmid.double(), w[..., 1:-1].double(), Nf, None if det else u.double())
    c = lambda t: t.to(dev) if t is not None else None
    z_all, z_std, z_s = npa.hip_backend.sample_fine(c(z), c(w), Nf, c(u), c(torch.linspace(0., 1., Nf)), want_samples=True)
    z_s64 = z_s.cpu().double()
    diff = (z_s64 - ref_s).abs()
    if det:
        # the u == 1.0 endpoint is rounding-dependent in the reference when the last bin is (near) empty
        unstable = orc.endpoint_unstable(w)
        last = diff[:, -1]
        # the two legal outcomes: right edge of the last bin (cdf[-1] <= 1) or its left edge (cdf[-1] > 1)
        right = (z_s64[:, -1] - zmid[:, -1].double()).abs()
        left = (z_s64[:, -1] - zmid[:, -2].double()).abs()
        assert ((last <= 2e-3) | (unstable & ((left <= 2e-3) | (right <= 2e-3)))).all(), (last.max(), unstable.sum())
        assert (last[~unstable] <= 2e-3).all()
        diff = diff[:, :-1]
    # fp32 cdf rounding moves samples inside (near-)empty bins: bound by a fraction of a bin width
    assert diff.max() <= 2e-3, diff.max()
    assert np.median(diff.numpy()) <= 1e-6
    # sort and std are checked against this call's own samples (independent of the endpoint flip)
    assert (z_all[:, 1:] >= z_all[:, :-1]).all()
    assert torch.equal(z_all.cpu(), torch.sort(torch.cat([z, z_s.cpu()], -1), -1)[0])
    assert maxdiff(z_std, torch.std(z_s64, dim=-1, unbiased=False)) <= 1e-5
    # standalone sample_pdf entry point
    s2 = npa.hip_backend.sample_pdf(c(zmid.contiguous()), c(w[..., 1:-1].contiguous()), Nf, c(u), c(torch.linspace(0., 1., Nf)))
    assert torch.equal(s2, z_s)
    if det:
        s3 = npa.sample_pdf(c(zmid), c(w[..., 1:-1]), Nf, det=True)
        assert torch.equal(s3, z_s)


@pytest.mark.parametrize("n_rays,S", [(64, 64), (37, 192), (5, 3), (130, 50)])
def test_field_forward(npa, dev, nets, n_rays, S):
    nc, nf, Pc, Pf = nets
    rays = orc.synthetic_rays(n_rays, seed=S)
    z = torch.sort(torch.rand(n_rays, S, generator=torch.Generator().manual_seed(S)) * 4.0 + 2.0, -1)[0]
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
    z = torch.sort(torch.rand(n_rays, S, generator=torch.Generator().manual_seed(3)) * 4.0 + 2.0, -1)[0]
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
    g = torch.Generator().manual_seed(7 * n_rays + S)
    rays = orc.synthetic_rays(n_rays, seed=S + 1)
    z = torch.sort(torch.rand(n_rays, S, generator=g) * 4.0 + 2.0, -1)[0]
    d_raw = torch.randn(n_rays, S, 4, generator=g)
    P64 = {k: v.double().requires_grad_(True) for k, v in Pf.items()}
    pts = rays[:, None, 0:3] + rays[:, None, 3:6] * z[..., None]
    # ReLU kinks: a unit whose pre-activation is within fp32 rounding of 0 legitimately lands on either side
    # (about 1 unit per million); such points get no upstream gradient so the comparison is well defined.
    with torch.no_grad():
        feats = torch.cat([orc.posenc(pts.reshape(-1, 3).double(), 10),
                           orc.posenc(rays[:, None, 8:11].expand(n_rays, S, 3).reshape(-1, 3).double(), 4)], -1)
        _, hidden, feat, hv = orc.field_mlp({k: v.detach() for k, v in P64.items()}, feats, return_hidden=True)
        lin = torch.nn.functional.linear
        pre_min = torch.full((feats.shape[0],), float("inf"), dtype=torch.float64)
        h_in = feats[:, :63]
        for i in range(8):
            pre = lin(h_in, P64[f"pts_linears.{i}.weight"].detach(), P64[f"pts_linears.{i}.bias"].detach())
            pre_min = torch.minimum(pre_min, pre.abs().min(-1)[0])
            h_in = torch.relu(pre)
            if i == 4:
                h_in = torch.cat([feats[:, :63], h_in], -1)
        pre = lin(torch.cat([feat, feats[:, 63:]], -1), P64["views_linears.0.weight"].detach(), P64["views_linears.0.bias"].detach())
        pre_min = torch.minimum(pre_min, pre.abs().min(-1)[0])
        risky = (pre_min < 2e-5).reshape(n_rays, S)          # fp32 pre-activation error is ~1e-6
    d_raw[risky] = 0.0
    raw, act = npa.hip_backend.field_fwd(nf.packed_params(), rays.to(dev), z.to(dev), save_act=True)
    grad = torch.full((595844,), float("nan"), device=dev)
    npa.hip_backend.field_bwd(nf.packed_params(), act, d_raw.to(dev), grad, accumulate=False)
    ref = orc.query_field(P64, pts.double(), rays[:, 8:11].double())
    (ref * d_raw.double()).sum().backward()
    grad = grad.cpu()
    assert not torch.isnan(grad).any(), "wgrad left parts of the gradient vector unwritten"
    worst = {}
    for nm, off, shape in npa.hip_backend.param_table():
        gg = grad[off:off + int(np.prod(shape))].view(shape)
        r = P64[nm].grad
        worst[nm] = maxdiff(gg, r) / max(float(r.abs().max()), 1e-30)
    print("kink-adjacent points excluded:", int(risky.sum()), "of", risky.numel(),
          "| field_bwd max|err|/max|grad| per tensor:", {k: f"{v:.1e}" for k, v in worst.items()})
    assert risky.float().mean() < 0.15
    assert max(worst.values()) <= 1e-4, worst
    # accumulate=True adds
    npa.hip_backend.field_bwd(nf.packed_params(), act, d_raw.to(dev), grad_dev := grad.to(dev), accumulate=True)
    assert maxdiff(grad_dev, 2 * grad) <= 1e-3 * float(grad.abs().max())


@pytest.mark.parametrize("n_rays,S", [(48, 64), (11, 192)])
def test_field_backward_no_exclusion(npa, dev, nets, n_rays, S):
    """The same comparison with NO point excluded: ReLU units within rounding of zero may take the other side of the
    kink (about one unit per million in fp32), so the bound is on each tensor's direction and norm, not on every entry."""
    nc, nf, Pc, Pf = nets
    g = torch.Generator().manual_seed(7 * n_rays + S)
    rays = orc.synthetic_rays(n_rays, seed=S + 1)
    z = torch.sort(torch.rand(n_rays, S, generator=g) * 4.0 + 2.0, -1)[0]
    d_raw = torch.randn(n_rays, S, 4, generator=g)
    raw, act = npa.hip_backend.field_fwd(nf.packed_params(), rays.to(dev), z.to(dev), save_act=True)
    grad = torch.full((595844,), float("nan"), device=dev)
    npa.hip_backend.field_bwd(nf.packed_params(), act, d_raw.to(dev), grad, accumulate=False)
    P64 = {k: v.double().requires_grad_(True) for k, v in Pf.items()}
    pts = rays[:, None, 0:3] + rays[:, None, 3:6] * z[..., None]
    (orc.query_field(P64, pts.double(), rays[:, 8:11].double()) * d_raw.double()).sum().backward()
    grad = grad.cpu().double()
    rel, cosdef = {}, {}
    for nm, off, shape in npa.hip_backend.param_table():
        gg, r = grad[off:off + int(np.prod(shape))].view(shape), P64[nm].grad
        rel[nm] = float((gg - r).norm() / r.norm())
        cosdef[nm] = 1.0 - float((gg * r).sum() / (gg.norm() * r.norm()))
    wk = max(rel, key=rel.get)
    print("no exclusion: worst relative L2 error", rel[wk], "on", wk, "worst cosine deficit", max(cosdef.values()))
    # one unit taking the other side of its kink at one of ~3000 points moves a small tensor by ~1e-3 of its norm
    # (measured: 1.1e-3 / 8e-7 on the two cases); the direction is unaffected to 6e-7
    assert max(rel.values()) <= 5e-3, rel
    assert max(cosdef.values()) <= 1e-5, cosdef


def test_raw2outputs_weight_and_depth_gradients(npa, dev):
    """All five outputs of raw2outputs carry gradients to raw in the reference (a depth / weight-sparsity loss term):
    d_weights and d_depth through the HIP adjoint vs fp64 autograd of the oracle."""
    n, S = 70, 96
    g = torch.Generator().manual_seed(4)
    raw = torch.randn(n, S, 4, generator=g) * 3.0
    z = torch.sort(torch.rand(n, S, generator=g) * 4.0 + 2.0, -1)[0]
    rays_d = torch.randn(n, 3, generator=g)
    cw, cd, crgb = torch.randn(n, S, generator=g), torch.randn(n, generator=g), torch.randn(n, 3, generator=g)
    raw64 = raw.double().requires_grad_(True)
    rgb, disp, acc, w, depth = orc.composite(raw64, z.double(), rays_d.double(), None, True)
    ((w * cw.double()).sum() + (depth * cd.double()).sum() + (rgb * crgb.double()).sum()).backward()
    rawg = raw.to(dev).requires_grad_(True)
    o = npa.raw2outputs(rawg, z.to(dev), rays_d.to(dev), 0.0, True)
    ((o[3] * cw.to(dev)).sum() + (o[4] * cd.to(dev)).sum() + (o[0] * crgb.to(dev)).sum()).backward()
    scale = float(raw64.grad.abs().max())
    assert maxdiff(rawg.grad, raw64.grad) <= 2e-5 * max(1.0, scale), (maxdiff(rawg.grad, raw64.grad), scale)
    # weights-only loss (no rgb term at all)
    rawg2 = raw.to(dev).requires_grad_(True)
    (npa.raw2outputs(rawg2, z.to(dev), rays_d.to(dev), 0.0, True)[3] * cw.to(dev)).sum().backward()
    raw64b = raw.double().requires_grad_(True)
    (orc.composite(raw64b, z.double(), rays_d.double(), None, True)[3] * cw.double()).sum().backward()
    assert maxdiff(rawg2.grad, raw64b.grad) <= 2e-5 * max(1.0, float(raw64b.grad.abs().max()))


@pytest.mark.parametrize("datapath", BOUNDARY_DATAPATHS, indirect=True)
def test_render_rays_raw_output_carries_gradients(npa, dev, datapath):
    """extras['raw'] is differentiable in the reference (a sigma regulariser on raw[..., 3] trains the networks)."""
    Pc, Pf = orc.scene_params(seed=3)
    kw = dict(D=8, W=256, input_ch=63, output_ch=5, skips=[4], input_ch_views=27, use_viewdirs=True)
    nc, nf = npa.NeRF(**kw).to(dev), npa.NeRF(**kw).to(dev)
    nc.load_state_dict(Pc); nf.load_state_dict(Pf)
    rays = orc.synthetic_rays(40, seed=8)
    target = torch.rand(40, 3, generator=torch.Generator().manual_seed(1))
    out = npa.render_rays(rays.to(dev), nc, None, 64, N_importance=128, network_fine=nf, white_bkgd=True, retraw=True)
    loss = npa.img2mse(out["rgb_map"], target.to(dev)) + 1e-3 * torch.relu(out["raw"][..., 3]).mean()
    loss.backward()
    Pc_g = {k: v.clone().double().requires_grad_(True) for k, v in Pc.items()}
    Pf_g = {k: v.clone().double().requires_grad_(True) for k, v in Pf.items()}
    ref = orc.trace_rays(rays.double(), Pc_g, Pf_g, 64, 128, white_bkgd=True, retraw=True)
    (orc.mse(ref["rgb_map"], target.double()) + 1e-3 * torch.relu(ref["raw"][..., 3]).mean()).backward()
    stable = ~orc.endpoint_unstable(ref["_weights0"])
    assert stable.all(), "pick another seed: this test wants no endpoint-unstable ray"
    for k, p in nf.named_parameters():
        r = Pf_g[k].grad
        assert maxdiff(p.grad, r) <= BOUNDARY_TOL[datapath]["grad"] * float(r.abs().max()) + 1e-9, (k, maxdiff(p.grad, r), float(r.abs().max()))
    assert all(p.grad is None or float(p.grad.abs().max()) == 0.0 for p in nc.parameters())   # coarse net: rgb0 not in this loss


@pytest.mark.parametrize("datapath", BOUNDARY_DATAPATHS, indirect=True)
def test_second_backward_fails_loudly(npa, dev, nets, datapath):
    nc, nf, Pc, Pf = nets
    rays = orc.synthetic_rays(8, seed=1).to(dev)
    out = npa.render_rays(rays, nc, None, 64, N_importance=128, network_fine=nf, white_bkgd=True)
    loss = out["rgb_map"].sum()
    loss.backward(retain_graph=True)
    with pytest.raises(RuntimeError, match="already consumed"):
        loss.backward()
    pts = torch.randn(5, 3, device=dev)
    vd = torch.nn.functional.normalize(torch.randn(5, 3, device=dev), dim=-1)
    y = npa.query_points(nf, pts, vd).sum()
    y.backward(retain_graph=True)
    with pytest.raises(RuntimeError, match="already consumed"):
        y.backward()
    for m in (nc, nf):
        m.zero_grad()


def test_packed_cache_invalidation_after_raw_writes(npa, dev):
    """Writers that do not advance tensor version counters (c10d broadcast / all_reduce, `.data`, raw pointers) must be
    followed by NeRF.invalidate_packed() (parallel.broadcast_parameters does it): otherwise the kernels keep evaluating
    the previous fragment repack."""
    Pc, _ = orc.scene_params(seed=7)
    kw = dict(D=8, W=256, input_ch=63, output_ch=5, skips=[4], input_ch_views=27, use_viewdirs=True)
    net = npa.NeRF(**kw).to(dev)
    net.load_state_dict(Pc)
    pts = torch.randn(32, 3, device=dev)
    vd = torch.nn.functional.normalize(torch.randn(32, 3, device=dev), dim=-1)
    with torch.no_grad():
        before = npa.query_points(net, pts, vd).clone()
        flat = net.flat_params()
        versions = (flat._version, tuple(p._version for p in net.parameters()))
        flat.data.mul_(1.01)                                  # `.data` write: no version counter moves
        assert versions == (flat._version, tuple(p._version for p in net.parameters()))
        stale = npa.query_points(net, pts, vd)
        assert torch.equal(stale, before), "expected the cached repack to be (wrongly) reused without invalidation"
        net.invalidate_packed()
        fresh = npa.query_points(net, pts, vd)
        net2 = npa.NeRF(**kw).to(dev)
        net2.load_state_dict(net.state_dict())
        assert float((fresh - before).abs().max()) > 1e-4 and torch.equal(fresh, npa.query_points(net2, pts, vd))


# ---------------------------------------------------------------- end to end vs golden (reference-produced)
# Stated tolerances of the two parity datapaths against numbers produced by the REAL reference (fp32, CPU).
#   fp32   : every per-ray quantity within max(floor, 10 x the reference's own fp32-vs-fp64 rounding noise on the same
#            inputs, stored per quantity in the fixture).
#   bf16x3 : products carry ~1e-5 relative error (16-17 significand bits per operand).  Coarse-pass quantities (no
#            hierarchical sampling upstream) are held to absolute per-ray bounds 100x the fp32 ones.  The fine pass
#            inherits sample_pdf's conditioning (helpers:234-236 divides by ~1e-5 in bins the coarse pass found empty; the
#            reference's own fp32-vs-fp64 runs differ by 1e-4..1e-2 there), so the few samples drawn in such bins move
#            by a fraction of a bin under ANY perturbation of the coarse weights: fine quantities are held to a bound on
#            95 % of the rays plus an image-level bound (PSNR between our image and the reference's), which is the
#            north-star's form of the criterion ("reduced-precision variants are judged on PSNR").
GOLD_TOL = {
    "fp32": dict(coarse=1e-5, fine_floor=1e-5, fine_stat="max", fine_max=None, disp_rel=2e-5, zstd_floor=1e-4, raw_floor=5e-4,
                 loss_floor=2e-6, grad=2e-4, grad_max=1e-3, img_psnr_db=85.0),
    # measured on MI355X (round 2): coarse 4e-5..1.1e-4; fine p95 2e-5..6e-5, fine max 6e-4..1.4e-3; image PSNR vs the
    # reference 84.5..102 dB; gradients 3e-4..4.8e-3 of max|g|
    "bf16x3": dict(coarse=3e-4, fine_floor=3e-4, fine_stat="p95", fine_max=5e-3, disp_rel=1e-3, zstd_floor=1e-3, raw_floor=2e-2,
                   loss_floor=2e-5, grad=2e-2, grad_max=2e-2, img_psnr_db=75.0),
    # fp16 three-term split (round 4): products ~2^-22 (4 x fp32's rounding), operands of the weight-gradient GEMM stored with
    # 11 bits.  Coarse-pass quantities sit at fp32-class bounds; the fine pass keeps the p95 + worst-ray + image form because
    # sample_pdf's conditioning amplifies ANY perturbation of the coarse weights (the reference's own fp32-vs-fp64 distance is
    # the `10 x noise` term)
    # round 5: gradient bound tightened to what is measured.  Worst sampled gradient entry / max|g| per fixture, fp16x3 (fp32 datapath
    # in brackets): cfg2 4.7e-4 (7.5e-4), cfg3 1.8e-4 (2.1e-4), lego_det 1.8e-4 (1.3e-4), fern_train 1.4e-4 (9e-5), coarse_only 1.4e-4
    # (1.4e-4), fern_ndc_train 1.3e-3 (9e-5), lego_train / lego_render_train 3.9e-3 (9.3e-4).  The last two exceed this bound and are
    # admitted by the fixture's own `10 x noise` term only (the reference's fp32-vs-fp64 distance of that entry is 4e-4 of max|g|:
    # hierarchical samples that move under ANY rounding); everything else is held to 1.5e-3 (rounds 3-4: 3e-3)
    "fp16x3": dict(coarse=3e-5, fine_floor=3e-5, fine_stat="p95", fine_max=5e-3, disp_rel=1e-4, zstd_floor=3e-4, raw_floor=2e-3,
                   loss_floor=5e-6, grad=1.5e-3, grad_max=1.5e-3, img_psnr_db=85.0),
}


def _golden_randoms(seed, n, args):
    """replay the CPU generator stream the reference consumed (run_nerf.py:371, :285, helpers:208, :285)"""
    if seed is None:
        return None
    torch.manual_seed(seed)
    n_f = args["N_importance"]
    randoms = {}
    if args["perturb"] > 0:
        randoms["t_rand"] = torch.rand(n, 64)
    if args["raw_noise_std"] > 0:
        randoms["noise_c"] = torch.randn(n, 64)
    if n_f > 0 and args["perturb"] > 0:
        randoms["u"] = torch.rand(n, n_f)
    if n_f > 0 and args["raw_noise_std"] > 0:
        randoms["noise_f"] = torch.randn(n, 64 + n_f)
    return randoms


def _check_golden(npa, dev, nets, name, kw, seed, precision="fp32", render=None, n=256, target_seed=99, raw_ray_stride=1, tol=None):
    """HIP render_rays (or, with `render`, the whole render() boundary incl. view directions and the NDC warp) + loss +
    backward vs numbers produced by the real reference (fp32, CPU); tolerances: GOLD_TOL[precision].  n / target_seed / raw_ray_stride:
    the round-4 fixtures at BASELINE's batch size (4096 rays; `raw` stored for every 16th ray)."""
    nc, nf, Pc, Pf = nets
    T = dict(GOLD_TOL[precision], **(tol or {}))
    gold = np.load(f"{GOLD}/{name}.npz")
    target = torch.tensor(np.random.RandomState(target_seed).rand(n, 3), dtype=torch.float32)
    args = dict(N_samples=64, retraw=True, N_importance=128, network_fine=nf, perturb=0., white_bkgd=True,
                raw_noise_std=0., lindisp=False)
    args.update(kw)
    n_f = args["N_importance"]
    randoms = _golden_randoms(seed, n, args)
    for m in (nc, nf):
        m.zero_grad()
    npa.set_precision(precision)
    try:
        if render is None:
            rays = orc.synthetic_rays(n, seed=7)
            assert abs(float(rays.double().abs().sum()) - float(gold["rays_checksum"])) < 1e-6
            out = npa.render_rays(rays.to(dev), nc, None, randoms=randoms, **args)
        else:
            cfg, batch = render
            assert abs(float(batch.double().abs().sum()) - float(gold["rays_checksum"])) < 1e-6
            rgb, disp, acc, extras = npa.render(cfg["H"], cfg["W"], orc.intrinsics(cfg), chunk=1024 * 32, rays=batch.to(dev),
                                                ndc=cfg["ndc"], near=cfg["near"], far=cfg["far"], use_viewdirs=True,
                                                network_fn=nc, network_query_fn=None, randoms=randoms, **args)
            out = dict(extras, rgb_map=rgb, disp_map=disp, acc_map=acc)
            rays = orc.assemble_render_rays(cfg["H"], cfg["W"], orc.intrinsics(cfg), batch[0], batch[1], cfg["ndc"],
                                            cfg["near"], cfg["far"])
        loss = npa.img2mse(out["rgb_map"], target.to(dev))
        if "rgb0" in out:
            loss = loss + npa.img2mse(out["rgb0"], target.to(dev))
        loss.backward()
    finally:
        npa.set_precision("fp32")
    out = {k: v.detach().cpu() for k, v in out.items()}
    # rays whose deterministic u == 1.0 sample is rounding-dependent in the reference itself
    stable = torch.ones(n, dtype=torch.bool)
    if n_f > 0 and args["perturb"] == 0.:
        o = orc.trace_rays(rays, Pc, Pf, 64, n_f, perturb=0., white_bkgd=args["white_bkgd"], lindisp=args["lindisp"])
        stable = ~orc.endpoint_unstable(o["_weights0"])
        assert stable.float().mean() > 0.8
    report, fails = {"unstable_rays": int((~stable).sum())}, []
    noise = lambda k: float(gold["noise/" + k]) if ("noise/" + k) in gold.files else 0.0

    def check(key, err, tol):
        report[key] = (float(err), float(tol))
        if not float(err) <= tol:
            fails.append(key)

    def per_ray(a, b):      # worst component of each ray, NaN == NaN
        d = (a.double() - b.double()).abs()
        d = d.masked_fill(torch.isnan(a) & torch.isnan(b), 0.0).nan_to_num(nan=float("inf"))
        return d.reshape(d.shape[0], -1).max(-1)[0]

    def stat(err, fine):
        if err.numel() == 0:
            return 0.0
        if fine and T["fine_stat"] == "p95":
            return float(torch.quantile(err, 0.95))
        return float(err.max())
    everything = torch.ones(n, dtype=torch.bool)
    fine_keys = ("rgb_map", "acc_map", "disp_map", "z_std", "raw") if n_f > 0 else ()
    for k in ("rgb0", "acc0", "rgb_map", "acc_map"):
        if k in gold.files:
            fine = k in fine_keys
            sel = stable if fine else everything
            err = per_ray(out[k][sel], torch.tensor(gold[k])[sel])
            check(k, stat(err, fine), max(T["fine_floor"], 10 * noise(k)) if fine else T["coarse"])
            report[k + " max"] = float(err.max()) if err.numel() else 0.0
            if fine and T["fine_max"] is not None:
                check(k + " worst ray", report[k + " max"], max(T["fine_max"], 10 * noise(k)))
    for k in ("disp_map", "disp0"):
        if k in gold.files:
            fine = k in fine_keys
            sel = stable if fine else everything
            a, b = out[k][sel], torch.tensor(gold[k])[sel]
            acc_k = torch.tensor(gold["acc_map" if k == "disp_map" else ("acc0" if "acc0" in gold.files else "acc_map")])[sel]
            if precision == "fp32":
                ok = ~(torch.isnan(a) & torch.isnan(b))      # exact datapath: same empty rays (disp = NaN there, reference quirk)
            else:
                # disp = 1 / max(1e-10, depth / acc) is discontinuous at acc -> 0 (NaN on an empty ray, ~1e10 on a ray whose
                # only opacity is one sample with sigma ~ 0+): a reduced-precision datapath is compared where it is defined
                ok = acc_k > 1e-3
                report[k + " rays compared"] = int(ok.sum())
            check(k, stat(per_ray(a[ok], b[ok]), fine), max(T["disp_rel"] * float(b[ok].abs().max()), 10 * noise(k)))
    if "z_std" in gold.files:
        check("z_std", stat(per_ray(out["z_std"][stable], torch.tensor(gold["z_std"])[stable]), True), max(T["zstd_floor"], 10 * noise("z_std")))
    graw = torch.tensor(gold["raw"])
    sel = stable if n_f > 0 else everything
    check("raw", stat(per_ray(out["raw"][::raw_ray_stride, ::8][sel[::raw_ray_stride]], graw[sel[::raw_ray_stride]]), n_f > 0),
          max(T["raw_floor"] * max(1.0, float(graw.abs().max()) / 10), 10 * noise("raw")))
    # image-level criterion over ALL rays (unstable ones included): PSNR between our image and the reference's
    mse_img = float(((out["rgb_map"].double() - torch.tensor(gold["rgb_map"]).double()) ** 2).mean())
    report["psnr_vs_ref_dB"] = (orc.psnr(max(mse_img, 1e-30)), T["img_psnr_db"])
    if report["psnr_vs_ref_dB"][0] < T["img_psnr_db"]:
        fails.append("psnr_vs_ref_dB")
    check("loss", abs(loss.item() - float(gold["loss"])), max(T["loss_floor"], 10 * noise("rgb_map") * 0.05))
    unstable_slack = 50.0 if (~stable).any() else 1.0
    worst, dots = 0.0, [0.0, 0.0, 0.0]
    for tag, net in (("c", nc), ("f", nf)):
        for nm, p in net.named_parameters():
            key = f"{tag}/{nm}/max"
            if key not in gold.files:
                assert p.grad is None or float(p.grad.abs().max()) == 0.0, (tag, nm)
                continue
            g = p.grad.detach().cpu().reshape(-1)
            idx, val = gold[f"{tag}/{nm}/idx"], gold[f"{tag}/{nm}/val"]
            gmax, gnoise = float(gold[key]), float(gold[f"{tag}/{nm}/noise"])
            got = g[idx].numpy().astype(np.float64)
            err = float(np.abs(got - val).max())
            tol = max(T["grad"] * gmax, 10 * gnoise) * (unstable_slack if tag == "f" else 1.0)
            worst = max(worst, err / max(gmax, 1e-30))
            check(f"grad {tag}/{nm}", err, tol)
            check(f"grad {tag}/{nm} max", abs(float(g.abs().max()) - gmax), max(T["grad_max"] * gmax, 10 * gnoise) * unstable_slack)
            dots[0] += float((got * val).sum()) / gmax ** 2
            dots[1] += float((got * got).sum()) / gmax ** 2
            dots[2] += float((val.astype(np.float64) ** 2).sum()) / gmax ** 2
    report["worst grad err/max"] = worst
    cos = dots[0] / max(np.sqrt(dots[1] * dots[2]), 1e-300)
    check("grad cosine deficit (sampled entries, per-tensor normalised)", 1.0 - cos, 1e-6 if precision == "fp32" else 1e-3)
    print(name, precision, {k: v for k, v in report.items() if not k.startswith("grad ") or k in fails})
    assert not fails, {k: report[k] for k in fails}


# ---- the reduced INFERENCE class (fp16_fp8c: fp16 main term + fp8 correction terms, ~2^-15 per product, no_grad rendering only) against
# the same reference-produced fixtures, forward maps only.  Coarse-pass quantities per ray (measured bound, 10x above the three-term
# fp16 class as the products are 2^7 coarser -- and counted against the fp16x3 bound as well), the fine pass in the p95 + worst-ray +
# image form of the split datapaths, the image at >= 85 dB of the reference's, and the count of FLIP rays: rays whose last sample's
# density changes sign under the datapath's rounding, which dists[-1] = 1e10 (run_nerf.py:277-278) turns into a step of the ray's
# opacity (bf16x3 has three of them among 32,768 rays).  The reduced class evaluates every ray's last sample on the three-term
# products (the guard launch), so the expected count is 0.
REDUCED_TOL = dict(coarse=3e-4, fine_floor=3e-4, fine_max=5e-3, zstd_floor=1e-3, raw_floor=2e-2, disp_rel=1e-3, img_psnr_db=85.0, flip=1e-2)


def _check_golden_forward_reduced(npa, dev, nets, name, kw, seed, render=None, n=256, raw_ray_stride=1, tol=None):
    """no_grad render_rays / render() on the reduced inference class vs the reference's forward maps of fixture `name`"""
    nc, nf, Pc, Pf = nets
    T = dict(REDUCED_TOL, **(tol or {}))
    gold = np.load(f"{GOLD}/{name}.npz")
    args = dict(N_samples=64, retraw=True, N_importance=128, network_fine=nf, perturb=0., white_bkgd=True, raw_noise_std=0., lindisp=False)
    args.update(kw)
    n_f = args["N_importance"]
    randoms = _golden_randoms(seed, n, args)
    npa.set_precision("fp16_fp8c")
    try:
        with torch.no_grad():
            if render is None:
                rays = orc.synthetic_rays(n, seed=7)
                assert abs(float(rays.double().abs().sum()) - float(gold["rays_checksum"])) < 1e-6
                out = npa.render_rays(rays.to(dev), nc, None, randoms=randoms, **args)
            else:
                cfg, batch = render
                assert abs(float(batch.double().abs().sum()) - float(gold["rays_checksum"])) < 1e-4
                H, W, K = (800, 800, orc.intrinsics(dict(orc.LEGO, H=800, W=800, focal=1111.0))) if n > 4096 else (cfg["H"], cfg["W"], orc.intrinsics(cfg))
                rgb, disp, acc, extras = npa.render(H, W, K, chunk=1024 * 32, rays=batch.to(dev), ndc=cfg["ndc"], near=cfg["near"], far=cfg["far"],
                                                    use_viewdirs=True, network_fn=nc, network_query_fn=None, randoms=randoms, **args)
                out = dict(extras, rgb_map=rgb, disp_map=disp, acc_map=acc)
                rays = orc.assemble_render_rays(H, W, K, batch[0], batch[1], cfg["ndc"], cfg["near"], cfg["far"])
    finally:
        npa.set_precision("fp32")
    out = {k: v.detach().cpu() for k, v in out.items()}
    stable = torch.ones(n, dtype=torch.bool)
    if n_f > 0 and args["perturb"] == 0.:
        o = orc.trace_rays(rays, Pc, Pf, 64, n_f, perturb=0., white_bkgd=args["white_bkgd"], lindisp=args["lindisp"])
        stable = ~orc.endpoint_unstable(o["_weights0"])
    noise = lambda k: float(gold["noise/" + k]) if ("noise/" + k) in gold.files else 0.0
    report, fails = {}, []

    def per_ray(a, b):
        d = (a.double() - b.double()).abs()
        d = d.masked_fill(torch.isnan(a) & torch.isnan(b), 0.0).nan_to_num(nan=float("inf"))
        return d.reshape(d.shape[0], -1).max(-1)[0]
    flips = torch.zeros(n, dtype=torch.bool)
    for k in (("rgb0", "acc0") if n_f > 0 else ("rgb_map", "acc_map")):       # the coarse pass (no hierarchical sampling upstream): per ray
        if k not in gold.files:
            continue
        err = per_ray(out[k], torch.tensor(gold[k]))
        report[k] = float(err.max())
        report[k + " rays over the fp16x3 bound"] = int((err > GOLD_TOL["fp16x3"]["coarse"]).sum())
        flips |= err > T["flip"]
        # (with a refining pass the coarse pass runs on the three-term products, render._field_pass: the fp16x3 bound applies)
        if report[k] > (GOLD_TOL["fp16x3"]["coarse"] if n_f > 0 and npa.hip_backend.REDUCED_COARSE_THREE_TERM else T["coarse"]):
            fails.append(k)
    report["flip rays"] = int(flips.sum())
    if report["flip rays"]:
        fails.append("flip rays")
    if n_f > 0:
        for k in ("rgb_map", "acc_map", "z_std"):
            if k not in gold.files:
                continue
            err = per_ray(out[k][stable], torch.tensor(gold[k])[stable])
            floor = T["zstd_floor"] if k == "z_std" else T["fine_floor"]
            report[k + " p95"] = float(torch.quantile(err, 0.95))
            report[k + " max"] = float(err.max())
            if report[k + " p95"] > max(floor, 10 * noise(k)):
                fails.append(k + " p95")
            if report[k + " max"] > max(T["fine_max"], 10 * noise(k), 1e-2 if k == "z_std" else 0.0):
                fails.append(k + " max")
    if "raw" in gold.files:
        graw = torch.tensor(gold["raw"])
        err = per_ray(out["raw"][::raw_ray_stride, ::8][stable[::raw_ray_stride]], graw[stable[::raw_ray_stride]])
        report["raw p95"] = float(torch.quantile(err, 0.95)) if n_f > 0 else float(err.max())
        if report["raw p95"] > max(T["raw_floor"] * max(1.0, float(graw.abs().max()) / 10), 10 * noise("raw")):
            fails.append("raw p95")
    mse_img = float(((out["rgb_map"].double() - torch.tensor(gold["rgb_map"]).double()) ** 2).mean())
    report["psnr_vs_ref_dB"] = orc.psnr(max(mse_img, 1e-30))
    if report["psnr_vs_ref_dB"] < T["img_psnr_db"]:
        fails.append("psnr_vs_ref_dB")
    print(name, "fp16_fp8c", report)
    assert not fails, {k: report.get(k) for k in fails}


PARITY_DATAPATHS = ["fp32", "bf16x3", "fp16x3"]
# round 6: the two-word datapath against the reference-produced goldens at BASELINE's batch sizes (tests/test_gpu_golden_cfg.py) and the
# train()-shaped loop (tests/test_train_loop_gpu.py): its forward is fp16x3's, so it is held to fp16x3's bounds
GOLD_TOL["fp16x3w"] = GOLD_TOL["fp16x3"]
GOLDEN_CFG_DATAPATHS = PARITY_DATAPATHS + ["fp16x3w"]


@pytest.mark.parametrize("precision", PARITY_DATAPATHS)
def test_golden_lego_det(npa, dev, nets, precision):
    _check_golden(npa, dev, nets, "lego_det", {}, None, precision)


def test_golden_lego_det_reduced_inference_class(npa, dev, nets):
    _check_golden_forward_reduced(npa, dev, nets, "lego_det", {}, None)


def test_golden_fern_train_reduced_inference_class(npa, dev, nets):
    _check_golden_forward_reduced(npa, dev, nets, "fern_train", dict(perturb=1.0, raw_noise_std=1.0, white_bkgd=False, N_importance=64, lindisp=True), 321)


def test_golden_coarse_only_reduced_inference_class(npa, dev, nets):
    _check_golden_forward_reduced(npa, dev, nets, "lego_coarse_only", dict(perturb=1.0, N_importance=0, network_fine=None), 11)


@pytest.mark.parametrize("precision", PARITY_DATAPATHS)
def test_golden_lego_train(npa, dev, nets, precision):
    _check_golden(npa, dev, nets, "lego_train", dict(perturb=1.0), 123, precision)


@pytest.mark.parametrize("precision", PARITY_DATAPATHS)
def test_golden_fern_train(npa, dev, nets, precision):
    _check_golden(npa, dev, nets, "fern_train", dict(perturb=1.0, raw_noise_std=1.0, white_bkgd=False, N_importance=64, lindisp=True), 321, precision)


@pytest.mark.parametrize("precision", PARITY_DATAPATHS)
def test_golden_coarse_only(npa, dev, nets, precision):
    _check_golden(npa, dev, nets, "lego_coarse_only", dict(perturb=1.0, N_importance=0, network_fine=None), 11, precision)


@pytest.mark.parametrize("precision", PARITY_DATAPATHS)
def test_golden_fern_ndc_through_render(npa, dev, nets, precision):
    """BASELINE.json configs[2]: the reference's render(H=378, W=504, K(focal 407.5), rays=..., ndc=True, near=0, far=1,
    use_viewdirs=True, perturb=1, raw_noise_std=1, white_bkgd=False, N_importance=128) on forward-facing rays (d_z < 0)
    -- run_nerf.py:110-123, run_nerf_helpers.py:175-192, configs/fern.txt -- vs npa.render() with the same arguments."""
    _check_golden(npa, dev, nets, "fern_ndc_train", dict(perturb=1.0, raw_noise_std=1.0, white_bkgd=False, N_importance=128), 77,
                  precision, render=(orc.FERN, orc.fern_batch(256, seed=3)))


@pytest.mark.parametrize("precision", PARITY_DATAPATHS)
def test_golden_lego_through_render(npa, dev, nets, precision):
    """BASELINE.json configs[1] through the rays=... branch of render() (run_nerf.py:95-134), as train() calls it (:760)."""
    _check_golden(npa, dev, nets, "lego_render_train", dict(perturb=1.0), 123, precision,
                  render=(orc.LEGO, orc.lego_batch(256, seed=7)))


# ---------------------------------------------------------------- the north-star acceptance gate
GATE_FLOOR_DB = {"fp32": 110.0, "bf16x3": 90.0, "fp16x3": 100.0, "fp16_fp8c": 85.0}     # PSNR(our image, reference image); measured 120..134 / 96.6..104 dB


def _gate(npa, dev, nets, which, precision):
    nc, nf, Pc, Pf = nets
    cfg = orc.LEGO if which == "lego" else orc.FERN
    batch = orc.lego_batch(1024, seed=31) if which == "lego" else orc.fern_batch(1024, seed=32)
    gold = np.load(f"{GOLD}/gate_{which}.npz")
    assert abs(float(batch.double().abs().sum()) - float(gold["rays_checksum"])) < 1e-5
    npa.set_precision(precision)
    try:
        with torch.no_grad():
            rgb, _, _, _ = npa.render(cfg["H"], cfg["W"], orc.intrinsics(cfg), chunk=1024 * 32, rays=batch.to(dev), ndc=cfg["ndc"],
                                      near=cfg["near"], far=cfg["far"], use_viewdirs=True, network_fn=nc, network_query_fn=None,
                                      N_samples=64, N_importance=128, network_fine=nf, perturb=0., raw_noise_std=0.,
                                      white_bkgd=cfg["white_bkgd"])
    finally:
        npa.set_precision("fp32")
    return orc.precision_gate(rgb, torch.tensor(gold["rgb_ref"]), torch.tensor(gold["target"]))


@pytest.mark.parametrize("precision", ["fp32", "bf16x3", "fp16x3", "fp16_fp8c"])
@pytest.mark.parametrize("which", ["lego", "fern"])
def test_precision_gate_psnr_on_a_teacher_target(npa, dev, nets, which, precision):
    """north_star: 'PSNR delta < 0.01 dB'.  Target = the image of a teacher scene (workloads.teacher_params) rendered by
    the REAL reference; the networks under test sit at 31 dB (lego-like) / 37 dB (fern-like, NDC) from it -- trained-NeRF
    territory, where an rgb error of 1e-3 moves the PSNR by ~0.1 dB (a uniform-random target at ~6 dB would hide it).
    Both images of the fixture come from the reference's render() (tests/golden/make_golden.py --round2)."""
    g = _gate(npa, dev, nets, which, precision)
    print(which, precision, g)
    assert g["target_psnr_db"] >= 30.0, g
    assert g["psnr_delta_db"] < 0.01, g
    assert g["psnr_vs_ref_db"] >= GATE_FLOOR_DB[precision], g


def test_precision_gate_can_fail(npa, dev, nets):
    """The gate is not vacuous: an image off by a uniform 2e-3 (a plain-bf16-sized error) violates the 0.01 dB bar."""
    gold = np.load(f"{GOLD}/gate_lego.npz")
    ref, tgt = torch.tensor(gold["rgb_ref"]), torch.tensor(gold["target"])
    bad = orc.precision_gate(ref + 2e-3 * torch.sign(ref - tgt), ref, tgt)
    assert bad["psnr_delta_db"] > 0.01 and bad["psnr_vs_ref_db"] < GATE_FLOOR_DB["bf16x3"], bad
    # an error well inside the 0.01 dB bar but larger than the split-bf16 datapath's is still caught by the image floor
    mild = orc.precision_gate(ref + 1e-4 * torch.sign(ref - tgt), ref, tgt)
    assert mild["psnr_delta_db"] < 0.01 and mild["psnr_vs_ref_db"] < GATE_FLOOR_DB["bf16x3"], mild
    ok = orc.precision_gate(ref, ref, tgt)
    assert ok["psnr_delta_db"] == 0.0 and abs(ok["target_psnr_db"] - float(gold["target_psnr_db"])) < 1e-6


@pytest.mark.parametrize("precision", PARITY_DATAPATHS)
def test_render_c2w_ndc_matches_rays_branch_and_oracle(npa, dev, nets, precision):
    """render(c2w=..., ndc=True) (one HIP launch builds the NDC ray records) == render(rays=get_rays(...), ndc=True)
    == the oracle on the same image (run_nerf.py:95-123)."""
    nc, nf, Pc, Pf = nets
    H, W, focal = 10, 14, 11.0
    K = np.array([[focal, 0, 0.5 * W], [0, focal, 0.5 * H], [0, 0, 1]])
    c2w = orc.fern_poses(3)[1]
    kw = dict(network_fn=nc, network_query_fn=None, N_samples=64, N_importance=128, network_fine=nf, perturb=0.,
              white_bkgd=False, raw_noise_std=0., retraw=True, ndc=True, near=0., far=1., use_viewdirs=True)
    npa.set_precision(precision)
    try:
        with torch.no_grad():
            a = npa.render(H, W, K, chunk=64, c2w=c2w.to(dev), **kw)
            ro, rd = npa.get_rays(H, W, K, c2w.to(dev))
            b = npa.render(H, W, K, chunk=1 << 15, rays=torch.stack([ro.reshape(-1, 3), rd.reshape(-1, 3)], 0), **kw)
    finally:
        npa.set_precision("fp32")
    assert a[0].shape == (H, W, 3) and b[0].shape == (H * W, 3)
    o, d = orc.pinhole_rays(H, W, K, c2w)
    flat = orc.assemble_render_rays(H, W, K, o.reshape(-1, 3), d.reshape(-1, 3), True, 0., 1.)
    ref = orc.trace_rays(flat, Pc, Pf, 64, 128, perturb=0., white_bkgd=False)
    stable = ~orc.endpoint_unstable(ref["_weights0"])
    tol0 = 1e-5 if precision == "fp32" else 1e-3
    assert maxdiff(a[3]["rgb0"].reshape(-1, 3), ref["rgb0"]) <= tol0
    assert maxdiff(b[3]["rgb0"], ref["rgb0"]) <= tol0
    assert maxdiff(a[3]["rgb0"].reshape(-1, 3), b[3]["rgb0"]) <= 2e-6 + (0 if precision == "fp32" else 1e-4)
    ref64 = orc.trace_rays(flat.double(), {k: v.double() for k, v in Pc.items()}, {k: v.double() for k, v in Pf.items()},
                           64, 128, perturb=0., white_bkgd=False)
    floor = maxdiff(ref["rgb_map"][stable], ref64["rgb_map"][stable])
    lim = max(1e-5, 10 * floor) if precision == "fp32" else max(3e-3, 10 * floor)
    assert float(torch.quantile((a[0].reshape(-1, 3).cpu()[stable] - ref["rgb_map"][stable]).abs().max(-1)[0], 0.95)) <= lim


# ---------------------------------------------------------------- split-bf16 datapath (precision "bf16x3")
@pytest.mark.parametrize("n_rays,S", [(64, 64), (37, 192), (5, 3)])
def test_field_forward_bf16x3(npa, dev, nets, n_rays, S):
    """W*x = W_hi*x_hi + W_hi*x_lo + W_lo*x_hi on bf16 MFMA: ~1e-5 relative per product (fp32: 6e-8, bf16: 4e-3); what is saved
    for the backward: bf16 roundings of the activations (rows in 16-point tiles, encodings in 32-point tiles) and ReLU bitmasks that
    are exactly the signs of the saved rows."""
    nc, nf, Pc, Pf = nets
    rays = orc.synthetic_rays(n_rays, seed=S)
    z = torch.sort(torch.rand(n_rays, S, generator=torch.Generator().manual_seed(S)) * 4.0 + 2.0, -1)[0]
    raw, _ = npa.hip_backend.field_fwd(nf.packed_params("bf16x3"), rays.to(dev), z.to(dev), save_act=False, precision="bf16x3")
    raw32, _ = npa.hip_backend.field_fwd(nf.packed_params("fp32"), rays.to(dev), z.to(dev), save_act=False)
    pts = rays[:, None, 0:3] + rays[:, None, 3:6] * z[..., None]
    ref64 = orc.query_field({k: v.double() for k, v in Pf.items()}, pts.double(), rays[:, 8:11].double())
    scale = max(1.0, float(ref64.abs().max()))
    e3, e32 = maxdiff(raw, ref64), maxdiff(raw32, ref64)
    print(f"bf16x3 max|raw-ref64| = {e3:.2e} (fp32 kernel: {e32:.2e}) at |raw|max = {scale:.1f}")
    assert e3 <= 3e-4 * scale, (e3, scale)
    raw_s, act = npa.hip_backend.field_fwd(nf.packed_params("bf16x3"), rays.to(dev), z.to(dev), save_act=True, precision="bf16x3")
    assert torch.equal(raw, raw_s)          # inference and the saving forward are the same kernel
    assert npa.hip_backend.buffer_layout(act)[0] == 4
    P = n_rays * S
    feats = torch.cat([orc.posenc(pts.reshape(-1, 3), 10), orc.posenc(rays[:, None, 8:11].expand(n_rays, S, 3).reshape(-1, 3), 4)], -1)
    _, hidden, feat, hv = orc.field_mlp(Pf, feats, return_hidden=True)
    rows = lambda region: npa.hip_backend.saved_rows(act, n_rays, S, region, "bf16x3").cpu()
    bf = 2.0 ** -8                              # bf16 rounding of the saved values
    for l in range(8):
        assert maxdiff(rows(f"h{l}"), hidden[l]) <= (bf + 3e-4) * max(1.0, float(hidden[l].abs().max())), l
    # (`feature` is not saved by this datapath: feature_linear is folded into the view branch, csrc/nerf_common.h)
    assert maxdiff(rows("hv"), hv) <= (bf + 3e-4) * max(1.0, float(hv.abs().max()))
    assert maxdiff(rows("enc")[:, :63], feats[:, :63]) <= bf * float(feats[:, :63].abs().max()) + 5e-6
    # the ReLU bitmasks the backward reads must be exactly the signs of the rows saved next to them: word (layer, p,
    # half), bit i <-> feature 32*(i>>4) + d32row(i&15, half) (csrc/nerf_common.h)
    i = torch.arange(128)
    feat_of = lambda half: 32 * (i >> 4) + ((i & 15) & 3) + 8 * ((i & 15) >> 2) + 4 * half
    words = npa.hip_backend.saved_masks(act, n_rays, S, "bf16x3").cpu().view(9, P, 2, 4)
    for layer, region, width in [(l, f"h{l}", 256) for l in range(8)] + [(8, "hv", 128)]:
        pos = rows(region) > 0
        for half in range(2):
            bits = ((words[layer, :, half, :, None] >> torch.arange(32)) & 1).reshape(P, 128).bool()
            n = width // 2
            assert torch.equal(bits[:, :n], pos[:, feat_of(half)[:n]]), (layer, half)
    npa.hip_backend.WORKSPACE.give(act)


@pytest.mark.parametrize("n_rays,S", [(48, 64), (11, 192), (3, 5), (1, 1)])      # (3, 5), (1, 1): odd point counts (ragged lane pairs / tiles)
def test_field_backward_bf16x3(npa, dev, nets, n_rays, S):
    """Split-bf16 forward + dgrad + wgrad vs fp64 autograd.  Besides the ~1e-5 product error, ReLU units whose
    pre-activation lies within the forward's ~1e-4 error of zero pick the other side of the kink (a few units per
    point out of 2176), which moves a gradient by up to ~1e-2 of its max while its direction is unchanged
    (cosine >= 0.9999): that is the stated tolerance of this datapath."""
    nc, nf, Pc, Pf = nets
    g = torch.Generator().manual_seed(7 * n_rays + S)
    rays = orc.synthetic_rays(n_rays, seed=S + 1)
    z = torch.sort(torch.rand(n_rays, S, generator=g) * 4.0 + 2.0, -1)[0]
    d_raw = torch.randn(n_rays, S, 4, generator=g)
    packed3 = nf.packed_params("bf16x3")
    raw, act = npa.hip_backend.field_fwd(packed3, rays.to(dev), z.to(dev), save_act=True, precision="bf16x3")
    assert npa.hip_backend.buffer_layout(act)[0] == 4         # the library's own record of what it wrote (nerf_buffer_layout)
    grad = torch.full((595844,), float("nan"), device=dev)
    npa.hip_backend.field_bwd(packed3, act, d_raw.to(dev), grad, accumulate=False, precision="bf16x3", params=nf.flat_params())
    # accumulate=True adds (also through the fold kernel that produces dWf, dbf and dWv[:, :256])
    g2 = grad.clone()
    npa.hip_backend.field_bwd(packed3, act, d_raw.to(dev), g2, accumulate=True, precision="bf16x3", params=nf.flat_params())
    assert maxdiff(g2, 2 * grad) <= 1e-3 * float(grad.abs().max())
    with pytest.raises(npa.hip_backend.NerfHipError):
        npa.hip_backend.field_bwd(packed3, act, d_raw.to(dev), g2, accumulate=False, precision="bf16x3")       # params required
    P64 = {k: v.double().requires_grad_(True) for k, v in Pf.items()}
    pts = rays[:, None, 0:3] + rays[:, None, 3:6] * z[..., None]
    ref = orc.query_field(P64, pts.double(), rays[:, 8:11].double())
    (ref * d_raw.double()).sum().backward()
    grad = grad.cpu()
    assert not torch.isnan(grad).any()
    worst, cos = {}, {}
    for nm, off, shape in npa.hip_backend.param_table():
        gg = grad[off:off + int(np.prod(shape))].view(shape).double()
        r = P64[nm].grad
        worst[nm] = maxdiff(gg, r) / max(float(r.abs().max()), 1e-30)
        cos[nm] = float((gg * r).sum() / (gg.norm() * r.norm() + 1e-30))
    print("bf16x3 bwd max|err|/max|grad|:", {k: f"{v:.1e}" for k, v in worst.items()}, "min cosine:", min(cos.values()))
    assert max(worst.values()) <= 5e-2, worst
    assert min(cos.values()) >= 0.9999, cos


def test_bf16x3_training_step_tracks_fp32(npa, dev):
    """Three Adam steps in the bf16x3 datapath stay on the fp32 datapath's loss trajectory."""
    Pc, Pf = orc.scene_params(seed=2)
    kw = dict(D=8, W=256, input_ch=63, output_ch=5, skips=[4], input_ch_views=27, use_viewdirs=True)
    losses = {}
    rays = orc.synthetic_rays(256, seed=41).to(dev)
    target = torch.rand(256, 3, generator=torch.Generator().manual_seed(9)).to(dev)
    for prec in ("fp32", "bf16x3"):
        nc, nf = npa.NeRF(**kw).to(dev), npa.NeRF(**kw).to(dev)
        nc.load_state_dict(Pc); nf.load_state_dict(Pf)
        opt = torch.optim.Adam(list(nc.parameters()) + list(nf.parameters()), lr=5e-4)
        npa.set_precision(prec)
        try:
            out_l = []
            for step in range(3):
                opt.zero_grad()
                out = npa.render_rays(rays, nc, None, 64, N_importance=128, network_fine=nf, white_bkgd=True)
                loss = npa.img2mse(out["rgb_map"], target) + npa.img2mse(out["rgb0"], target)
                loss.backward()
                opt.step()
                out_l.append(loss.item())
        finally:
            npa.set_precision("fp32")
        losses[prec] = out_l
    print("loss trajectories:", losses)
    for a, b in zip(losses["fp32"], losses["bf16x3"]):
        assert abs(a - b) <= 2e-3 * abs(a), losses


def _flat_grads_through_render(npa, dev, n, seed=17, precision="bf16x3"):
    """gradient of the training loss (teacher-scene target, rendered on the bf16x3 datapath) w.r.t. both networks through
    render() on the datapath `precision`"""
    import workloads as wl
    cfg = wl.LEGO
    Pc, Pf = wl.scene_params()
    Tc, Tf = wl.teacher_params()
    kw = dict(D=8, W=256, input_ch=63, output_ch=5, skips=[4], input_ch_views=27, use_viewdirs=True)
    nets = [npa.NeRF(**kw).to(dev) for _ in range(4)]
    for m, P in zip(nets, (Pc, Pf, Tc, Tf)):
        m.load_state_dict(P)
    batch = wl.lego_batch(n, seed=seed).to(dev)
    rnd = {k: v.to(dev) for k, v in wl.synthetic_randoms(n, 64, 128, seed=seed).items() if k in ("t_rand", "u")}
    args = dict(chunk=1 << 15, ndc=False, near=cfg["near"], far=cfg["far"], use_viewdirs=True, network_query_fn=None,
                N_samples=64, N_importance=128, perturb=1.0, white_bkgd=True, raw_noise_std=0.)
    npa.set_precision("bf16x3")
    try:
        with torch.no_grad():
            target = npa.render(cfg["H"], cfg["W"], wl.intrinsics(cfg), rays=batch, randoms=rnd, network_fn=nets[2],
                                network_fine=nets[3], **args)[0]
        npa.set_precision(precision)
        rgb, _, _, ex = npa.render(cfg["H"], cfg["W"], wl.intrinsics(cfg), rays=batch, randoms=rnd, network_fn=nets[0],
                                   network_fine=nets[1], **args)
        (npa.img2mse(rgb, target) + npa.img2mse(ex["rgb0"], target)).backward()
    finally:
        npa.set_precision("fp32")
    return torch.cat([nets[0].last_flat_grad, nets[1].last_flat_grad]).double().cpu()


def test_bf16x3_render_close_to_oracle_per_ray(npa, dev, nets):
    """bf16x3 inference vs the oracle ray by ray (the image-level PSNR criterion is test_precision_gate_*)."""
    nc, nf, Pc, Pf = nets
    rays = orc.synthetic_rays(512, seed=31)
    npa.set_precision("bf16x3")
    try:
        with torch.no_grad():
            out = npa.render_rays(rays.to(dev), nc, None, 64, N_importance=128, network_fine=nf, white_bkgd=True, retraw=True)
    finally:
        npa.set_precision("fp32")
    ref = orc.trace_rays(rays, Pc, Pf, 64, 128, white_bkgd=True)
    d0 = maxdiff(out["rgb0"], ref["rgb0"])
    err = (out["rgb_map"].cpu() - ref["rgb_map"]).abs().max(-1)[0]
    print(f"bf16x3: max|rgb0 - ref| = {d0:.2e}, rgb_map err median {float(err.median()):.2e} p95 {float(torch.quantile(err, 0.95)):.2e} max {float(err.max()):.2e}")
    assert d0 <= 3e-4                                       # measured 4.6e-5
    assert float(torch.quantile(err, 0.95)) <= 3e-4         # measured 2.4e-5 (worst ray 1.6e-3)
    assert float(err.max()) <= 1e-2


@pytest.mark.parametrize("precision", ["fp32", "bf16x3", "fp16x3"])
@pytest.mark.parametrize("n", [0, 1, 33, 129])
def test_ragged_and_empty_batches(npa, dev, nets, precision, n):
    """Edge cases through the full autograd path: empty batch, one ray, and sizes that leave partially filled
    waves / workgroups (16- and 32-point wave blocks, 128-point workgroups) in both datapaths."""
    nc, nf, Pc, Pf = nets
    rays = orc.synthetic_rays(max(n, 1), seed=50 + n)[:n]
    target = torch.rand(n, 3, generator=torch.Generator().manual_seed(n))
    for m in (nc, nf):
        m.zero_grad()
    npa.set_precision(precision)
    try:
        out = npa.render_rays(rays.to(dev), nc, None, 64, N_importance=128, network_fine=nf, white_bkgd=True, retraw=True)
        assert out["rgb_map"].shape == (n, 3) and out["raw"].shape == (n, 192, 4) and out["z_std"].shape == (n,)
        if n == 0:
            return
        loss = npa.img2mse(out["rgb_map"], target.to(dev)) + npa.img2mse(out["rgb0"], target.to(dev))
        loss.backward()
    finally:
        npa.set_precision("fp32")
    ref = orc.trace_rays(rays, Pc, Pf, 64, 128, white_bkgd=True)
    tol = 1e-5 if precision == "fp32" else 3e-4
    assert maxdiff(out["rgb0"], ref["rgb0"]) <= tol
    stable = ~orc.endpoint_unstable(ref["_weights0"])
    if stable.any():
        assert maxdiff(out["rgb_map"][stable.to(dev)], ref["rgb_map"][stable]) <= max(20 * tol, 2e-3)
    g = nf.last_flat_grad
    assert g is not None and torch.isfinite(g).all() and float(g.abs().max()) > 0


def test_fused_adam_matches_torch_adam(npa, dev):
    """nerf_adam_step (one launch per flat vector) vs torch.optim.Adam over the 24 tensors."""
    kw = dict(D=8, W=256, input_ch=63, output_ch=5, skips=[4], input_ch_views=27, use_viewdirs=True)
    torch.manual_seed(0)
    a, b = npa.NeRF(**kw).to(dev), npa.NeRF(**kw).to(dev)
    b.load_state_dict(a.state_dict())
    oa, ob = npa.FlatAdam(a.parameters(), lr=5e-4), torch.optim.Adam(b.parameters(), lr=5e-4)
    table = npa.hip_backend.param_table()
    for it in range(4):
        fg = (torch.randn(595844, generator=torch.Generator().manual_seed(it)) * 10 ** (it - 2)).to(dev)
        for m in (a, b):
            for (nm, off, shape), p in zip(table, m.param_list()):
                g = fg[off:off + p.numel()].view(shape)
                p.grad = g if m is a else g.clone()
        oa.step()
        ob.step()
    d = maxdiff(a.flat_params(), b.flat_params())
    assert d <= 2e-7, d
    assert maxdiff(oa.state[a.pts_linears[3].weight]["exp_avg_sq"], ob.state[b.pts_linears[3].weight]["exp_avg_sq"]) <= 1e-6 * 1e2


def test_fused_adam_step_reaches_the_kernels(npa, dev):
    """FlatAdam writes the parameters through raw pointers (no autograd version bump): the fragment repack must still
    be refreshed, i.e. the next forward has to see the updated weights -- same outputs as a fresh module that loads them."""
    Pc, Pf = orc.scene_params(seed=4)
    kw = dict(D=8, W=256, input_ch=63, output_ch=5, skips=[4], input_ch_views=27, use_viewdirs=True)
    rays = orc.synthetic_rays(64, seed=12).to(dev)
    target = torch.rand(64, 3, generator=torch.Generator().manual_seed(2)).to(dev)
    for prec in ("fp32", "bf16x3"):
        nc, nf = npa.NeRF(**kw).to(dev), npa.NeRF(**kw).to(dev)
        nc.load_state_dict(Pc); nf.load_state_dict(Pf)
        opt = npa.FlatAdam(list(nc.parameters()) + list(nf.parameters()), lr=1e-2)
        npa.set_precision(prec)
        try:
            render = lambda a, b: npa.render_rays(rays, a, None, 64, N_importance=128, network_fine=b, white_bkgd=True)
            before = render(nc, nf)
            loss = npa.img2mse(before["rgb_map"], target) + npa.img2mse(before["rgb0"], target)
            loss.backward()
            opt.step()
            with torch.no_grad():
                after = render(nc, nf)["rgb_map"]
                nc2, nf2 = npa.NeRF(**kw).to(dev), npa.NeRF(**kw).to(dev)
                nc2.load_state_dict(nc.state_dict()); nf2.load_state_dict(nf.state_dict())
                fresh = render(nc2, nf2)["rgb_map"]
        finally:
            npa.set_precision("fp32")
        assert float((after - before["rgb_map"].detach()).abs().max()) > 1e-3, "the step did not reach the kernels"
        assert torch.equal(after, fresh)


@pytest.mark.parametrize("mode", ["frozen_first_layer", "heads_only", "first_grad_none"])
def test_fused_adam_on_a_subset_of_the_network_reaches_the_kernels(npa, dev, mode):
    """FlatAdam hands the kernel the run of parameters that HAVE gradients, which starts anywhere inside the network's
    flat vector (first layer frozen, heads-only fine-tuning, an optimizer over a subset, a parameter whose grad is None).  The
    repack cache's epoch is keyed on the STORAGE of the vector (hip_backend._epoch_key), so such a step must still invalidate the
    fragment repack: the next forward equals a fresh module that loads the updated weights, and differs from the one before."""
    Pc, Pf = orc.scene_params(seed=4)
    kw = dict(D=8, W=256, input_ch=63, output_ch=5, skips=[4], input_ch_views=27, use_viewdirs=True)
    rays = orc.synthetic_rays(64, seed=12).to(dev)
    target = torch.rand(64, 3, generator=torch.Generator().manual_seed(2)).to(dev)
    for prec in ("fp16x3", "fp32"):
        nc = npa.NeRF(**kw).to(dev)
        nc.load_state_dict(Pc)
        if mode == "frozen_first_layer":
            for p_ in nc.pts_linears[0].parameters():
                p_.requires_grad_(False)
            trained = [p_ for p_ in nc.parameters() if p_.requires_grad]
        elif mode == "heads_only":
            trained = list(nc.rgb_linear.parameters()) + list(nc.alpha_linear.parameters())
        else:
            trained = list(nc.parameters())
        opt = npa.FlatAdam(trained, lr=1e-2)
        npa.set_precision(prec)
        try:
            render = lambda m: npa.render_rays(rays, m, None, 64, N_importance=0, white_bkgd=True)
            before = render(nc)
            npa.img2mse(before["rgb_map"], target).backward()
            if mode == "first_grad_none":
                nc.pts_linears[0].weight.grad = None
            w0 = nc.pts_linears[0].weight.detach().clone()
            opt.step()
            with torch.no_grad():
                after = render(nc)["rgb_map"]
                nc2 = npa.NeRF(**kw).to(dev)
                nc2.load_state_dict(nc.state_dict())
                fresh = render(nc2)["rgb_map"]
        finally:
            npa.set_precision("fp32")
        assert torch.equal(nc.pts_linears[0].weight.detach(), w0), "a parameter outside the optimizer's run moved"
        assert float((after - before["rgb_map"].detach()).abs().max()) > 1e-4, (mode, prec, "the step did not reach the kernels")
        assert torch.equal(after, fresh), (mode, prec)


def test_training_reaches_the_same_psnr_in_every_datapath(npa, dev):
    """End-to-end training equivalence: a student field is fitted to a teacher scene with the fused optimizer for 150
    steps in each datapath (same init, same batches); held-out PSNR rises from 12 dB to > 38 dB and the three
    datapaths end within 0.1 dB of each other (300 steps: 42.47 / 42.45 / 42.46 dB, tools/exp_converge.py)."""
    import math
    kw = dict(D=8, W=256, input_ch=63, output_ch=5, skips=[4], input_ch_views=27, use_viewdirs=True)
    Tc, Tf = orc.scene_params(seed=5)
    Sc, Sf = orc.scene_params(seed=6)

    def net(P):
        m = npa.NeRF(**kw).to(dev)
        m.load_state_dict(P)
        return m
    tc, tf = net(Tc), net(Tf)
    pool = orc.synthetic_rays(8192, seed=77).to(dev)
    held = orc.synthetic_rays(1024, seed=78).to(dev)
    rk = dict(N_samples=64, N_importance=128, white_bkgd=True, raw_noise_std=0.)
    with torch.no_grad():
        tgt_pool = torch.cat([npa.render_rays(pool[i:i + 4096], tc, None, network_fine=tf, perturb=0., **rk)["rgb_map"]
                              for i in range(0, 8192, 4096)])
        tgt_held = npa.render_rays(held, tc, None, network_fine=tf, perturb=0., **rk)["rgb_map"]

    def psnr(nc, nf):
        with torch.no_grad():
            out = npa.render_rays(held, nc, None, network_fine=nf, perturb=0., **rk)["rgb_map"]
        return -10 * math.log10(float(((out - tgt_held) ** 2).mean()))
    final = {}
    for prec in ("fp32", "fp16x3", "bf16x3"):
        nc, nf = net(Sc), net(Sf)
        opt = npa.FlatAdam(list(nc.parameters()) + list(nf.parameters()), lr=5e-4)
        start = psnr(nc, nf)
        g = torch.Generator().manual_seed(1)
        torch.manual_seed(0)
        npa.set_precision(prec)
        try:
            for step in range(150):
                idx = torch.randint(0, 8192, (1024,), generator=g).to(dev)
                opt.zero_grad()
                out = npa.render_rays(pool[idx], nc, None, network_fine=nf, perturb=1.0, **rk)
                loss = npa.img2mse(out["rgb_map"], tgt_pool[idx]) + npa.img2mse(out["rgb0"], tgt_pool[idx])
                loss.backward()
                opt.step()
        finally:
            npa.set_precision("fp32")
        final[prec] = psnr(nc, nf)
        assert start < 15.0 and final[prec] > 38.0, (prec, start, final)
    print("held-out PSNR after 150 steps:", {k: round(v, 3) for k, v in final.items()})
    assert max(final.values()) - min(final.values()) <= 0.1, final


def test_adversarial_scene_psnr_delta(npa, dev):
    """Unrelated coarse/fine networks with full-strength 2^9-frequency columns: per-ray agreement is not defined (the
    reference's own fp32-vs-fp64 runs disagree at 1e-2 here), the image-level criterion is.  Target = the oracle's
    image of the same networks perturbed by 0.3 % (a teacher scene; a random target would make the criterion vacuous)."""
    Pc, Pf = orc.scene_params_adversarial()
    rs = np.random.RandomState(5)
    Tc, Tf = ({k: v * torch.tensor(1.0 + 3e-3 * rs.standard_normal(tuple(v.shape)), dtype=torch.float32) for k, v in P.items()}
              for P in (Pc, Pf))
    kw = dict(D=8, W=256, input_ch=63, output_ch=5, skips=[4], input_ch_views=27, use_viewdirs=True)
    nc, nf = npa.NeRF(**kw).to(dev), npa.NeRF(**kw).to(dev)
    nc.load_state_dict(Pc); nf.load_state_dict(Pf)
    rays = orc.synthetic_rays(512, seed=13)
    ref = orc.trace_rays(rays, Pc, Pf, 64, 128, white_bkgd=True)
    target = orc.trace_rays(rays, Tc, Tf, 64, 128, white_bkgd=True)["rgb_map"]
    for prec, floor in (("fp32", 65.0), ("fp16x3", 63.0), ("bf16x3", 60.0)):        # measured 77.0 / - / 71.8 dB
        npa.set_precision(prec)
        try:
            with torch.no_grad():
                out = npa.render_rays(rays.to(dev), nc, None, 64, N_importance=128, network_fine=nf, white_bkgd=True)
        finally:
            npa.set_precision("fp32")
        g = orc.precision_gate(out["rgb_map"], ref["rgb_map"], target)
        print("adversarial", prec, g, "max|rgb0 - ref|", maxdiff(out["rgb0"], ref["rgb0"]))
        assert maxdiff(out["rgb0"], ref["rgb0"]) <= (1e-5 if prec == "fp32" else 1e-3)
        assert g["psnr_delta_db"] < 0.01 and g["psnr_vs_ref_db"] >= floor, g


# ---------------------------------------------------------------- workspace leases / recomputing backward
@pytest.mark.parametrize("precision", PARITY_DATAPATHS)
def test_large_chunks_backpropagate_in_subchunks(npa, dev, nets, precision, monkeypatch):
    """Ray chunks whose backward scratch exceeds the per-launch budget are rendered in equal sub-chunks.  While the saved
    activations of all sub-chunks fit SAVE_TOTAL_BYTES every sub-chunk keeps its own lease (no recomputation); beyond
    that the forward runs without saving and the backward re-runs it sub-chunk by sub-chunk.  Either way: outputs
    identical, gradients equal to the one-shot backward up to the fp32 summation order of the per-sub-chunk partial sums."""
    nc, nf, Pc, Pf = nets
    hb = npa.hip_backend
    n = 2500
    rays = orc.synthetic_rays(n, seed=77).to(dev)
    target = torch.rand(n, 3, generator=torch.Generator().manual_seed(3)).to(dev)
    rnd = {k: v.to(dev) for k, v in orc.synthetic_randoms(n, 64, 128, seed=5).items()}
    kw = dict(N_samples=64, N_importance=128, network_fine=nf, white_bkgd=True, perturb=1.0, raw_noise_std=0.5, retraw=True)

    def run():
        for m in (nc, nf):
            m.zero_grad()
        out = npa.render_rays(rays, nc, None, randoms=rnd, **kw)
        loss = npa.img2mse(out["rgb_map"], target) + npa.img2mse(out["rgb0"], target) + 1e-4 * out["raw"][..., 3].abs().mean()
        loss.backward()
        return {k: v.detach().clone() for k, v in out.items()}, nc.last_flat_grad.clone(), nf.last_flat_grad.clone()
    npa.set_precision(precision)
    try:
        assert hb.max_saved_rays(64, 128, precision) >= n
        out_a, gc_a, gf_a = run()
        monkeypatch.setattr(hb, "SAVE_BUDGET_BYTES", 4 * hb.workspace_floats(1024, 64, 128, True, precision) + 1)      # -> 1024-ray sub-chunks
        assert hb.max_saved_rays(64, 128, precision) == 1024       # (sized for THIS datapath's layouts: 16-bit tiles on the split ones)
        # (a) the sub-chunks' saved activations fit SAVE_TOTAL_BYTES: each keeps its lease, the forward runs ONCE per sub-chunk
        calls = []
        real_fwd = hb.field_fwd
        monkeypatch.setattr(hb, "field_fwd", lambda *a, **k: (calls.append(k.get("save_act")), real_fwd(*a, **k))[1])
        out_t, gc_t, gf_t = run()
        assert calls == [True] * 6, calls           # 3 sub-chunks (2500 rays -> 3 x 896) x (coarse, fine), all saving, none re-run
        # (a') the free-memory estimate was too optimistic: the device runs out of memory while the third sub-chunk leases its buffers
        #  -> everything is handed back and the recompute plan takes over, same results
        import sys
        render_mod = sys.modules["nerf_pytorch_amd.render"]
        real_take, takes = hb.WORKSPACE.take, []

        def failing_take(n_floats, device):
            takes.append(n_floats)
            if len(takes) == 5:
                raise torch.cuda.OutOfMemoryError("simulated: the lease of the third sub-chunk does not fit")
            return real_take(n_floats, device)
        monkeypatch.setattr(hb.WORKSPACE, "take", failing_take)
        del calls[:]
        out_o, gc_o, gf_o = run()
        monkeypatch.setattr(hb.WORKSPACE, "take", real_take)
        assert render_mod.LAST_BACKWARD_PLAN[0] == "recompute" and calls[:5] == [True] * 5 and calls[5:7] == [False, False] and calls[7:] == [True] * 6, calls
        # (b) they do not fit: forward without saving, backward re-runs it sub-chunk by sub-chunk
        monkeypatch.setattr(hb, "SAVE_TOTAL_BYTES", 0)
        del calls[:]
        out_b, gc_b, gf_b = run()
        assert calls == [False, False] + [True] * 6, calls
        assert torch.equal(gc_o, gc_b) and torch.equal(gf_o, gf_b)
    finally:
        npa.set_precision("fp32")
    for out_x, gc_x, gf_x in ((out_t, gc_t, gf_t), (out_b, gc_b, gf_b)):
        for k in out_a:
            assert torch.equal(torch.isnan(out_a[k]), torch.isnan(out_x[k])) and torch.equal(torch.nan_to_num(out_a[k]), torch.nan_to_num(out_x[k])), k
        for ga, gb in ((gc_a, gc_x), (gf_a, gf_x)):
            assert float((ga - gb).abs().max()) <= 2e-5 * float(ga.abs().max()), float((ga - gb).abs().max()) / float(ga.abs().max())
    assert torch.equal(gc_t, gc_b) and torch.equal(gf_t, gf_b)      # same sub-chunks, same kernels: the two modes agree bit for bit


@pytest.mark.parametrize("datapath", BOUNDARY_DATAPATHS, indirect=True)
def test_training_steps_reuse_the_same_workspace_buffers(npa, dev, nets, datapath):
    """No per-step allocation of the backward scratch: consecutive steps lease the same device buffers (hb.WORKSPACE)."""
    nc, nf, Pc, Pf = nets
    hb = npa.hip_backend
    rays = orc.synthetic_rays(256, seed=9).to(dev)
    target = torch.rand(256, 3, device=dev)
    seen = []
    real_take = hb.WORKSPACE.take

    def spy(n_floats, device):
        t = real_take(n_floats, device)
        seen[-1].append(t.data_ptr())
        return t
    hb.WORKSPACE.take = spy
    try:
        for step in range(3):
            seen.append([])
            for m in (nc, nf):
                m.zero_grad()
            out = npa.render_rays(rays, nc, None, 64, N_importance=128, network_fine=nf, white_bkgd=True, perturb=1.0)
            (npa.img2mse(out["rgb_map"], target) + npa.img2mse(out["rgb0"], target)).backward()
    finally:
        hb.WORKSPACE.take = real_take
    assert len(seen[1]) == 6 and seen[1] == seen[2], seen        # 2 act + 2 x (delta, partial) leases per step, same pointers


# ---------------------------------------------------------------- boundary: render() / run_network / NeRF.forward
@pytest.mark.parametrize("datapath", BOUNDARY_DATAPATHS, indirect=True)
def test_render_boundary_and_chunking(npa, dev, nets, datapath):
    tol = BOUNDARY_TOL[datapath]
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
    stable = ~orc.endpoint_unstable(ref["_weights0"])
    assert maxdiff(extras["rgb0"].reshape(-1, 3), ref["rgb0"]) <= tol["coarse"]
    ref64 = orc.trace_rays(flat.double(), {k: v.double() for k, v in Pc.items()}, {k: v.double() for k, v in Pf.items()},
                           64, 128, perturb=0., white_bkgd=True)
    floor = maxdiff(ref["rgb_map"][stable], ref64["rgb_map"][stable])
    if datapath == "fp32":
        assert maxdiff(rgb.reshape(-1, 3)[stable], ref["rgb_map"][stable]) <= max(tol["fine"], 10 * floor)
    else:   # the split datapaths' fine pass: p95 + worst ray, as the goldens hold it (GOLD_TOL: sample_pdf amplifies any coarse rounding)
        err = (rgb.reshape(-1, 3).cpu() - ref["rgb_map"]).abs().max(-1)[0][stable]
        assert float(err.quantile(0.95)) <= max(tol["fine"], 10 * floor) and float(err.max()) <= 5e-3, (float(err.quantile(0.95)), float(err.max()))
    # run_network / NeRF.forward on explicit points
    pts = torch.randn(7, 5, 3)
    vd = torch.nn.functional.normalize(torch.randn(7, 3), dim=-1)
    with torch.no_grad():
        got = npa.run_network(pts.to(dev), vd.to(dev), nf, None, None)
        emb = torch.cat([orc.posenc(pts.reshape(-1, 3), 10), orc.posenc(vd[:, None].expand(7, 5, 3).reshape(-1, 3), 4)], -1)
        got2 = nf(emb.to(dev))
    ref = orc.query_field(Pf, pts, vd)
    assert maxdiff(got, ref) <= tol["field"] * max(1.0, float(ref.abs().max()) / 10)
    assert maxdiff(got2.reshape(7, 5, 4), ref) <= tol["field"] * max(1.0, float(ref.abs().max()) / 10)


@pytest.mark.parametrize("ndc", [False, True])
@pytest.mark.parametrize("static", [False, True])
def test_make_rays_matches_reference_ray_setup(npa, dev, ndc, static):
    """One launch = get_rays + view directions + ndc_rays + near / far of render(c2w=...) (run_nerf.py:95-123).
    Same operation order as the reference; stated 2e-6 relative (torch's 3-term sums / norm may associate differently)."""
    H, W, focal = 19, 23, 31.0
    K = np.array([[focal, 0, 0.5 * W], [0, focal * 1.1, 0.5 * H], [0, 0, 1]], dtype=np.float32)
    c2w = torch.tensor([[0.96, 0.0, 0.28, 0.1], [0.0, 1.0, 0.0, -0.2], [-0.28, 0.0, 0.96, 0.4]])
    c2w_s = torch.tensor([[1.0, 0.0, 0.0, 0.0], [0.0, 0.8, -0.6, 0.1], [0.0, 0.6, 0.8, 0.3]]) if static else None
    near, far = (0., 1.) if ndc else (2., 6.)
    got = npa.hip_backend.make_rays(H, W, K, c2w, c2w_s, ndc, near, far, dev).cpu()
    ro, rd = orc.pinhole_rays(H, W, K, c2w)
    view = rd / torch.norm(rd, dim=-1, keepdim=True)
    if static:
        ro, rd = orc.pinhole_rays(H, W, K, c2w_s)
    if ndc:
        ro, rd = orc.ndc_warp(H, W, K[0][0], 1., ro, rd)
    want = torch.cat([ro.reshape(-1, 3), rd.reshape(-1, 3), torch.full((H * W, 1), near), torch.full((H * W, 1), far),
                      view.reshape(-1, 3)], -1).float()
    assert got.shape == (H * W, 11)
    scale = want.abs().amax(0).clamp_min(1.0)
    assert float(((got - want).abs() / scale).max()) <= 2e-6
    assert torch.equal(got[:, 6:8], want[:, 6:8])


@pytest.mark.parametrize("ndc", [False, True])
def test_assemble_rays_matches_reference_ray_assembly(npa, dev, ndc):
    """nerf_assemble_rays = the rays=... branch of render() (run_nerf.py:95-123): view directions, ndc_rays, near / far."""
    cfg = orc.FERN if ndc else orc.LEGO
    batch = orc.fern_batch(777, seed=9) if ndc else orc.lego_batch(777, seed=9)
    K = orc.intrinsics(cfg)
    got = npa.hip_backend.assemble_rays(batch[0].to(dev), batch[1].to(dev), ndc, cfg["H"], cfg["W"], K[0][0], cfg["near"], cfg["far"]).cpu()
    want = orc.assemble_render_rays(cfg["H"], cfg["W"], K, batch[0], batch[1], ndc, cfg["near"], cfg["far"])
    assert got.shape == (777, 11)
    scale = want.abs().amax(0).clamp_min(1.0)
    assert float(((got - want).abs() / scale).max()) <= 2e-6
    assert torch.equal(got[:, 6:8], want[:, 6:8])
    # render() takes this path for rays=(rays_o, rays_d) on the GPU and the torch formulation otherwise: same records
    import sys
    R = sys.modules[npa.render.__module__]
    o, d = batch[0].to(dev), batch[1].to(dev)
    vd = d / torch.norm(d, dim=-1, keepdim=True)
    if ndc:
        o, d = R.ndc_rays(cfg["H"], cfg["W"], K[0][0], 1., o, d)
    ref = torch.cat([o, d, cfg["near"] * torch.ones_like(d[..., :1]), cfg["far"] * torch.ones_like(d[..., :1]), vd], -1).cpu()
    assert float(((got - ref).abs() / scale).max()) <= 2e-6


@pytest.mark.parametrize("datapath", BOUNDARY_DATAPATHS, indirect=True)
def test_render_path_overlapped_output(npa, dev, nets, tmp_path, datapath):
    """render_path (run_nerf.py:137-175): frames rendered from poses, device-side to8b, asynchronous copies and PNG
    encoding on worker threads -- arrays and files equal to the synchronous formulation."""
    nc, nf, Pc, Pf = nets
    H, W, focal = 10, 12, 15.0
    K = np.array([[focal, 0, 0.5 * W], [0, focal, 0.5 * H], [0, 0, 1]])
    poses = torch.stack([torch.tensor([[1.0, 0, 0, 0.1 * i], [0, 0.8, -0.6, 0.2], [0, 0.6, 0.8, 4.0], [0, 0, 0, 1.0]]) for i in range(3)]).to(dev)
    kw = dict(network_fn=nc, network_query_fn=None, N_samples=64, N_importance=128, network_fine=nf, perturb=0.,
              white_bkgd=True, raw_noise_std=0., ndc=False, near=2., far=6., use_viewdirs=True)
    with torch.no_grad():
        rgbs, disps = npa.render_path(poses, (H, W, focal), K, 1 << 15, kw, savedir=str(tmp_path))
        one = npa.render(H, W, K, chunk=1 << 15, c2w=poses[1][:3, :4], **kw)
    assert rgbs.shape == (3, H, W, 3) and disps.shape == (3, H, W) and rgbs.dtype == np.float32
    assert np.array_equal(rgbs[1], one[0].cpu().numpy()) and np.array_equal(disps[1], one[1].cpu().numpy(), equal_nan=True)
    import zlib, struct
    for i in range(3):
        data = open(tmp_path / f"{i:03d}.png", "rb").read()
        assert data[:8] == b"\x89PNG\r\n\x1a\n"
        pos, idat = 8, b""
        while pos < len(data):
            n, tag = struct.unpack(">I", data[pos:pos + 4])[0], data[pos + 4:pos + 8]
            if tag == b"IDAT":
                idat += data[pos + 8:pos + 8 + n]
            pos += 12 + n
        rows = np.frombuffer(zlib.decompress(idat), dtype=np.uint8).reshape(H, 1 + 3 * W)
        assert np.array_equal(rows[:, 1:].reshape(H, W, 3), npa.to8b(rgbs[i]))


@pytest.mark.parametrize("datapath", BOUNDARY_DATAPATHS, indirect=True)
def test_training_step_moves_parameters_like_the_oracle(npa, dev, datapath):
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
        assert abs(loss.item() - lo.item()) <= BOUNDARY_TOL[datapath]["loss"], (step, loss.item(), lo.item())
    assert nc._is_bound() and nf._is_bound()


# ---------------------------------------------------------------- network_query_fn as a live hook (run_nerf.py:385, :401)
@pytest.mark.gpu
@pytest.mark.parametrize("datapath", BOUNDARY_DATAPATHS, indirect=True)
def test_user_network_query_fn_is_called_for_every_pass(npa, dev, nets, datapath):
    """A user's own network_query_fn in render_kwargs is CALLED (with the reference's arguments) for the coarse and the fine pass and
    what it returns is what gets composited; None / the function create_nerf builds select the fused kernels.  The stock body behind
    a user's wrapper reproduces the fused path (same kernels per stage, same arithmetic), values and gradients."""
    nc, nf, Pc, Pf = nets
    n = 48
    rays = orc.synthetic_rays(n, seed=12).to(dev)
    target = torch.rand(n, 3, device=dev)
    rnd = {k: v.to(dev) for k, v in orc.synthetic_randoms(n, 64, 128, seed=5).items()}
    kw = dict(N_samples=64, N_importance=128, network_fine=nf, white_bkgd=True, perturb=1.0, raw_noise_std=1.0, retraw=True, randoms=rnd)
    calls = []

    def hook(pts, viewdirs, net):
        calls.append((tuple(pts.shape), tuple(viewdirs.shape), net))
        return npa.run_network(pts, viewdirs, net, None, None)

    def grads(query_fn):
        for m in (nc, nf):
            m.zero_grad()
        out = npa.render_rays(rays, nc, query_fn, **kw)
        (npa.img2mse(out["rgb_map"], target) + npa.img2mse(out["rgb0"], target)).backward()
        return out, [p.grad.clone() for m in (nc, nf) for p in m.parameters()]
    fused, g_fused = grads(None)
    assert calls == []
    hooked, g_hook = grads(hook)
    assert calls == [((n, 64, 3), (n, 3), nc), ((n, 192, 3), (n, 3), nf)]
    assert set(hooked) == set(fused)
    for k in fused:        # the points o + d z are formed by torch here and inside the kernel there: the same fp32 operations
        assert maxdiff(hooked[k], fused[k]) <= 1e-6 * max(1.0, float(fused[k].detach().abs().nan_to_num().max())), k
    for a, b in zip(g_hook, g_fused):
        assert maxdiff(a, b) <= 1e-4 * float(b.abs().max()) + 1e-12
    # ... and it is live: a hook that changes raw changes the image (here: a density floor on raw[..., 3])
    denser = npa.render_rays(rays, nc, lambda p, v, m: torch.cat([hook(p, v, m)[..., :3], hook(p, v, m)[..., 3:] + 5.0], -1), **kw)
    assert float((denser["acc_map"] - fused["acc_map"]).abs().max()) > 1e-3
    # create_nerf's own function is recognised as the stock one
    import types
    args = npa.config_parser().parse_args([])
    tr = npa.create_nerf(types.SimpleNamespace(**dict(vars(args), basedir=None, expname=None)), device=dev)[0]
    calls.clear()
    with torch.no_grad():
        stock = npa.render_rays(rays, tr["network_fn"], tr["network_query_fn"], N_samples=64, N_importance=128, network_fine=tr["network_fine"])
    assert calls == [] and stock["rgb_map"].shape == (n, 3)
    for m in (nc, nf):
        m.zero_grad()
