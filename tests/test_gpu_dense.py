"""Architectures outside the fused kernels (-m gpu): NeRF(...) with any arguments the reference's constructor accepts -- other
depths / widths / frequency counts, the identity embedding of --i_embed -1, use_viewdirs=False with its output_linear head --
evaluated layer by layer (nerf_pytorch_amd/dense.py over csrc/dense.hip).  Checked against the oracle's field_mlp_arch /
trace_rays(arch=...), which oracle/pin_against_reference.py pins bit-identical to the real reference for these same
architectures (outputs and every gradient)."""
import math

import numpy as np
import pytest
import torch

import nerf_oracle as orc
from test_gpu_parity import dev, maxdiff, npa  # noqa: F401  (fixtures)

pytestmark = pytest.mark.gpu

ARCHS = {
    "no_viewdirs": orc.arch_of(use_viewdirs=False),                                         # --use_viewdirs absent: output_linear, 5 channels
    "narrow_shallow": orc.arch_of(D=6, W=128, multires=6, multires_views=2),                # --netdepth 6 --netwidth 128 --multires 6 ...
    "identity_embedding": orc.arch_of(D=4, W=64, multires=-1, multires_views=-1, output_ch=4),   # --i_embed -1
    "no_viewdirs_small": orc.arch_of(D=7, W=96, multires=3, use_viewdirs=False, output_ch=4),
}
CTOR = ("D", "W", "input_ch", "input_ch_views", "output_ch", "skips", "use_viewdirs")


def _net(npa, dev, arch, seed):
    P = orc.make_arch_params(arch, seed)
    net = npa.NeRF(**{k: arch[k] for k in CTOR}).to(dev)
    assert type(net).__name__ == "DenseNeRF"
    assert [(k, tuple(v.shape)) for k, v in net.state_dict().items()] == orc.arch_param_shapes(arch)
    net.load_state_dict(P)
    return net, P


@pytest.mark.parametrize("name", list(ARCHS))
def test_dense_layer_stack_against_the_oracle(npa, dev, name):
    """NeRF.forward on embedded inputs (run_nerf_helpers.py:96-119): outputs and every parameter gradient vs fp64."""
    arch = ARCHS[name]
    net, P = _net(npa, dev, arch, 31)
    M = 777
    g = torch.Generator().manual_seed(5)
    x3 = torch.randn(M, 3, generator=g) * 2.0
    feats = orc.posenc_or_identity(x3, arch["multires"])
    if arch["use_viewdirs"]:
        d3 = torch.nn.functional.normalize(torch.randn(M, 3, generator=g), dim=-1)
        feats = torch.cat([feats, orc.posenc_or_identity(d3, arch["multires_views"])], -1)
    up = torch.randn(M, 4 if arch["use_viewdirs"] else arch["output_ch"], generator=g)
    P64 = {k: v.double().requires_grad_(True) for k, v in P.items()}
    ref = orc.field_mlp_arch(P64, feats.double(), arch)
    (ref * up.double()).sum().backward()
    out = net(feats.to(dev))
    assert out.shape == ref.shape
    (out * up.to(dev)).sum().backward()
    scale = max(1.0, float(ref.detach().abs().max()))
    assert maxdiff(out, ref) <= 2e-5 * scale, (name, maxdiff(out, ref), scale)
    for k, p in net.named_parameters():
        r = P64[k].grad
        if r is None:
            assert p.grad is None, k        # views_linears without view directions: unused, as in the reference
            continue
        got = p.grad.cpu().double()
        rel = float((got - r).norm() / r.norm().clamp_min(1e-300))
        assert rel <= 5e-3, (name, k, rel)      # a ReLU unit within fp32 rounding of its kink may flip (cf. test_field_backward_no_exclusion)
        cos = float((got * r).sum() / (got.norm() * r.norm()).clamp_min(1e-300))
        assert cos >= 1.0 - 1e-5, (name, k, cos)


@pytest.mark.parametrize("name", list(ARCHS))
def test_dense_render_rays_against_the_oracle(npa, dev, name):
    """render_rays end to end (coarse + fine, jitter, density noise, white background) for the other architectures: the
    reference's dict incl. extras, loss and gradients of both networks vs the pinned oracle in fp64."""
    arch = ARCHS[name]
    net_c, Pc = _net(npa, dev, arch, 31)
    net_f, Pf = _net(npa, dev, arch, 32)
    n, n_c, n_f = 96, 24, 40
    rays = orc.synthetic_rays(n, seed=3)
    rr = rays if arch["use_viewdirs"] else rays[:, :8].contiguous()
    rnd = orc.synthetic_randoms(n, n_c, n_f, seed=21)
    target = torch.rand(n, 3, generator=torch.Generator().manual_seed(1))
    P64 = [{k: v.double().requires_grad_(True) for k, v in P.items()} for P in (Pc, Pf)]
    kw = dict(perturb=1.0, white_bkgd=True, raw_noise_std=0.5, retraw=True, arch=arch)
    o64 = orc.trace_rays(rr.double(), P64[0], P64[1], n_c, n_f, **kw, **{k: v.double() for k, v in rnd.items()})
    with torch.no_grad():
        o32 = orc.trace_rays(rr, Pc, Pf, n_c, n_f, **kw, **rnd)
    loss64 = ((o64["rgb_map"] - target.double()) ** 2).mean() + ((o64["rgb0"] - target.double()) ** 2).mean()
    loss64.backward()
    out = npa.render_rays(rr.to(dev), net_c, None, N_samples=n_c, N_importance=n_f, network_fine=net_f, perturb=1.0, white_bkgd=True,
                          raw_noise_std=0.5, retraw=True, randoms={k: v.to(dev) for k, v in rnd.items()})
    assert set(out) == {"rgb_map", "disp_map", "acc_map", "raw", "rgb0", "disp0", "acc0", "z_std"}
    assert out["raw"].shape == (n, n_c + n_f, 4 if arch["use_viewdirs"] else arch["output_ch"])
    for k in ("rgb0", "acc0"):
        noise = (o32[k].double() - o64[k].detach()).abs()
        err = (out[k].detach().cpu().double() - o64[k].detach()).abs()
        assert float((err - 10 * noise).max()) <= 1e-5, (name, k, float(err.max()), float(noise.max()))
    for k in ("rgb_map", "acc_map", "z_std"):
        noise = (o32[k].double() - o64[k].detach()).abs()
        err = (out[k].detach().cpu().double() - o64[k].detach()).abs()
        frac = float((err <= torch.clamp(10 * noise, min=1e-5)).double().mean())
        assert frac >= 0.9, (name, k, frac, float(err.max()))
    loss = npa.img2mse(out["rgb_map"], target.to(dev)) + npa.img2mse(out["rgb0"], target.to(dev))
    loss.backward()
    assert abs(float(loss.detach()) - float(loss64.detach())) <= 1e-4 * max(1.0, float(loss64.detach())), (name, float(loss.detach()), float(loss64.detach()))
    for net, P in ((net_c, P64[0]), (net_f, P64[1])):
        got = torch.cat([p.grad.reshape(-1) for _, p in net.named_parameters() if p.grad is not None]).cpu().double()
        ref = torch.cat([P[k].grad.reshape(-1) for k, p in net.named_parameters() if p.grad is not None])
        cos = float((got * ref).sum() / (got.norm() * ref.norm()).clamp_min(1e-300))
        assert cos >= 1.0 - 1e-4, (name, cos)


def test_dense_path_through_create_nerf_render_and_adam(npa, dev):
    """The reference's command line without --use_viewdirs, with --netdepth 4 --netwidth 64 --i_embed -1: create_nerf builds the
    networks, render() runs a frame chunk (c2w branch, ray records built on the device), two optimizer steps lower the loss."""
    parser = npa.config_parser()
    args = parser.parse_args(["--netdepth", "4", "--netwidth", "64", "--netdepth_fine", "6", "--netwidth_fine", "96", "--i_embed", "-1",
                              "--N_samples", "16", "--N_importance", "24", "--white_bkgd", "--dataset_type", "blender", "--no_reload",
                              "--expname", "dense_test", "--basedir", "/tmp/nerf_dense_test", "--lrate", "2e-3"])
    assert not args.use_viewdirs
    torch.manual_seed(0)
    kw_train, kw_test, start, grad_vars, opt = npa.create_nerf(args, device=dev)
    assert type(kw_train["network_fn"]).__name__ == "DenseNeRF" and kw_train["network_fn"].W == 64 and kw_train["network_fine"].D == 6
    H, W, focal = 12, 16, 20.0
    K = np.array([[focal, 0, 0.5 * W], [0, focal, 0.5 * H], [0, 0, 1]])
    c2w = torch.tensor([[1.0, 0, 0, 0.0], [0, 1.0, 0, 0.0], [0, 0, 1.0, 4.0]], device=dev)
    with torch.no_grad():
        rgb, disp, acc, extras = npa.render(H, W, K, chunk=64, c2w=c2w, near=2.0, far=6.0, retraw=True, **kw_test)
    assert rgb.shape == (H, W, 3) and extras["raw"].shape == (H, W, 40, 5) and not torch.isnan(rgb).any()
    rays_o, rays_d = npa.get_rays(H, W, K, c2w.cpu())
    batch = torch.stack([rays_o.reshape(-1, 3), rays_d.reshape(-1, 3)], 0).to(dev)
    target = torch.rand(H * W, 3, device=dev)
    losses = []
    for _ in range(8):          # deterministic sampling (test-time kwargs), so the loss sequence is a function of the parameters only
        opt.zero_grad()
        rgb, disp, acc, extras = npa.render(H, W, K, chunk=100, rays=batch, near=2.0, far=6.0, **kw_test)
        loss = npa.img2mse(rgb, target) + npa.img2mse(extras["rgb0"], target)
        loss.backward()
        assert all(p.grad is not None and bool(torch.isfinite(p.grad).all()) for n_, p in kw_train["network_fn"].named_parameters()
                   if not n_.startswith("views_linears"))
        opt.step()
        losses.append(float(loss.detach()))
    assert losses[-1] < losses[0], losses


def test_networks_of_the_two_kinds_do_not_mix(npa, dev):
    arch = ARCHS["narrow_shallow"]
    dense, _ = _net(npa, dev, arch, 1)
    fused = npa.NeRF(D=8, W=256, input_ch=63, output_ch=5, skips=[4], input_ch_views=27, use_viewdirs=True).to(dev)
    rays = orc.synthetic_rays(8, seed=1).to(dev)
    with pytest.raises(NotImplementedError):
        npa.render_rays(rays, fused, None, N_samples=8, N_importance=8, network_fine=dense)
    with pytest.raises(ValueError):
        npa.NeRF(D=5, W=64, input_ch=63, input_ch_views=27, skips=[4], use_viewdirs=True)       # the reference fails at its first forward


@pytest.mark.parametrize("name", sorted(orc.DENSE_CASES))
def test_dense_architectures_against_reference_produced_fixtures(npa, dev, name):
    """tests/golden/dense_*.npz hold what the REAL reference computed through render() for these architectures (128 rays, 24 + 40
    samples, jitter, density noise, white background) together with its own fp32-vs-fp64 distance per ray.  render() here, same
    rays / weights / draws: coarse quantities per ray within 10 x that distance (floor 1e-5), fine quantities on 90 % of the rays
    (sample_pdf amplifies rounding in empty bins, in the reference itself), loss, and every gradient tensor."""
    import os
    gold = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", name + ".npz"))
    arch, Pc, Pf, batch, target, n_c, n_f = orc.dense_case(name)
    n = batch.shape[1]
    nets = []
    for P in (Pc, Pf):
        net = npa.NeRF(**{k: arch[k] for k in CTOR}).to(dev)
        net.load_state_dict(P)
        nets.append(net)
    torch.manual_seed(55)       # the stream the reference consumed
    rnd = dict(t_rand=torch.rand(n, n_c), noise_c=torch.randn(n, n_c), u=torch.rand(n, n_f), noise_f=torch.randn(n, n_c + n_f))
    cfg = orc.LEGO
    rgb, disp, acc, extras = npa.render(cfg["H"], cfg["W"], orc.intrinsics(cfg), chunk=1024, rays=batch.to(dev), ndc=False, near=2.0, far=6.0,
                                        use_viewdirs=arch["use_viewdirs"], network_fn=nets[0], network_query_fn=None, N_samples=n_c,
                                        N_importance=n_f, network_fine=nets[1], perturb=1.0, white_bkgd=True, raw_noise_std=0.5, retraw=True,
                                        randoms={k: v.to(dev) for k, v in rnd.items()})
    out = dict(extras, rgb_map=rgb, disp_map=disp, acc_map=acc)
    assert out["raw"].shape == gold["raw"].shape
    for k in ("rgb0", "acc0"):
        err = np.abs(out[k].detach().cpu().numpy().astype(np.float64) - gold[k])
        assert float((err - 10 * gold[k + "/noise"]).max()) <= 1e-5, (name, k, float(err.max()))
    for k in ("rgb_map", "acc_map", "z_std"):
        err = np.abs(out[k].detach().cpu().numpy().astype(np.float64) - gold[k])
        frac = float((err <= np.maximum(10 * gold[k + "/noise"], 1e-5)).mean())
        assert frac >= 0.9, (name, k, frac, float(err.max()))
    loss = npa.img2mse(rgb, target.to(dev)) + npa.img2mse(extras["rgb0"], target.to(dev))
    loss.backward()
    assert abs(float(loss.detach()) - float(gold["loss"])) <= 1e-4, (name, float(loss.detach()), float(gold["loss"]))
    for tag, net in (("c", nets[0]), ("f", nets[1])):
        for nm, p in net.named_parameters():
            key = f"grad_{tag}/{nm}/val"
            if key not in gold.files:
                assert p.grad is None, (tag, nm)
                continue
            got = p.grad.reshape(-1)[torch.tensor(gold[f"grad_{tag}/{nm}/idx"], device=dev)].cpu().numpy().astype(np.float64)
            tol = max(2e-4 * float(gold[f"grad_{tag}/{nm}/max"]), 10 * float(gold[f"grad_{tag}/{nm}/noise"]))
            assert float(np.abs(got - gold[key]).max()) <= tol, (name, tag, nm, float(np.abs(got - gold[key]).max()), tol)
