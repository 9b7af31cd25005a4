"""GPU tests (-m gpu) of "fp16x3w" (round 6): the fp16 three-term split whose weight-gradient GEMM contracts TWO-WORD operands.
The forward (SAVE = 3) and the delta chain (TWO) store the lo words T(v - hi) next to the hi words, wgrad1_kernel<SplitF16, 3>
evaluates d_hi^T X_hi + d_hi^T X_lo + d_lo^T X_hi.  What must hold: the forward's values and hi words are fp16x3's bit for bit; hi + lo
represents the fp32 activation / delta to ~2^-22; the gradient error against fp64 autograd (the kernel's own ReLU pattern forced) drops
from the operand rounding's 2^-12 class to the product class of the forward."""
import numpy as np
import pytest
import torch

import nerf_oracle as orc
from test_gpu_parity import npa, dev, nets, maxdiff, _flat_grads_through_render      # noqa: F401  (fixtures)
from test_gpu_round3 import _decode_masks, _field_with_forced_relu

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("n_rays,S", [(64, 64), (37, 192), (5, 3), (1, 1)])
def test_two_word_forward_saves_hi_and_lo(npa, dev, nets, n_rays, S):
    nc, nf, Pc, Pf = nets
    hb = npa.hip_backend
    rays = orc.synthetic_rays(n_rays, seed=S)
    z = torch.sort(torch.rand(n_rays, S, generator=torch.Generator().manual_seed(S)) * 4.0 + 2.0, -1)[0]
    packed = nf.packed_params("fp16x3w")
    assert packed is nf.packed_params("fp16x3")                 # the same fragment repack
    raw1, act1 = hb.field_fwd(packed, rays.to(dev), z.to(dev), save_act=True, precision="fp16x3")
    raw2, act2 = hb.field_fwd(packed, rays.to(dev), z.to(dev), save_act=True, precision="fp16x3w")
    assert torch.equal(raw1, raw2)
    assert hb.buffer_layout(act1)[0] == 5 and hb.buffer_layout(act2)[0] == 6
    assert act2.numel() >= hb.act_floats(n_rays, S, "fp16x3w") == 2 * hb.act_floats(n_rays, S, "fp16x3")
    assert torch.equal(hb.saved_masks(act1, n_rays, S, "fp16x3"), hb.saved_masks(act2, n_rays, S, "fp16x3w"))
    assert torch.equal(hb.saved_dir(act1, n_rays, S, "fp16x3")[:, :27], hb.saved_dir(act2, n_rays, S, "fp16x3w")[:, :27])    # (columns 27..31: unused)
    pts = (rays[:, None, 0:3] + rays[:, None, 3:6] * z[..., None])
    P64 = {k: v.double() for k, v in Pf.items()}
    feats = torch.cat([orc.posenc(pts.reshape(-1, 3).double(), 10), orc.posenc(rays[:, None, 8:11].expand(n_rays, S, 3).reshape(-1, 3).double(), 4)], -1)
    _, hidden, _, hv = orc.field_mlp(P64, feats, return_hidden=True)
    refs = {**{f"h{l}": hidden[l] for l in range(8)}, "hv": hv, "enc": feats[:, :63]}
    for region, ref in refs.items():
        hi1 = hb.saved_rows(act1, n_rays, S, region, precision="fp16x3")
        hi = hb.saved_rows(act2, n_rays, S, region, precision="fp16x3w")
        lo = hb.saved_rows(act2, n_rays, S, region, precision="fp16x3w", part="lo")
        if region == "enc":
            hi1, hi, lo = hi1[:, :63], hi[:, :63], lo[:, :63]
        assert torch.equal(hi1, hi), region                     # the hi words are fp16x3's
        scale = max(1.0, float(ref.abs().max()))
        err_hi = maxdiff(hi.cpu().double(), ref)
        err_two = maxdiff(hi.cpu().double() + lo.cpu().double(), ref)
        # lo = fp16(v - hi): at most half an ulp of hi, and hi + lo is the fp32 value to ~2^-22 (the forward's own distance to fp64
        # is of that size: 2e-5 of the layer's scale is test_field_forward_fp16x3's bound for raw)
        assert float((lo.abs() - 2.0 ** -11 * hi.abs().clamp_min(2.0 ** -14)).max()) <= 0, region
        assert err_two <= 2e-5 * scale, (region, err_two, scale)
        if err_hi > 1e-4 * scale:
            assert err_two <= err_hi / 20, (region, err_hi, err_two)
    hb.WORKSPACE.give(act1)
    hb.WORKSPACE.give(act2)


def _grad_vs_fp64(npa, dev, nets, precision, n_rays, S, d_raw_of, chunk=64):
    """(relative L2 of the whole gradient, worst tensor max|err| / max|g|) of hb.field_bwd on `precision` against fp64 autograd of the
    reference network with the kernel's own ReLU pattern forced; d_raw_of(raw, rays, z) builds the upstream gradient on the device"""
    nc, nf, Pc, Pf = nets
    hb = npa.hip_backend
    P = n_rays * S
    g = torch.Generator().manual_seed(11 + n_rays)
    rays = orc.synthetic_rays(n_rays, seed=29 + S)
    z = torch.sort(torch.rand(n_rays, S, generator=g) * 4.0 + 2.0, -1)[0]
    packed = nf.packed_params(precision)
    rd, zd = rays.to(dev), z.to(dev)
    raw, act = hb.field_fwd(packed, rd, zd, save_act=True, precision=precision)
    d_raw = d_raw_of(raw, rd, zd, g)
    masks = _decode_masks(npa, act, P, n_rays)
    grad = torch.full((595844,), float("nan"), device=dev)
    hb.field_bwd(packed, act, d_raw, grad, accumulate=False, precision=precision, params=nf.flat_params())
    hb.WORKSPACE.give(act)
    grad = grad.cpu().double()
    assert not torch.isnan(grad).any()
    P64 = {k: v.double().requires_grad_(True) for k, v in Pf.items()}
    d64 = d_raw.cpu().double().reshape(-1, 4)
    for lo in range(0, n_rays, chunk):
        r = rays[lo:lo + chunk]
        pts = (r[:, None, 0:3] + r[:, None, 3:6] * z[lo:lo + chunk, :, None]).reshape(-1, 3).double()
        dirs = r[:, None, 8:11].expand(r.shape[0], S, 3).reshape(-1, 3).double()
        feats = torch.cat([orc.posenc(pts, 10), orc.posenc(dirs, 4)], -1)
        out, _ = _field_with_forced_relu(P64, feats, [m[lo * S:(lo + chunk) * S] for m in masks])
        (out * d64[lo * S:(lo + chunk) * S]).sum().backward()
    ref = torch.cat([P64[nm].grad.reshape(-1) for nm, _, _ in hb.param_table()])
    rel = float((grad - ref).norm() / ref.norm())
    worst = max(maxdiff(grad[off:off + int(np.prod(shape))], P64[nm].grad.reshape(-1)) / float(P64[nm].grad.abs().max()) for nm, off, shape in hb.param_table())
    return rel, worst


def _training_upstream(npa, target_seed=5):
    hb = npa.hip_backend

    def make(raw, rd, zd, g):
        n = raw.shape[0]
        target = torch.rand(n, 3, generator=torch.Generator().manual_seed(target_seed)).to(raw.device)
        rgb, _, _, _, _ = hb.raw2outputs(raw, zd, rd, 11, None, 0.0, True, rays_d_offset=3)
        d_rgb = (2.0 / (3 * n)) * (rgb - target)
        return hb.raw2outputs_bwd(raw, zd, rd, 11, None, 0.0, True, d_rgb.contiguous(), None, None, rays_d_offset=3)
    return make


def _random_upstream(raw, rd, zd, g):
    return (torch.randn(raw.shape, generator=g) * 3e-6).to(raw.device)


@pytest.mark.parametrize("n_rays,S", [(48, 64), (11, 192), (3, 5)])
def test_two_word_backward_under_a_random_upstream_gradient(npa, dev, nets, n_rays, S):
    """The worst case for operand rounding (incoherent sums do not average it down): fp16x3 sits at 2.4e-4 .. 3.5e-4 of the whole gradient
    (test_gpu_fp16x3.py: 2^-12 sqrt 2), the two-word operands at the product class of the forward."""
    rel1, worst1 = _grad_vs_fp64(npa, dev, nets, "fp16x3", n_rays, S, _random_upstream)
    rel2, worst2 = _grad_vs_fp64(npa, dev, nets, "fp16x3w", n_rays, S, _random_upstream)
    print(f"random upstream gradient, {n_rays} x {S}: whole-gradient rel. L2 vs fp64 -- fp16x3 {rel1:.2e} (worst entry {worst1:.1e}), fp16x3w {rel2:.2e} ({worst2:.1e})")
    assert rel2 <= 3e-6 and worst2 <= 1e-5, (rel2, worst2)        # measured 8.0e-7 .. 1.3e-6 (fp16x3: 2.1e-4 .. 3.4e-4)
    assert rel2 <= rel1 / 50, (rel1, rel2)


def test_two_word_backward_under_a_training_losss_upstream_gradient(npa, dev, nets):
    """98 k points, d_raw = the adjoint of raw2outputs for an MSE loss: fp16x3 1.9e-5 (bound 5e-5); two-word operands: the forward's class."""
    rel1, worst1 = _grad_vs_fp64(npa, dev, nets, "fp16x3", 512, 192, _training_upstream(npa))
    rel2, worst2 = _grad_vs_fp64(npa, dev, nets, "fp16x3w", 512, 192, _training_upstream(npa))
    print(f"training-loss upstream gradient, 98 k points: whole-gradient rel. L2 vs fp64 -- fp16x3 {rel1:.2e} (worst entry {worst1:.1e}), fp16x3w {rel2:.2e} ({worst2:.1e})")
    assert rel1 <= 5e-5
    assert rel2 <= 1.5e-6 and worst2 <= 3e-6, (rel2, worst2)       # measured 4.2e-7 / 5.6e-7 (fp16x3: 1.9e-5 / 3.1e-5)
    assert rel2 <= rel1 / 10, (rel1, rel2)


def test_two_word_deltas_are_the_split_of_the_one_word_chain(npa, dev, nets):
    """The delta chain is the same arithmetic (the stored words do not feed back into it): the hi words of every delta region are
    fp16x3's, and the lo words are remainders (|lo| <= half an ulp of hi); d_raw's tiled copy carries both words too."""
    nc, nf, Pc, Pf = nets
    hb = npa.hip_backend
    L = hb.lib()
    n, S = 40, 64
    g = torch.Generator().manual_seed(3)
    rays = orc.synthetic_rays(n, seed=5).to(dev)
    z = torch.sort(torch.rand(n, S, generator=g) * 4.0 + 2.0, -1)[0].to(dev)
    d_raw = (torch.randn(n, S, 4, generator=g) * 1e-5).to(dev)
    packed = nf.packed_params("fp16x3")
    s = torch.cuda.current_stream().cuda_stream
    deltas = {}
    for prec, split in (("fp16x3", 1), ("fp16x3w", 5)):
        _, act = hb.field_fwd(packed, rays, z, save_act=True, precision=prec)
        delta = torch.zeros(hb.delta_floats(n, S, prec), device=dev)
        assert L.nerf_field_dgrad_split(packed.data_ptr(), act.data_ptr(), d_raw.data_ptr(), n, S, delta.data_ptr(), split, s) == 0, L.nerf_last_error()
        deltas[prec] = delta
        hb.WORKSPACE.give(act)
    assert hb.buffer_layout(deltas["fp16x3w"])[0] == 4
    assert torch.equal(hb.delta_scale_word(deltas["fp16x3"], n, S), hb.delta_scale_word(deltas["fp16x3w"], n, S))
    scale = float(torch.tensor([hb.delta_scale_word(deltas["fp16x3w"], n, S).item()], dtype=torch.int32).view(torch.float32))
    assert scale > 0
    for region in [f"h{l}" for l in range(8)] + ["hv", "graw"]:
        hi1 = hb.delta_rows(deltas["fp16x3"], n, S, region, "fp16x3")
        hi = hb.delta_rows(deltas["fp16x3w"], n, S, region, "fp16x3w")
        lo = hb.delta_rows(deltas["fp16x3w"], n, S, region, "fp16x3w", part="lo")
        assert torch.equal(hi1, hi), region
        assert float((lo.abs() - 2.0 ** -11 * hi.abs().clamp_min(2.0 ** -14)).max()) <= 0, region
        assert float(lo.abs().max()) > 0, region
    # graw = s * d_raw exactly as hi + lo (the scaled maximum sits in [16, 32): 2^-22 of it)
    k = 4 - (int(np.frexp(float(d_raw.abs().max()))[1]) - 1)
    got = (hb.delta_rows(deltas["fp16x3w"], n, S, "graw", "fp16x3w").double() + hb.delta_rows(deltas["fp16x3w"], n, S, "graw", "fp16x3w", part="lo").double()).cpu()
    want = d_raw.reshape(-1, 4).cpu().double() * 2.0 ** k
    assert maxdiff(got, want) <= 2.0 ** -17, maxdiff(got, want)     # hi + lo: 2^-23 relative of values below 32


def test_two_word_backward_is_exactly_homogeneous_and_deterministic(npa, dev, nets):
    nc, nf, Pc, Pf = nets
    hb = npa.hip_backend
    n_rays, S = 33, 64
    g = torch.Generator().manual_seed(3)
    rays = orc.synthetic_rays(n_rays, seed=5).to(dev)
    z = torch.sort(torch.rand(n_rays, S, generator=g) * 4.0 + 2.0, -1)[0].to(dev)
    d_raw = (torch.randn(n_rays, S, 4, generator=g) * torch.exp(torch.randn(n_rays, S, 1, generator=g) * 3)).to(dev)
    packed = nf.packed_params("fp16x3w")
    _, act = hb.field_fwd(packed, rays, z, save_act=True, precision="fp16x3w")
    grads = {}
    for j in (0, 0.5, -40, 20):
        grad = torch.full((595844,), float("nan"), device=dev)
        hb.field_bwd(packed, act, (d_raw * 2.0 ** int(j)).contiguous(), grad, accumulate=False, precision="fp16x3w", params=nf.flat_params())
        grads[j] = grad.clone()
        assert bool(torch.isfinite(grad).all()) and float(grad.abs().max()) > 0
    assert torch.equal(grads[0.5], grads[0])                    # the same call twice: bit-identical (no atomics)
    for j in (-40, 20):
        assert torch.equal(grads[j], grads[0] * 2.0 ** j), j
    hb.WORKSPACE.give(act)


def test_two_word_datapath_through_render_and_its_subchunks(npa, dev, nets, monkeypatch):
    """render() under set_precision("fp16x3w"): images bit-identical to fp16x3's (same forward), gradients through the autograd node --
    also when the ray chunk is back-propagated in resident sub-chunks (the two-word buffers are twice the size: the planner sizes them
    with nerf_*_floats_dp(datapath = 2)) -- and the buffer records refuse a one-word / two-word mix."""
    nc, nf, Pc, Pf = nets
    hb = npa.hip_backend
    L = hb.lib()
    assert hb.max_saved_rays(64, 128, "fp16x3w") < hb.max_saved_rays(64, 128, "fp16x3")
    g32 = _flat_grads_through_render(npa, dev, 256, precision="fp32")
    g1 = _flat_grads_through_render(npa, dev, 256, precision="fp16x3")
    g2 = _flat_grads_through_render(npa, dev, 256, precision="fp16x3w")
    rel1, rel2 = float((g1 - g32).norm() / g32.norm()), float((g2 - g32).norm() / g32.norm())
    print(f"256 rays through render(): whole-gradient rel. L2 vs the fp32 datapath -- fp16x3 {rel1:.2e}, fp16x3w {rel2:.2e} "
          "(both contain the hierarchical sampling's sensitivity to the forward's 2^-22)")
    assert rel2 <= 3e-3 and rel2 <= 1.2 * rel1 + 1e-6, (rel1, rel2)
    # sub-chunks: a budget that forces 2 resident sub-chunks of 128 rays
    monkeypatch.setattr(hb, "SAVE_BUDGET_BYTES", 4 * hb.workspace_floats(128, 64, 128, True, "fp16x3w"))
    g2s = _flat_grads_through_render(npa, dev, 256, precision="fp16x3w")
    assert float((g2s - g2).norm() / g2.norm()) <= 1e-6        # (the chunk boundaries of the deterministic reduction moved)
    # a two-word dgrad on a one-word save buffer is refused
    n, S = 16, 16
    rays = orc.synthetic_rays(n, seed=1).to(dev)
    z = torch.sort(torch.rand(n, S, device=dev) * 4 + 2, -1)[0]
    packed = nf.packed_params("fp16x3")
    _, act = hb.field_fwd(packed, rays, z, save_act=True, precision="fp16x3")
    delta = torch.empty(hb.delta_floats(n, S, "fp16x3w"), device=dev)
    d_raw = torch.randn(n, S, 4, device=dev)
    s = torch.cuda.current_stream().cuda_stream
    assert L.nerf_field_dgrad_split(packed.data_ptr(), act.data_ptr(), d_raw.data_ptr(), n, S, delta.data_ptr(), 5, s) == -1
    assert b"two-word" in L.nerf_last_error()
    hb.WORKSPACE.give(act)
