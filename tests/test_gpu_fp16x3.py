"""GPU tests (-m gpu) of the fp16 three-term split ("fp16x3", round 4; csrc/split_types.h): the same kernels as bf16x3 with
IEEE-half (hi, lo) parts -- ~2^-22 per product, 11-bit operands for the weight-gradient GEMM, deltas scaled by a power of two
per launch.  The six reference-produced goldens, the PSNR gates, ragged batches, the one-call ABI, the one-launch inference and
the train()-shaped loop run on this datapath through the parametrised tests of test_gpu_parity.py / test_gpu_round3.py /
test_gpu_fuzz.py / test_train_loop_gpu.py; here are the bounds that are specific to it."""
import numpy as np
import pytest
import torch

import nerf_oracle as orc
from test_gpu_parity import npa, dev, nets, maxdiff, _flat_grads_through_render      # noqa: F401  (fixtures)
from test_gpu_round3 import _decode_masks, _field_with_forced_relu, GOLDEN_CASES, _golden_grads

pytestmark = pytest.mark.gpu


def _f16_split(x):
    """hi = fp16(x), lo = fp16(x - hi) on the host (numpy, round to nearest even): csrc/split_types.h SplitF16"""
    x = np.asarray(x, dtype=np.float32)
    hi = x.astype(np.float16)
    lo = (x - hi.astype(np.float32)).astype(np.float16)
    return hi, lo


def test_fp16_repack_is_the_host_split_of_every_weight(npa, dev, nets):
    """nerf_pack_params_split(split = 1): every 16-bit element of the 16-point forward stream is the fp16 hi / lo part of the
    parameter (or of the folded W' = Wv[:, :256] Wf) the gather table names -- the same table as the bf16 repack."""
    nc, nf, Pc, Pf = nets
    hb = npa.hip_backend
    L = hb.lib()
    flat = nf.flat_params()
    packed = hb.pack_params(flat, precision="fp16x3").cpu()
    tab = hb.pack_table16()
    off16 = 2 * (L.nerf_packed3_floats() - hb.N_DERIVED - hb.P16F_WORDS)          # P16F region, in 16-bit elements
    got = packed.view(torch.int16).numpy().view(np.uint16)[off16:off16 + 2 * hb.P16F_WORDS]
    derived = packed[-hb.N_DERIVED:].numpy()
    src = np.concatenate([flat.detach().cpu().numpy(), derived])
    idx, is_lo = tab >> 1, (tab & 1).astype(bool)
    hi, lo = _f16_split(src[np.maximum(idx, 0)])
    want = np.where(is_lo, lo.view(np.uint16), hi.view(np.uint16))
    want[tab < 0] = 0
    assert np.array_equal(got, want), int((got != want).sum())
    # the folded layer itself: W' in fp64 from the canonical parameters, rounded once
    Wv = Pf["views_linears.0.weight"].double()[:, :256]
    Wp = (Wv @ Pf["feature_linear.weight"].double()).float().reshape(-1).numpy()
    assert np.abs(derived[:128 * 256] - Wp).max() <= 1e-6 * max(1.0, np.abs(Wp).max())


@pytest.mark.parametrize("n_rays,S", [(64, 64), (37, 192), (5, 3), (1, 1)])
def test_field_forward_fp16x3(npa, dev, nets, n_rays, S):
    """raw vs the fp64 oracle: fp32-class (bound 2e-5 of |raw|max: 15 x tighter than bf16x3's 3e-4); inference and the saving
    forward are the same kernel (bit-identical raw); what is saved for the backward are the fp16 roundings of the activations."""
    nc, nf, Pc, Pf = nets
    hb = npa.hip_backend
    rays = orc.synthetic_rays(n_rays, seed=S)
    z = torch.sort(torch.rand(n_rays, S, generator=torch.Generator().manual_seed(S)) * 4.0 + 2.0, -1)[0]
    P = n_rays * S
    packed = nf.packed_params("fp16x3")
    raw, _ = hb.field_fwd(packed, rays.to(dev), z.to(dev), save_act=False, precision="fp16x3")
    raw2, act = hb.field_fwd(packed, rays.to(dev), z.to(dev), save_act=True, precision="fp16x3")
    assert torch.equal(raw, raw2)
    assert hb.buffer_layout(act)[0] == 5
    pts = (rays[:, None, 0:3] + rays[:, None, 3:6] * z[..., None])
    P64 = {k: v.double() for k, v in Pf.items()}
    ref64 = orc.query_field(P64, pts.double(), rays[:, 8:11].double())
    ref32 = orc.query_field(Pf, pts, rays[:, 8:11])
    scale = max(1.0, float(ref64.abs().max()))
    err, noise = maxdiff(raw, ref64), maxdiff(ref32, ref64)
    print(f"fp16x3 raw vs fp64: {err:.2e} (reference fp32 vs fp64: {noise:.2e}), |raw|max {scale:.1f}")
    assert err <= 2e-5 * scale, (err, noise, scale)
    feats = torch.cat([orc.posenc(pts.reshape(-1, 3).double(), 10), orc.posenc(rays[:, None, 8:11].expand(n_rays, S, 3).reshape(-1, 3).double(), 4)], -1)
    _, hidden, _, hv = orc.field_mlp(P64, feats, return_hidden=True)
    for l in (0, 3, 7):
        got = hb.saved_rows(act, n_rays, S, f"h{l}", precision="fp16x3").cpu().double()
        tol = 2.0 ** -11 * float(hidden[l].abs().max()) + 2e-5
        assert maxdiff(got, hidden[l]) <= tol, (l, maxdiff(got, hidden[l]), tol)
    got = hb.saved_rows(act, n_rays, S, "hv", precision="fp16x3").cpu().double()
    assert maxdiff(got, hv) <= 2.0 ** -11 * float(hv.abs().max()) + 2e-5
    got = hb.saved_rows(act, n_rays, S, "enc", precision="fp16x3").cpu().double()[:, :63]
    assert maxdiff(got, feats[:, :63]) <= 2.0 ** -11 * float(feats[:, :63].abs().max()) + 1e-6
    hb.WORKSPACE.give(act)


@pytest.mark.parametrize("n_rays,S", [(48, 64), (11, 192), (70, 20), (3, 5)])
def test_field_backward_fp16x3_arithmetic_and_flips(npa, dev, nets, n_rays, S):
    """The round-3 separation of arithmetic error from ReLU-kink flips (test_gpu_round3.py), on the fp16 split with its 16-bit
    operand storage: gradient vs fp64 autograd of the reference network evaluated with the kernel's OWN ReLU pattern, under a
    RANDOM upstream gradient -- the worst case for the zero-mean 2^-12 rounding of the stored operands (incoherent sums do not
    average it down relative to the result): <= 1e-3 of max|g| for every 256-wide tensor (measured <= 8.6e-4; bf16 storage:
    5.4e-3, bound 8e-3), <= 2e-3 for the two heads' few-entry tensors (rgb_linear: 387 entries, measured 1.4e-3); relative L2 of
    the WHOLE gradient <= 5e-4 (2^-12 sqrt(2) = 3.5e-4 is the operand rounding's size for an incoherent sum: measured 2.6e-4 .. 3.5e-4;
    bf16 storage: 2.1e-3) and
    <= 1e-3 for any single weight tensor.  Units on the other side of their kink only within 2e-5 of the layer's scale, and rare."""
    nc, nf, Pc, Pf = nets
    hb = npa.hip_backend
    g = torch.Generator().manual_seed(7 * n_rays + S)
    rays = orc.synthetic_rays(n_rays, seed=S + 1)
    z = torch.sort(torch.rand(n_rays, S, generator=g) * 4.0 + 2.0, -1)[0]
    d_raw = torch.randn(n_rays, S, 4, generator=g) * 3e-6          # the size of a training loss's upstream gradient
    P = n_rays * S
    packed = nf.packed_params("fp16x3")
    raw, act = hb.field_fwd(packed, rays.to(dev), z.to(dev), save_act=True, precision="fp16x3")
    masks = _decode_masks(npa, act, P, n_rays)
    grad = torch.full((595844,), float("nan"), device=dev)
    hb.field_bwd(packed, act, d_raw.to(dev), grad, accumulate=False, precision="fp16x3", params=nf.flat_params())
    hb.WORKSPACE.give(act)
    grad = grad.cpu().double()
    assert not torch.isnan(grad).any()
    P64 = {k: v.double().requires_grad_(True) for k, v in Pf.items()}
    pts = (rays[:, None, 0:3] + rays[:, None, 3:6] * z[..., None]).reshape(-1, 3).double()
    dirs = rays[:, None, 8:11].expand(n_rays, S, 3).reshape(-1, 3).double()
    feats = torch.cat([orc.posenc(pts, 10), orc.posenc(dirs, 4)], -1)
    out, pres = _field_with_forced_relu(P64, feats, masks)
    (out * d_raw.reshape(-1, 4).double()).sum().backward()
    worst, rel2 = {}, {}
    ref_all = torch.cat([P64[nm].grad.reshape(-1) for nm, _, _ in hb.param_table()])
    rel_all = float((grad - ref_all).norm() / ref_all.norm())
    for nm, off, shape in hb.param_table():
        gg = grad[off:off + int(np.prod(shape))].view(shape)
        r = P64[nm].grad
        worst[nm] = maxdiff(gg, r) / max(float(r.abs().max()), 1e-30)
        rel2[nm] = float((gg - r).norm() / r.norm().clamp_min(1e-300))
    n_units = flips = 0
    worst_pre = 0.0
    for pre, m in zip(pres, masks):
        diff = (pre.detach() > 0) != m
        n_units += diff.numel()
        flips += int(diff.sum())
        if diff.any():
            worst_pre = max(worst_pre, float(pre.detach().abs()[diff].max()) / max(1.0, float(pre.detach().abs().max())))
    print(f"fp16x3 backward vs fp64 with the kernel's own ReLU pattern: max|err|/max|g| {max(worst.values()):.1e} "
          f"({max(worst, key=worst.get)}), relative L2 of the whole gradient {rel_all:.1e} / of the worst weight tensor {max(v for k, v in rel2.items() if not k.endswith('bias')):.1e}; units on the other side of their kink: {flips} of {n_units}, largest |pre| among them {worst_pre:.1e}")
    small = ("rgb_linear.weight", "rgb_linear.bias", "alpha_linear.weight", "alpha_linear.bias")
    assert max(v for k, v in worst.items() if k not in small) <= 1e-3, worst
    assert max(worst.values()) <= 2e-3, worst
    assert rel_all <= 5e-4, rel_all
    assert max(v for k, v in rel2.items() if not k.endswith("bias")) <= 1e-3, rel2
    assert worst_pre <= 2e-5, worst_pre
    assert flips <= 2e-4 * n_units, (flips, n_units)


@pytest.mark.parametrize("precision", ["fp16x3", "bf16x3"])
def test_split_backward_under_a_training_losss_upstream_gradient(npa, dev, nets, precision):
    """The same fp64 comparison (the kernel's own ReLU pattern forced) with the upstream gradient a training step produces:
    d_raw = the adjoint of raw2outputs for an MSE loss on 512 rays x 192 samples (98 k points).  Its sums over points are
    COHERENT, so the zero-mean rounding of the stored weight-gradient operands averages down relative to the result: whole-gradient
    relative L2 <= 5e-5 with fp16 operands (2^-12 per element), <= 4e-4 with bf16 operands (2^-9) -- the isolated, same-forward
    measurement of what the operand storage costs (a cross-datapath comparison would measure the hierarchical
    sampling's sensitivity to forward rounding instead, see test_fp16x3_training_gradient_full_batch_vs_the_fp32_datapath)."""
    nc, nf, Pc, Pf = nets
    hb = npa.hip_backend
    n_rays, S = 512, 192
    P = n_rays * S
    g = torch.Generator().manual_seed(11)
    rays = orc.synthetic_rays(n_rays, seed=29)
    z = torch.sort(torch.rand(n_rays, S, generator=g) * 4.0 + 2.0, -1)[0]
    target = torch.rand(n_rays, 3, generator=g)
    packed = nf.packed_params(precision)
    rd, zd = rays.to(dev), z.to(dev)
    raw, act = hb.field_fwd(packed, rd, zd, save_act=True, precision=precision)
    rgb, _, _, _, _ = hb.raw2outputs(raw, zd, rd, 11, None, 0.0, True, rays_d_offset=3)
    d_rgb = (2.0 / (3 * n_rays)) * (rgb - target.to(dev))
    d_raw = hb.raw2outputs_bwd(raw, zd, rd, 11, None, 0.0, True, d_rgb.contiguous(), None, None, rays_d_offset=3)
    masks = _decode_masks(npa, act, P, n_rays)
    grad = torch.full((595844,), float("nan"), device=dev)
    hb.field_bwd(packed, act, d_raw, grad, accumulate=False, precision=precision, params=nf.flat_params())
    hb.WORKSPACE.give(act)
    grad = grad.cpu().double()
    P64 = {k: v.double().requires_grad_(True) for k, v in Pf.items()}
    d64 = d_raw.cpu().double().reshape(-1, 4)
    for lo in range(0, n_rays, 64):         # (chunks: the fp64 autograd graph of 98 k points at once is ~3 GB)
        r = rays[lo:lo + 64]
        pts = (r[:, None, 0:3] + r[:, None, 3:6] * z[lo:lo + 64, :, None]).reshape(-1, 3).double()
        dirs = r[:, None, 8:11].expand(r.shape[0], S, 3).reshape(-1, 3).double()
        feats = torch.cat([orc.posenc(pts, 10), orc.posenc(dirs, 4)], -1)
        out, _ = _field_with_forced_relu(P64, feats, [m[lo * S:(lo + 64) * S] for m in masks])
        (out * d64[lo * S:(lo + 64) * S]).sum().backward()
    ref = torch.cat([P64[nm].grad.reshape(-1) for nm, _, _ in hb.param_table()])
    rel = float((grad - ref).norm() / ref.norm())
    worst = max(maxdiff(grad[off:off + int(np.prod(shape))], P64[nm].grad.reshape(-1)) / float(P64[nm].grad.abs().max()) for nm, off, shape in hb.param_table())
    print(f"{precision}, training-loss upstream gradient, 98 k points: whole-gradient relative L2 vs fp64 (own ReLU pattern) {rel:.2e}; worst tensor max|err|/max|g| {worst:.1e}")
    assert rel <= (5e-5 if precision == "fp16x3" else 4e-4), rel
    assert worst <= (2e-4 if precision == "fp16x3" else 1.5e-3), worst


def test_delta_scale_makes_the_backward_exactly_homogeneous(npa, dev, nets):
    """The chain runs on s * d_raw with s = 2^k chosen from max|d_raw| (delta_scale_kernel), and the reduction multiplies by 2^-k:
    gradients of 2^j * d_raw are EXACTLY 2^j times the gradients of d_raw, for upstream gradients from 1e-12 to 1e+6; an all-zero
    upstream gradient gives exact zeros (no 0 * inf)."""
    nc, nf, Pc, Pf = nets
    hb = npa.hip_backend
    n_rays, S = 33, 64
    g = torch.Generator().manual_seed(3)
    rays = orc.synthetic_rays(n_rays, seed=5).to(dev)
    z = torch.sort(torch.rand(n_rays, S, generator=g) * 4.0 + 2.0, -1)[0].to(dev)
    d_raw = (torch.randn(n_rays, S, 4, generator=g) * torch.exp(torch.randn(n_rays, S, 1, generator=g) * 3)).to(dev)
    packed = nf.packed_params("fp16x3")
    _, act = hb.field_fwd(packed, rays, z, save_act=True, precision="fp16x3")
    grads = {}
    for j in (0, -40, 20, -3):
        grad = torch.full((595844,), float("nan"), device=dev)
        hb.field_bwd(packed, act, (d_raw * 2.0 ** j).contiguous(), grad, accumulate=False, precision="fp16x3", params=nf.flat_params())
        grads[j] = grad.clone()
        assert bool(torch.isfinite(grad).all()) and float(grad.abs().max()) > 0
    for j in (-40, 20, -3):
        assert torch.equal(grads[j], grads[0] * 2.0 ** j), j
    grad = torch.full((595844,), float("nan"), device=dev)
    hb.field_bwd(packed, act, torch.zeros_like(d_raw), grad, accumulate=False, precision="fp16x3", params=nf.flat_params())
    assert bool((grad == 0).all())
    hb.WORKSPACE.give(act)


def _compare_with_the_fp32_datapath(g, g32, label):
    rel = float((g - g32).norm() / g32.norm())
    cosdef = 1.0 - float((g * g32).sum() / (g.norm() * g32.norm()))
    print(f"{label}: relative L2 {rel:.2e}, cosine deficit {cosdef:.1e}")
    return rel, cosdef


def test_fp16x3_training_gradient_full_batch_vs_the_fp32_datapath(npa, dev):
    """BASELINE configs[1] batch (4096 rays x (64+128)): the gradient of the training loss on the fp16 split (fp16 operand storage)
    against the EXACT-fp32 datapath's, next to the split-bf16 datapath against the same reference.  (Not one split datapath against
    another: the hierarchical samples depend on the coarse pass's rounding, so two different forwards differ by sample positions,
    not by gradient arithmetic -- measured 6.6e-4 between fp16x3 and bf16x3.)  What the distance contains: the fp16 operand
    rounding (2^-12, zero-mean, averaged over 262 k / 786 k points; isolated: test_split_backward_under_a_training_losss_upstream_
    gradient), the chain's 2^-22 products and the sampling's sensitivity to forward rounding at the 2^-22 level."""
    g32 = _flat_grads_through_render(npa, dev, 4096, precision="fp32")
    rel, cosdef = _compare_with_the_fp32_datapath(_flat_grads_through_render(npa, dev, 4096, precision="fp16x3"), g32, "4096 rays, fp16x3 vs fp32 datapath")
    rel_b, _ = _compare_with_the_fp32_datapath(_flat_grads_through_render(npa, dev, 4096), g32, "4096 rays, bf16x3 vs fp32 datapath")
    assert rel <= 6e-4 and cosdef <= 2e-7, (rel, cosdef)          # measured 3.5e-4 (bf16x3: 8.4e-4)
    assert rel < rel_b, (rel, rel_b)


@pytest.mark.parametrize("case", range(6))
def test_fp16x3_gradient_on_every_golden_configuration_vs_the_fp32_datapath(npa, dev, nets, case):
    """The same comparison on the six golden configurations (256 rays: 16 k / 49 k points per contraction): fp16x3 vs the exact-fp32
    datapath, whole-gradient relative L2, next to bf16x3."""
    name, kw, seed, through = GOLDEN_CASES[case]
    render = {None: None, "fern": (orc.FERN, orc.fern_batch(256, seed=3)), "lego": (orc.LEGO, orc.lego_batch(256, seed=7))}[through]
    g32 = _golden_grads(npa, dev, nets, kw, seed, render, precision="fp32")
    rel, cosdef = _compare_with_the_fp32_datapath(_golden_grads(npa, dev, nets, kw, seed, render, precision="fp16x3"), g32, f"{name}: fp16x3 vs fp32 datapath")
    rel_b, _ = _compare_with_the_fp32_datapath(_golden_grads(npa, dev, nets, kw, seed, render), g32, f"{name}: bf16x3 vs fp32 datapath")
    # measured 3.9e-5 .. 2.8e-4, and 2.0e-3 on the two lego_train configurations (bf16x3 there: 3.0e-3): one ray with fine samples in bins
    # the coarse pass found empty (helpers:234-236) moves under any forward rounding -- the reference's own fp32 and fp64 runs differ likewise
    assert rel <= 3e-3 and cosdef <= 3e-6 and rel <= 1.2 * rel_b, (name, rel, cosdef, rel_b)


def test_fp16x3_overflow_is_loud(npa, dev, nets):
    """Range of the fp16 split: an activation above 65504 cannot be represented; it must surface as a non-finite `raw`, never as
    a silently wrong finite value (docs: set_precision('bf16x3') is the unlimited-range datapath)."""
    nc, nf, Pc, Pf = nets
    hb = npa.hip_backend
    kw = dict(D=8, W=256, input_ch=63, output_ch=5, skips=[4], input_ch_views=27, use_viewdirs=True)
    big = npa.NeRF(**kw).to(dev)
    P = {k: v.clone() for k, v in Pf.items()}
    P["pts_linears.0.bias"] = P["pts_linears.0.bias"] + 1.0e5           # every layer-0 activation ~1e5
    big.load_state_dict(P)
    rays = orc.synthetic_rays(8, seed=1).to(dev)
    z = torch.sort(torch.rand(8, 16, generator=torch.Generator().manual_seed(1)) * 4.0 + 2.0, -1)[0].to(dev)
    raw, _ = hb.field_fwd(big.packed_params("fp16x3"), rays, z, save_act=False, precision="fp16x3")
    assert not bool(torch.isfinite(raw).all())
    raw_b, _ = hb.field_fwd(big.packed_params("bf16x3"), rays, z, save_act=False, precision="bf16x3")
    assert bool(torch.isfinite(raw_b).all())


# ---------------------------------------------------------------- reduced inference class: fp16 main term + fp8 correction terms
@pytest.mark.parametrize("n_rays,S", [(64, 64), (37, 192), (5, 3), (1, 1), (129, 70)])
def test_reduced_forward_against_fp64_and_the_last_sample_guard(npa, dev, nets, n_rays, S):
    """set_precision("fp16_fp8c"), no_grad: every product of the 256-wide layers = W_hi16 x_hi16 (fp16 MFMA) + W_hi8 x_lo8 +
    W_lo8 x_hi8 (fp8 e4m3 MFMAs of K = 128, power-of-two scales) -- csrc/field_ring8.h.  raw vs the fp64 oracle within 2e-4 of
    |raw|max (the class: ~2^-15 per product; fp16x3 holds 2e-5, bf16x3 3e-4), and every ray's LAST sample is the fp16x3 value bit
    for bit (nerf_field_fwd_last_sample: the reference's dists[-1] = 1e10 turns that sample's sign into a step of the opacity)."""
    nc, nf, Pc, Pf = nets
    hb = npa.hip_backend
    rays = orc.synthetic_rays(n_rays, seed=S)
    z = torch.sort(torch.rand(n_rays, S, generator=torch.Generator().manual_seed(S)) * 4.0 + 2.0, -1)[0]
    raw3, _ = hb.field_fwd(nf.packed_params("fp16x3"), rays.to(dev), z.to(dev), save_act=False, precision="fp16x3")
    raw8, _ = hb.field_fwd(nf.packed_params("fp16_fp8c"), rays.to(dev), z.to(dev), save_act=False, precision="fp16_fp8c",
                           guard_packed=nf.packed_params("fp16x3"))
    assert torch.equal(raw8[:, -1], raw3[:, -1])
    pts = (rays[:, None, 0:3] + rays[:, None, 3:6] * z[..., None])
    ref64 = orc.query_field({k: v.double() for k, v in Pf.items()}, pts.double(), rays[:, 8:11].double())
    scale = max(1.0, float(ref64.abs().max()))
    err8, err3 = maxdiff(raw8, ref64), maxdiff(raw3, ref64)
    print(f"reduced class raw vs fp64: {err8:.2e} (fp16x3: {err3:.2e}), |raw|max {scale:.1f}")
    assert err8 <= 2e-4 * scale, (err8, scale)
    if S > 1:
        assert not torch.equal(raw8[:, :-1], raw3[:, :-1])          # (it IS another product class)
    with pytest.raises(hb.NerfHipError):
        hb.field_fwd(nf.packed_params("fp16_fp8c"), rays.to(dev), z.to(dev), save_act=True, precision="fp16_fp8c", guard_packed=nf.packed_params("fp16x3"))


@pytest.mark.parametrize("precision", ["fp16x3", "bf16x3"])
def test_pair_repack_equals_two_repacks(npa, dev, nets, precision):
    """nerf_pack_params_split_pair (both networks in the two launches one takes, blockIdx.y = network) writes, word for word, what two
    nerf_pack_params_split calls write; field.packed_params_pair uses it exactly when both cached repacks are stale."""
    nc, nf, Pc, Pf = nets
    hb = npa.hip_backend
    fa, fb = nc.flat_params(), nf.flat_params()
    one_a, one_b = hb.pack_params(fa, precision=precision), hb.pack_params(fb, precision=precision)
    pair_a, pair_b = hb.pack_params_pair(fa, fb, precision)
    assert torch.equal(one_a.view(torch.int32), pair_a.view(torch.int32)) and torch.equal(one_b.view(torch.int32), pair_b.view(torch.int32))
    assert not torch.equal(pair_a.view(torch.int32), pair_b.view(torch.int32))
    from nerf_pytorch_amd.field import packed_params_pair
    for m in (nc, nf):
        m.invalidate_packed()
    pa, pb = packed_params_pair(nc, nf, precision)
    assert torch.equal(pa.view(torch.int32), one_a.view(torch.int32)) and torch.equal(pb.view(torch.int32), one_b.view(torch.int32))
    assert packed_params_pair(nc, nf, precision)[0] is pa and nc.packed_params(precision) is pa and nf.packed_params(precision) is pb      # cached
    nf.invalidate_packed()                                     # only one of the two stale: the single repack
    pa2, pb2 = packed_params_pair(nc, nf, precision)
    assert pa2 is pa and pb2 is not pb and torch.equal(pb2.view(torch.int32), one_b.view(torch.int32))
    same = packed_params_pair(nc, nc, precision)
    assert same[0] is same[1] is pa


def test_reduced_class_renders_without_gradients_and_trains_on_fp16x3(npa, dev, nets):
    """render_rays under set_precision("fp16_fp8c"): no_grad -> the reduced products (close to, not equal to, fp16x3's image);
    with gradients enabled -> the fp16x3 datapath itself: outputs and both networks' gradients bit-identical to fp16x3's."""
    nc, nf, Pc, Pf = nets
    rays = orc.synthetic_rays(200, seed=77).to(dev)
    target = torch.rand(200, 3, generator=torch.Generator().manual_seed(1)).to(dev)
    kw = dict(N_samples=64, N_importance=128, network_fine=nf, white_bkgd=True, retraw=True)
    out, grads = {}, {}
    for prec in ("fp16x3", "fp16_fp8c"):
        npa.set_precision(prec)
        try:
            with torch.no_grad():
                out[prec] = npa.render_rays(rays, nc, None, **kw)
            for m in (nc, nf):
                m.zero_grad()
            o = npa.render_rays(rays, nc, None, **kw)
            (npa.img2mse(o["rgb_map"], target) + npa.img2mse(o["rgb0"], target)).backward()
            grads[prec] = (o["rgb_map"].detach().clone(), nc.last_flat_grad.clone(), nf.last_flat_grad.clone())
        finally:
            npa.set_precision("fp32")
    for a, b in zip(grads["fp16x3"], grads["fp16_fp8c"]):
        assert torch.equal(a, b)
    # round 5: with a refining pass the COARSE pass of the reduced class runs on the three-term products (render._field_pass: sample_pdf
    # amplifies 2^-15 coarse errors) -- its outputs and the sampled depths are fp16x3's, bit for bit; the refining pass is the reduced one
    for k in ("rgb0", "acc0", "disp0", "z_std"):
        assert torch.equal(out["fp16_fp8c"][k].view(torch.int32), out["fp16x3"][k].view(torch.int32)), k        # (bit patterns: disp is NaN on empty rays)
    d = maxdiff(out["fp16_fp8c"]["rgb_map"], out["fp16x3"]["rgb_map"])
    assert 0.0 < d <= 3e-4, d
    ref = orc.trace_rays(rays.cpu(), Pc, Pf, 64, 128, white_bkgd=True)
    assert maxdiff(out["fp16_fp8c"]["rgb0"], ref["rgb0"]) <= 3e-5
    # without a refining pass the one pass IS the reduced one (last sample guarded)
    npa.set_precision("fp16_fp8c")
    try:
        with torch.no_grad():
            co8 = npa.render_rays(rays, nc, None, N_samples=64, N_importance=0, white_bkgd=True, retraw=True)
        npa.set_precision("fp16x3")
        with torch.no_grad():
            co3 = npa.render_rays(rays, nc, None, N_samples=64, N_importance=0, white_bkgd=True, retraw=True)
    finally:
        npa.set_precision("fp32")
    d0 = maxdiff(co8["rgb_map"], co3["rgb_map"])
    assert 0.0 < d0 <= 3e-4, d0
    assert torch.equal(co8["raw"][:, -1], co3["raw"][:, -1])           # the guard: every ray's last sample on the three-term products


@pytest.mark.parametrize("perturb,lindisp", [(0.0, False), (1.0, False), (1.0, True)])
def test_one_guard_launch_for_both_passes(npa, dev, nets, perturb, lindisp):
    """The guard launch of the coarse pass also evaluates the fine pass's last sample (nerf_field_fwd_last_sample(packed3_next)):
    sample_pdf draws inside [z_mid[0], z_mid[-1]] (helpers:196-239), so the sorted union of run_nerf.py:396 ends with the coarse
    pass's last depth -- checked here on the sampled depths themselves -- and the chain with ONE guard launch returns, bit for bit,
    what the chain with one guard launch per pass returns."""
    nc, nf, Pc, Pf = nets
    hb = npa.hip_backend
    n, Sc, Sf = 300, 64, 128
    rays = orc.synthetic_rays(n, seed=5).to(dev)
    g = torch.Generator().manual_seed(11)
    t_rand = torch.rand(n, Sc, generator=g).to(dev) if perturb > 0 else None
    u = torch.rand(n, Sf, generator=g).to(dev) if perturb > 0 else None
    t_lin = torch.linspace(0., 1., Sc, device=dev)
    z_c = hb.sample_coarse(rays, t_lin, lindisp, t_rand)
    p8c, p8f, p16c, p16f = nc.packed_params("fp16_fp8c"), nf.packed_params("fp16_fp8c"), nc.packed_params("fp16x3"), nf.packed_params("fp16x3")

    def chain(one_launch):
        raw_f = torch.full((n, Sc + Sf, 4), float("nan"), device=dev) if one_launch else None
        raw_c, _ = hb.field_fwd(p8c, rays, z_c, precision="fp16_fp8c", guard_packed=p16c, next_guard=(p16f, raw_f) if one_launch else None)
        _, _, _, w, _ = hb.raw2outputs(raw_c, z_c, rays, rays.shape[1], None, 0.0, True, want_weights=True, want_depth=False, rays_d_offset=3)
        z_f, _, _ = hb.sample_fine(z_c, w, Sf, u, None if u is not None else torch.linspace(0., 1., Sf, device=dev))
        raw_f, _ = hb.field_fwd(p8f, rays, z_f, precision="fp16_fp8c", guard_packed="done" if one_launch else p16f, raw=raw_f)
        return raw_c, z_f, raw_f
    rc1, zf1, rf1 = chain(True)
    rc2, zf2, rf2 = chain(False)
    assert torch.equal(zf1[:, -1], z_c[:, -1]) and torch.equal(zf1, zf2)
    assert torch.equal(rc1, rc2) and torch.equal(rf1, rf2) and bool(torch.isfinite(rf1).all())
    raw3, _ = hb.field_fwd(p16f, rays, zf1, precision="fp16x3")
    assert torch.equal(rf1[:, -1], raw3[:, -1])


def test_range_guard_rail_warns_before_the_nan(npa, dev, nets):
    """The fp16 split's cliff (|activation| >= 65520 -> NaN in `raw`) has a guard rail (round 6; the reference's DEBUG-gated NaN / Inf
    check is run_nerf.py:414-416): nerf_range_scan over what a training forward saved, polled without synchronisation.  A network
    whose layer 6 is scaled until its activations reach ~40,000 (finite, but past half the range; layer 7's weights are scaled down by the
    same factor, so the function and the deltas' range are unchanged -- scaling a layer alone overflows the DELTA chain first, which the
    monitor reports as well) trains three (small) steps with FINITE outputs and the warning names set_precision("bf16x3"); the healthy
    network trains silently; a NaN already in the saved rows reports inf."""
    import warnings
    nc, nf, Pc, Pf = nets
    hb = npa.hip_backend
    kw = dict(D=8, W=256, input_ch=63, output_ch=5, skips=[4], input_ch_views=27, use_viewdirs=True)
    n = 96
    rays = orc.synthetic_rays(n, seed=21).to(dev)
    target = torch.rand(n, 3, generator=torch.Generator().manual_seed(1)).to(dev)
    z = torch.sort(torch.rand(n, 64, generator=torch.Generator().manual_seed(2)) * 4.0 + 2.0, -1)[0].to(dev)

    def scaled_nets(scale):
        ncs, nfs = npa.NeRF(**kw).to(dev), npa.NeRF(**kw).to(dev)
        ncs.load_state_dict(Pc)
        nfs.load_state_dict(Pf)
        with torch.no_grad():       # h6 -> scale * h6 (ReLU is positively homogeneous), everything behind it unchanged
            for m in (ncs, nfs):
                m.pts_linears[6].weight.mul_(scale)
                m.pts_linears[6].bias.mul_(scale)
                m.pts_linears[7].weight.mul_(1.0 / scale)
        return ncs, nfs

    def largest_activation(scale):
        """the monitor's own reading of one training-mode render (both networks, the points render_rays really evaluates)"""
        hb.RANGE_MONITOR = hb.RangeMonitor()
        hb.RANGE_MONITOR.every = 1
        ncs, nfs = scaled_nets(scale)
        import warnings as _w
        with _w.catch_warnings():
            _w.simplefilter("ignore")
            npa.render_rays(rays, ncs, None, 64, N_importance=128, network_fine=nfs, white_bkgd=True, perturb=0.)
            return npa.check_range()["max_activation"]
    prev_monitor, prev_prec = hb.RANGE_MONITOR, npa.get_precision()
    npa.set_precision("fp16x3")
    try:
        # the scale that puts layer 6's largest activation at 40,000: h6 is positive-homogeneous in layer 6's weights and bias, so
        # two readings fix it (the first scale keeps h6 below 40,000 whatever layer holds the healthy maximum)
        m_all = largest_activation(1.0)
        assert 0.1 < m_all < 1000.0, m_all
        s0 = 40000.0 / m_all
        m1 = largest_activation(s0)
        assert m_all < m1 <= 40000.0 * 1.001, (m_all, m1)         # (perturb = 0: the same sample points in every render of this test)
        m7 = m1 / s0                                                # largest h6 of the healthy networks on these rays
        for scale, expect in ((1.0, False), (40000.0 / m7, True)):
            hb.RANGE_MONITOR = hb.RangeMonitor()
            hb.RANGE_MONITOR.every = 1
            ncs, nfs = scaled_nets(scale)
            # (a small learning rate: with gradients of this scale Adam moves every earlier layer by lr per weight in one coherent
            # direction -- at 5e-4 the second step is already past 65504, which is exactly the run the guard rail is for)
            opt = npa.FlatAdam(list(ncs.parameters()) + list(nfs.parameters()), lr=1e-6)
            with warnings.catch_warnings(record=True) as caught:
                warnings.simplefilter("always")
                for step in range(3):
                    opt.zero_grad()
                    out = npa.render_rays(rays, ncs, None, 64, N_importance=128, network_fine=nfs, white_bkgd=True, perturb=0., retraw=True)
                    loss = npa.img2mse(out["rgb_map"], target) + npa.img2mse(out["rgb0"], target)
                    loss.backward()
                    opt.step()
                    assert bool(torch.isfinite(out["raw"]).all()) and bool(torch.isfinite(loss)), (scale, step)
                rep = npa.check_range()
            msgs = [str(w.message) for w in caught if issubclass(w.category, RuntimeWarning)]
            print(f"layer 6 x {scale:.4g} (layer 7 / the same): largest saved activation {rep['max_activation']:.5g}, {len(msgs)} warning(s)")
            assert 0.0 < rep["max_scaled_delta"] < 32768.0, rep         # the delta chains of the scanned steps were scanned too
            if expect:
                assert 32768.0 <= rep["max_activation"] < 65504.0 and rep["warnings"] >= 1
                assert msgs and "bf16x3" in msgs[0] and "set_precision" in msgs[0] and "activation" in msgs[0], msgs
            else:
                assert rep["max_activation"] < 32768.0 and rep["warnings"] == 0 and not msgs, (rep, msgs)
                assert rep["max_activation"] >= 0.9 * m_all       # (the scan really read the rows)
        # a layer scaled ALONE overflows the delta chain before the forward: the monitor names the deltas
        hb.RANGE_MONITOR = hb.RangeMonitor()
        hb.RANGE_MONITOR.every = 1
        lone = npa.NeRF(**kw).to(dev)
        lone.load_state_dict(Pf)
        with torch.no_grad():
            lone.pts_linears[7].weight.mul_(3000.0)
        with warnings.catch_warnings(record=True) as caught:
            warnings.simplefilter("always")
            out = npa.render_rays(rays, lone, None, 64, N_importance=0, white_bkgd=True, perturb=0., retraw=True)
            npa.img2mse(out["rgb_map"], target).backward()
            rep = npa.check_range()
        print(f"layer 7's weights alone x 3000: largest activation {rep['max_activation']:.5g}, largest scaled delta {rep['max_scaled_delta']:.5g}")
        assert rep["max_scaled_delta"] >= 32768.0 and any("scaled delta" in str(w.message) for w in caught), rep
        # a NaN / inf already in the rows (the cliff itself) reports inf
        hb.RANGE_MONITOR = hb.RangeMonitor()
        hb.RANGE_MONITOR.every = 1
        big = scaled_nets(1e6 / m7)[1]
        with warnings.catch_warnings(record=True) as caught:
            warnings.simplefilter("always")
            out = npa.render_rays(rays, big, None, 64, N_importance=0, white_bkgd=True, retraw=True)
            out["rgb_map"].sum().backward()
            rep = npa.check_range()
        assert rep["max_activation"] == float("inf") and rep["warnings"] >= 1 and not bool(torch.isfinite(out["raw"]).all())
        # the scan refuses buffers that have no fp16 range to check
        _, act32 = hb.field_fwd(nf.packed_params("fp32"), rays, z, save_act=True, precision="fp32")
        w = torch.zeros(2, dtype=torch.int32, device=dev)
        assert hb.lib().nerf_range_scan(act32.data_ptr(), n, 64, w.data_ptr(), None) == -1
        hb.WORKSPACE.give(act32)
    finally:
        hb.RANGE_MONITOR = prev_monitor
        npa.set_precision(prev_prec)
        for m in (nc, nf):
            m.zero_grad()


@pytest.mark.parametrize("precision", ["fp16x3", "fp16x3w"])
def test_merged_alpha_row_equals_the_thirteen_job_plan(npa, dev, nets, precision, monkeypatch):
    """round 5 moved the alpha head's row onto the view layer's weight-gradient job (12 jobs x 21 chunks instead of
    13 x 19) and re-recorded the gradient digests in the same round.  The anchor the digests lost: the same launch on the 13-job plan
    (NERF_WG_MERGE_ALPHA=0, read per call) -- alpha_linear's and views_linears.0's gradients agree to fp32 summation order (the point
    chunks moved: 21 vs 19 per job), every other tensor likewise."""
    nc, nf, Pc, Pf = nets
    hb = npa.hip_backend
    n, S = 96, 192
    g = torch.Generator().manual_seed(5)
    rays = orc.synthetic_rays(n, seed=3).to(dev)
    z = torch.sort(torch.rand(n, S, generator=g) * 4.0 + 2.0, -1)[0].to(dev)
    d_raw = (torch.randn(n, S, 4, generator=g) * 1e-5).to(dev)
    packed = nf.packed_params(precision)
    grads = {}
    for merged in ("1", "0"):
        monkeypatch.setenv("NERF_WG_MERGE_ALPHA", merged)
        _, act = hb.field_fwd(packed, rays, z, save_act=True, precision=precision)
        grad = torch.full((595844,), float("nan"), device=dev)
        hb.field_bwd(packed, act, d_raw, grad, accumulate=False, precision=precision, params=nf.flat_params())
        hb.WORKSPACE.give(act)
        grads[merged] = grad.clone()
    monkeypatch.delenv("NERF_WG_MERGE_ALPHA")
    assert not torch.equal(grads["1"], grads["0"])              # (different chunking: the switch did something)
    for nm, off, shape in hb.param_table():
        a, b = grads["1"][off:off + int(np.prod(shape))], grads["0"][off:off + int(np.prod(shape))]
        assert bool(torch.isfinite(a).all())
        assert maxdiff(a, b) <= 3e-6 * float(b.abs().max()) + 1e-30, (nm, maxdiff(a, b) / float(b.abs().max()))
