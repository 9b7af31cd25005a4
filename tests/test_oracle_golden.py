"""CPU: the oracle reproduces every golden fixture produced by the real reference, and (when the
reference tree is present, i.e. in the build container) is bit-identical to it function by function."""
import os

import numpy as np
import pytest
import torch

import nerf_oracle as orc

GOLD = os.path.join(os.path.dirname(__file__), "golden")
CASES = {
    "lego_det": (dict(), None),
    "lego_train": (dict(perturb=1.0), 123),
    "fern_train": (dict(perturb=1.0, raw_noise_std=1.0, white_bkgd=False, n_fine=64, lindisp=True), 321),
    "lego_coarse_only": (dict(perturb=1.0, n_fine=0), 11),
}


@pytest.mark.parametrize("name", sorted(CASES))
def test_oracle_reproduces_reference_golden(name):
    kw, seed = CASES[name]
    gold = np.load(os.path.join(GOLD, name + ".npz"))
    rays = orc.synthetic_rays(256, seed=7)
    target = torch.tensor(np.random.RandomState(99).rand(256, 3), dtype=torch.float32)
    Pc, Pf = orc.scene_params()
    assert abs(float(rays.double().abs().sum()) - float(gold["rays_checksum"])) < 1e-9
    pcs = sum(float(torch.cat([v.reshape(-1) for v in P.values()]).double().abs().sum()) for P in (Pc, Pf))
    assert abs(pcs - float(gold["params_checksum"])) < 1e-6
    Pc = {k: v.requires_grad_(True) for k, v in Pc.items()}
    Pf = {k: v.requires_grad_(True) for k, v in Pf.items()}
    args = dict(n_coarse=64, n_fine=128, perturb=0.0, white_bkgd=True, raw_noise_std=0.0, lindisp=False, retraw=True)
    args.update(kw)
    rnd = {}
    if seed is not None:
        torch.manual_seed(seed)
        if args["perturb"] > 0:
            rnd["t_rand"] = torch.rand(256, 64)
        if args["raw_noise_std"] > 0:
            rnd["noise_c"] = torch.randn(256, 64)
        if args["n_fine"] > 0 and args["perturb"] > 0:
            rnd["u"] = torch.rand(256, args["n_fine"])
        if args["n_fine"] > 0 and args["raw_noise_std"] > 0:
            rnd["noise_f"] = torch.randn(256, 64 + args["n_fine"])
    out = orc.trace_rays(rays, Pc, Pf if args["n_fine"] > 0 else None, **args, **rnd)
    loss = orc.mse(out["rgb_map"], target)
    if "rgb0" in out:
        loss = loss + orc.mse(out["rgb0"], target)
    loss.backward()
    assert loss.item() == float(gold["loss"])
    for k in gold.files:
        if k in out and not k.startswith("noise/"):
            got = out[k].detach()
            got = got[:, ::8] if k == "raw" else got
            a, b = got.numpy(), gold[k]
            assert np.array_equal(a, b, equal_nan=True), k
    for tag, P in (("c", Pc), ("f", Pf)):
        for nm, p in P.items():
            key = f"{tag}/{nm}/val"
            if key in gold.files:
                assert np.array_equal(p.grad.reshape(-1)[gold[f"{tag}/{nm}/idx"]].numpy(), gold[key]), (tag, nm)
                assert float(p.grad.abs().max()) == float(gold[f"{tag}/{nm}/max"])


@pytest.mark.parametrize("name,cfgname,seed,kw", [
    ("fern_ndc_train", "FERN", 77, dict(perturb=1.0, raw_noise_std=1.0, white_bkgd=False)),
    ("lego_render_train", "LEGO", 123, dict(perturb=1.0, raw_noise_std=0.0, white_bkgd=True)),
])
def test_oracle_reproduces_the_render_boundary_fixtures(name, cfgname, seed, kw):
    """The fixtures produced through the reference's render() (view directions, NDC warp, near / far): BASELINE configs[2]
    (fern, NDC) and configs[1] through the rays=... branch."""
    gold = np.load(os.path.join(GOLD, name + ".npz"))
    cfg = getattr(orc, cfgname)
    batch = orc.fern_batch(256, seed=3) if cfgname == "FERN" else orc.lego_batch(256, seed=7)
    assert abs(float(batch.double().abs().sum()) - float(gold["rays_checksum"])) < 1e-9
    target = torch.tensor(np.random.RandomState(99).rand(256, 3), dtype=torch.float32)
    Pc, Pf = orc.scene_params()
    rays = orc.assemble_render_rays(cfg["H"], cfg["W"], orc.intrinsics(cfg), batch[0], batch[1], cfg["ndc"], cfg["near"], cfg["far"])
    torch.manual_seed(seed)
    rnd = {"t_rand": torch.rand(256, 64)}
    if kw["raw_noise_std"] > 0:
        rnd["noise_c"] = torch.randn(256, 64)
    rnd["u"] = torch.rand(256, 128)
    if kw["raw_noise_std"] > 0:
        rnd["noise_f"] = torch.randn(256, 192)
    out = orc.trace_rays(rays, Pc, Pf, 64, 128, retraw=True, **kw, **rnd)
    loss = orc.mse(out["rgb_map"], target) + orc.mse(out["rgb0"], target)
    assert loss.item() == float(gold["loss"])
    for k in ("rgb_map", "disp_map", "acc_map", "rgb0", "disp0", "acc0", "z_std"):
        assert np.array_equal(out[k].numpy(), gold[k], equal_nan=True), k
    assert np.array_equal(out["raw"][:, ::8].numpy(), gold["raw"])


@pytest.mark.parametrize("which", ["lego", "fern"])
def test_gate_fixtures_are_the_references_images(which):
    """The PSNR-gate fixtures: both images come from the reference's render(); the oracle (bit-identical to it on this CPU)
    reproduces them, the target sits at a trained-NeRF PSNR, and the gate arithmetic is what workloads.precision_gate says."""
    gold = np.load(os.path.join(GOLD, f"gate_{which}.npz"))
    cfg = orc.LEGO if which == "lego" else orc.FERN
    batch = orc.lego_batch(1024, seed=31) if which == "lego" else orc.fern_batch(1024, seed=32)
    assert abs(float(batch.double().abs().sum()) - float(gold["rays_checksum"])) < 1e-5
    rays = orc.assemble_render_rays(cfg["H"], cfg["W"], orc.intrinsics(cfg), batch[0], batch[1], cfg["ndc"], cfg["near"], cfg["far"])[:128]
    for key, (Pc, Pf) in (("rgb_ref", orc.scene_params()), ("target", orc.teacher_params())):
        with torch.no_grad():
            img = orc.trace_rays(rays, Pc, Pf, 64, 128, perturb=0.0, white_bkgd=cfg["white_bkgd"])["rgb_map"]
        # (a 128-ray slice of the 1024-ray batch: the CPU GEMMs block differently, so equal to rounding, not bit for bit)
        err = np.abs(img.numpy() - gold[key][:128]).max(-1)
        assert np.quantile(err, 0.95) <= 1e-5 and err.max() <= 5e-3, (key, float(np.quantile(err, 0.95)), float(err.max()))
    g = orc.precision_gate(torch.tensor(gold["rgb_ref"]), torch.tensor(gold["rgb_ref"]), torch.tensor(gold["target"]))
    assert g["psnr_delta_db"] == 0.0 and g["target_psnr_db"] >= 30.0 and abs(g["target_psnr_db"] - float(gold["target_psnr_db"])) < 1e-6


@pytest.mark.skipif(not os.path.isdir("/root/reference"), reason="reference tree only exists in the build container")
def test_oracle_is_bit_identical_to_the_reference():
    import pin_against_reference
    pin_against_reference.main(n_rays=48)


def test_known_answers():
    """Facts about the reference established in SURVEY §8c."""
    x = torch.tensor([[0.5, -1.0, 2.0]])
    e = orc.posenc(x, 10)
    assert e.shape == (1, 63)
    assert torch.equal(e[0, :3], x[0]) and torch.allclose(e[0, 3:6], torch.sin(x[0])) and torch.allclose(e[0, 6:9], torch.cos(x[0]))
    assert torch.allclose(e[0, 57:60], torch.sin(x[0] * 512))
    assert sum(int(np.prod(s)) for _, s in orc.param_shapes()) == 595844
    z = torch.linspace(2, 6, 64)[None]
    w = torch.rand(1, 64)
    s = orc.inverse_cdf(.5 * (z[..., 1:] + z[..., :-1]), w[..., 1:-1], 128)
    assert (s[:, 1:] >= s[:, :-1]).all() and s.min() >= z[0, 0] and s.max() <= z[0, -1]
    raw = torch.full((1, 64, 4), -3.0)
    rgb, disp, acc, _, _ = orc.composite(raw, z, torch.tensor([[0., 0., -1.]]), None, True)
    assert torch.equal(rgb, torch.ones(1, 3)) and acc.item() == 0 and torch.isnan(disp).all()


@pytest.mark.parametrize("name", sorted(orc.DENSE_CASES))
def test_oracle_reproduces_reference_golden_of_other_architectures(name):
    """tests/golden/dense_*.npz: the REAL reference through render() for architectures outside the fused kernels (no view
    directions + output_linear; 6 x 128 with 6 / 2 frequencies).  The oracle's general layer stack (field_mlp_arch) must
    reproduce them bit for bit on the CPU: outputs, loss, gradient digests."""
    gold = np.load(os.path.join(GOLD, name + ".npz"))
    arch, Pc, Pf, batch, target, n_c, n_f = orc.dense_case(name)
    n = batch.shape[1]
    assert abs(float(batch.double().abs().sum()) - float(gold["rays_checksum"])) < 1e-9
    cfg = orc.LEGO
    flat = orc.assemble_render_rays(cfg["H"], cfg["W"], orc.intrinsics(cfg), batch[0], batch[1], False, 2.0, 6.0)
    rr = flat if arch["use_viewdirs"] else flat[:, :8]
    torch.manual_seed(55)
    rnd = dict(t_rand=torch.rand(n, n_c), noise_c=torch.randn(n, n_c), u=torch.rand(n, n_f), noise_f=torch.randn(n, n_c + n_f))
    Pg = [{k: v.clone().requires_grad_(True) for k, v in P.items()} for P in (Pc, Pf)]
    out = orc.trace_rays(rr, Pg[0], Pg[1], n_c, n_f, perturb=1.0, white_bkgd=True, raw_noise_std=0.5, retraw=True, arch=arch, **rnd)
    loss = orc.mse(out["rgb_map"], target) + orc.mse(out["rgb0"], target)
    loss.backward()
    assert loss.item() == float(gold["loss"])
    for k in ("rgb_map", "disp_map", "acc_map", "rgb0", "acc0", "z_std", "raw"):
        assert np.array_equal(out[k].detach().numpy(), gold[k], equal_nan=True), k
    for tag, P in (("c", Pg[0]), ("f", Pg[1])):
        for nm, p in P.items():
            key = f"grad_{tag}/{nm}/val"
            if key in gold.files:
                assert np.array_equal(p.grad.reshape(-1)[gold[f"grad_{tag}/{nm}/idx"]].numpy(), gold[key]), (tag, nm)
            else:
                assert p.grad is None, (tag, nm)
