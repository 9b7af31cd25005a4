"""GPU tests (-m gpu) against reference-produced fixtures at BASELINE.json's OWN batch sizes (round 4;
tests/golden/make_golden.py --round4): configs[1] one 4096-ray lego training step and configs[2] one 4096-ray fern / NDC
training step through render() (outputs per ray, loss, a digest of every gradient, each bounded against the reference's own
fp32-vs-fp64 noise), and configs[3]'s 32,768-ray batch as ONE chunk -- the shape render(chunk=32768) back-propagates through
resident sub-chunks -- forward maps and loss compared with the reference instead of with another run of this library."""
import numpy as np
import pytest
import torch

import nerf_oracle as orc
from test_gpu_parity import GOLD, GOLD_TOL, PARITY_DATAPATHS, _check_golden, dev, nets, npa  # noqa: F401  (fixtures)

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("precision", PARITY_DATAPATHS)
def test_golden_cfg2_lego_4096_ray_training_step(npa, dev, nets, precision):
    """BASELINE.json configs[1] (run_nerf.py:760-772 with configs/lego.txt: perturb = 1, white background, 64 + 128)"""
    _check_golden(npa, dev, nets, "lego_cfg2_train", dict(perturb=1.0), 1234, precision, render=(orc.LEGO, orc.lego_batch(4096, seed=17)),
                  n=4096, target_seed=98, raw_ray_stride=16)


@pytest.mark.parametrize("precision", PARITY_DATAPATHS)
def test_golden_cfg3_fern_ndc_4096_ray_training_step(npa, dev, nets, precision):
    """BASELINE.json configs[2] (configs/fern.txt through render(ndc=True): raw_noise_std = 1, no white background)"""
    _check_golden(npa, dev, nets, "fern_cfg3_train", dict(perturb=1.0, raw_noise_std=1.0, white_bkgd=False, N_importance=128), 4321, precision,
                  render=(orc.FERN, orc.fern_batch(4096, seed=13)), n=4096, target_seed=98, raw_ray_stride=16)


@pytest.mark.parametrize("precision", PARITY_DATAPATHS)
def test_golden_cfg4_32768_rays_in_one_chunk(npa, dev, nets, precision):
    """BASELINE.json configs[3]'s batch on ONE GPU: render(chunk = 32768) with gradients enabled renders the chunk in sub-chunks
    whose saved activations stay resident (render._RenderRays, ~90 GB) -- maps and loss against the REFERENCE's forward of the
    same 32,768 rays in one chunk (the reference's draw order: t_rand [32768, 64] then u [32768, 128]); then the backward
    through all sub-chunks against the sum of the gradients of eight independent 4096-ray render() calls."""
    nc, nf, Pc, Pf = nets
    T = GOLD_TOL[precision]
    gold = np.load(f"{GOLD}/lego_cfg4_forward.npz")
    cfg2 = np.load(f"{GOLD}/lego_cfg2_train.npz")
    n = 32768
    batch = orc.lego_batch(n, seed=19)
    assert abs(float(batch.double().abs().sum()) - float(gold["rays_checksum"])) < 1e-4
    target = torch.tensor(np.random.RandomState(97).rand(n, 3), dtype=torch.float32).to(dev)
    torch.manual_seed(2024)
    rnd = {"t_rand": torch.rand(n, 64), "u": torch.rand(n, 128)}
    K = orc.intrinsics(dict(orc.LEGO, H=800, W=800, focal=1111.0))
    args = dict(ndc=False, near=2.0, far=6.0, use_viewdirs=True, network_fn=nc, network_query_fn=None, N_samples=64, N_importance=128,
                network_fine=nf, perturb=1.0, white_bkgd=True, raw_noise_std=0.0)
    for m in (nc, nf):
        m.zero_grad()
    npa.set_precision(precision)
    try:
        rgb, disp, acc, ex = npa.render(800, 800, K, chunk=1024 * 32, rays=batch.to(dev), randoms=rnd, **args)
        loss = npa.img2mse(rgb, target) + npa.img2mse(ex["rgb0"], target)
        loss.backward()
        g_all = torch.cat([nc.last_flat_grad, nf.last_flat_grad]).double().cpu()
        out = {k: v.detach().cpu() for k, v in dict(rgb_map=rgb, acc_map=acc, rgb0=ex["rgb0"], acc0=ex["acc0"], z_std=ex["z_std"]).items()}
        # the same gradient from eight independent 4096-ray calls (each its own autograd node, no sub-chunking)
        g_sum = torch.zeros_like(g_all)
        for lo in range(0, n, 4096):
            for m in (nc, nf):
                m.zero_grad()
            r_, _, _, e_ = npa.render(800, 800, K, chunk=1024 * 32, rays=batch[:, lo:lo + 4096].to(dev),
                                      randoms={k: v[lo:lo + 4096] for k, v in rnd.items()}, **args)
            ((((r_ - target[lo:lo + 4096]) ** 2).sum() + ((e_["rgb0"] - target[lo:lo + 4096]) ** 2).sum()) / (3 * n)).backward()
            g_sum += torch.cat([nc.last_flat_grad, nf.last_flat_grad]).double().cpu()
    finally:
        npa.set_precision("fp32")
    report = {}
    for k in ("rgb0", "acc0"):          # coarse pass: per ray
        err = (out[k].double() - torch.tensor(gold[k]).double()).abs().reshape(n, -1).max(-1)[0]
        report[k] = float(err.max())
        assert report[k] <= T["coarse"], (k, report[k])
    for k in ("rgb_map", "acc_map", "z_std"):       # behind sample_pdf: most rays per ray (the reference's own fp32-vs-fp64 runs move
        err = (out[k].double() - torch.tensor(gold[k]).double()).abs().reshape(n, -1).max(-1)[0]     # single rays by 1e-3), all as an image
        floor = T["zstd_floor"] if k == "z_std" else T["fine_floor"]
        report[k + " p95"] = float(torch.quantile(err, 0.95))
        report[k + " max"] = float(err.max())
        assert report[k + " p95"] <= floor, (k, report)
        assert report[k + " max"] <= max(10 * float(cfg2["noise/" + k]), T["fine_max"] or 0.0, 1e-2 if k == "z_std" else 0.0), (k, report)
    mse_img = float(((out["rgb_map"].double() - torch.tensor(gold["rgb_map"]).double()) ** 2).mean())
    report["psnr_vs_ref_dB"] = orc.psnr(max(mse_img, 1e-30))
    assert report["psnr_vs_ref_dB"] >= T["img_psnr_db"], report
    report["loss"] = abs(loss.item() - float(gold["loss"]))
    assert report["loss"] <= max(T["loss_floor"], 1e-5), report
    rel = float((g_all - g_sum).norm() / g_sum.norm())
    report["grad vs 8 x 4096-ray calls (rel L2)"] = rel
    print("lego_cfg4_forward", precision, report)
    assert rel <= (1e-5 if precision == "fp32" else 1e-4), rel
