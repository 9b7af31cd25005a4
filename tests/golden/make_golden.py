"""Generate tests/golden/*.npz from the REAL reference (build container only: needs /root/reference).

Runs the reference's render_rays (+ loss + backward) on the seeded synthetic workload every
test uses (oracle.synthetic_rays / scene_params: numpy MT19937, version independent) and stores
the outputs, so the GPU box -- where /root/reference does not exist -- can still compare against
reference-produced numbers.  Parameters and rays are NOT stored (regenerated from seeds); a
checksum of both is, so a drifted generator is detected instead of silently mis-compared.

    python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import nerf_oracle as orc  # noqa: E402
from pin_against_reference import load_reference, reference_networks  # noqa: E402

N_RAYS = 256
GRAD_SAMPLES = 64


def checksum(t):
    return float(t.double().abs().sum())


def grad_digest(named_grads, named_grads64):
    """per tensor: max |g|, fp32-vs-fp64 noise of the reference itself, GRAD_SAMPLES strided entries"""
    out = {}
    for k, g in named_grads.items():
        flat = g.reshape(-1)
        idx = np.linspace(0, flat.numel() - 1, num=min(GRAD_SAMPLES, flat.numel())).astype(np.int64)
        out[k + "/max"] = np.float64(flat.abs().max())
        out[k + "/noise"] = np.float64((flat.double() - named_grads64[k].reshape(-1)).abs().max())
        out[k + "/idx"] = idx
        out[k + "/val"] = flat[idx].numpy()
    return out


def oracle_fp64(rays, Pc, Pf, target, kw, rnd, chunk=512):
    """The same computation in float64 (oracle == reference bit for bit in fp32, see pin_against_reference):
    its distance from the fp32 reference is the reference's own rounding noise on these inputs.  Rays are independent and the
    loss is a mean, so large batches run in chunks of `chunk` rays whose gradients accumulate (memory: a 4096-ray batch
    in fp64 with autograd would hold ~50 GB)."""
    P64c = {k: v.double().requires_grad_(True) for k, v in Pc.items()}
    P64f = {k: v.double().requires_grad_(True) for k, v in Pf.items()}
    n_f = kw["N_importance"]
    n = rays.shape[0]
    pieces = {}
    for i in range(0, n, chunk):
        out = orc.trace_rays(rays[i:i + chunk].double(), P64c, P64f if n_f > 0 else None, 64, n_f, perturb=kw["perturb"],
                             lindisp=kw["lindisp"], white_bkgd=kw["white_bkgd"], raw_noise_std=kw["raw_noise_std"],
                             retraw=True, **{k: v[i:i + chunk].double() for k, v in rnd.items()})
        t = target[i:i + chunk].double()
        loss = ((out["rgb_map"] - t) ** 2).sum() / (n * 3)
        if "rgb0" in out:
            loss = loss + ((out["rgb0"] - t) ** 2).sum() / (n * 3)
        loss.backward()
        for k, v in out.items():
            pieces.setdefault(k, []).append(v.detach())
    out = {k: torch.cat(v, 0) for k, v in pieces.items()}
    return out, {k: v.grad for k, v in P64c.items() if v.grad is not None}, {k: v.grad for k, v in P64f.items() if v.grad is not None}


def draw_randoms(seed, kw, n=N_RAYS):
    """The reference's draw order (run_nerf.py:371, :285, helpers:208, :285) replayed on the CPU generator."""
    rnd = {}
    if seed is None:
        return rnd
    torch.manual_seed(seed)
    n_f = kw["N_importance"]
    if kw["perturb"] > 0:
        rnd["t_rand"] = torch.rand(n, 64)
    if kw["raw_noise_std"] > 0:
        rnd["noise_c"] = torch.randn(n, 64)
    if n_f > 0 and kw["perturb"] > 0:
        rnd["u"] = torch.rand(n, n_f)
    if n_f > 0 and kw["raw_noise_std"] > 0:
        rnd["noise_f"] = torch.randn(n, 64 + n_f)
    return rnd


def run_case(name, run_nerf, helpers, rays, nets, Pc, Pf, target, **cfg):
    """rays: [N,11] records fed to the reference's render_rays, or -- with cfg['render'] = dict(H, W, K, ndc, near, far)
    -- a [2,N,3] (rays_o, rays_d) batch fed to the reference's render() (run_nerf.py:69-134), whose ray assembly
    (view directions, NDC warp) is then part of what the fixture pins."""
    net_c, net_f = nets
    for n in nets:
        n.zero_grad()
    embed_fn, _ = helpers.get_embedder(10, 0)
    embeddirs_fn, _ = helpers.get_embedder(4, 0)
    qfn = lambda inputs, viewdirs, network_fn: run_nerf.run_network(
        inputs, viewdirs, network_fn, embed_fn=embed_fn, embeddirs_fn=embeddirs_fn, netchunk=1024 * 64)
    kw = dict(network_fn=net_c, network_query_fn=qfn, N_samples=64, retraw=True, N_importance=128,
              network_fine=net_f, lindisp=False, perturb=0.0, white_bkgd=True, raw_noise_std=0.0)
    kw.update(cfg.get("kw", {}))
    seed = cfg.get("seed")
    if seed is not None:
        torch.manual_seed(seed)
    rcfg = cfg.get("render")
    if rcfg is None:
        out = run_nerf.render_rays(rays, **kw)
        rays64 = rays.double()
    else:
        rgb, disp, acc, extras = run_nerf.render(rcfg["H"], rcfg["W"], rcfg["K"], chunk=1024 * 32, rays=rays, ndc=rcfg["ndc"],
                                                 near=rcfg["near"], far=rcfg["far"], use_viewdirs=True, **kw)
        out = dict(extras, rgb_map=rgb, disp_map=disp, acc_map=acc)
        o64, d64 = rays[0].double(), rays[1].double()
        vd = d64 / torch.norm(d64, dim=-1, keepdim=True)
        if rcfg["ndc"]:
            o64, d64 = orc.ndc_warp(rcfg["H"], rcfg["W"], rcfg["K"][0][0], 1.0, o64, d64)
        one = torch.ones_like(d64[..., :1])
        rays64 = torch.cat([o64, d64, rcfg["near"] * one, rcfg["far"] * one, vd], -1)
    loss = helpers.img2mse(out["rgb_map"], target)
    if "rgb0" in out:
        loss = loss + helpers.img2mse(out["rgb0"], target)
    loss.backward()
    out64, g64c, g64f = oracle_fp64(rays64, Pc, Pf, target, kw, draw_randoms(seed, kw, n=rays64.shape[0]))
    rec = {"loss": np.float64(loss.item()), "rays_checksum": checksum(rays),
           "params_checksum": checksum(torch.cat([v.reshape(-1) for v in Pc.values()])) +
           checksum(torch.cat([v.reshape(-1) for v in Pf.values()]))}
    for k, v in out.items():
        v = v.detach()
        d = (v.double() - out64[k].detach())
        both_nan = torch.isnan(v) & torch.isnan(out64[k].detach())
        rec["noise/" + k] = np.float64(d.abs().masked_fill(both_nan, 0.0).max())
        rec[k] = (v[::cfg.get("raw_ray_stride", 1), ::8] if k == "raw" else v).numpy()
    rec.update({"c/" + k: v for k, v in grad_digest({k: p.grad for k, p in net_c.named_parameters() if p.grad is not None}, g64c).items()})
    rec.update({"f/" + k: v for k, v in grad_digest({k: p.grad for k, p in net_f.named_parameters() if p.grad is not None}, g64f).items()})
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **rec)
    noise = {k[6:]: float(rec[k]) for k in rec if k.startswith("noise/")}
    gnoise = max(float(rec[k]) / max(float(rec[k[:-5] + "max"]), 1e-30) for k in rec if k.endswith("/noise"))
    print(f"{name}: loss {loss.item():.6f} -> {os.path.basename(path)} ({os.path.getsize(path)} B); reference fp32-vs-fp64 noise {noise}; "
          f"worst grad noise/max {gnoise:.2e}")


def main():
    run_nerf, helpers = load_reference()
    Pc, Pf = orc.scene_params()
    nets = (reference_networks(helpers, Pc), reference_networks(helpers, Pf))
    rays = orc.synthetic_rays(N_RAYS, seed=7)
    target = torch.tensor(np.random.RandomState(99).rand(N_RAYS, 3), dtype=torch.float32)
    # 1. test-time configuration of render_kwargs_test (run_nerf.py:255-257): deterministic
    run_case("lego_det", run_nerf, helpers, rays, nets, Pc, Pf, target)
    # 2. lego training configuration (configs/lego.txt: perturb=1, white_bkgd) with torch RNG seed 123:
    #    the GPU test replays the same CPU generator stream to obtain identical t_rand / u
    run_case("lego_train", run_nerf, helpers, rays, nets, Pc, Pf, target, seed=123, kw=dict(perturb=1.0))
    # 3. fern-like configuration (configs/fern.txt: raw_noise_std=1, no white_bkgd, N_importance=64), lindisp on
    run_case("fern_train", run_nerf, helpers, rays, nets, Pc, Pf, target, seed=321,
             kw=dict(perturb=1.0, raw_noise_std=1.0, white_bkgd=False, N_importance=64, lindisp=True))
    # 4. config 1 of BASELINE.json: coarse only (N_importance=0)
    run_case("lego_coarse_only", run_nerf, helpers, rays, nets, Pc, Pf, target, seed=11,
             kw=dict(perturb=1.0, N_importance=0, network_fine=None))


def gate_case(name, run_nerf, helpers, batch, cfg, n_importance=128):
    """PSNR-gate fixture: the reference's render() (test-time configuration: perturb=0, raw_noise_std=0,
    run_nerf.py:255-257) of `batch` with the networks under test (scene_params) and with the teacher scene
    (teacher_params) whose image is the target.  Stored: both images; bench.py and the GPU tests compute
    workloads.precision_gate(our image, reference image, target) from them."""
    embed_fn, _ = helpers.get_embedder(10, 0)
    embeddirs_fn, _ = helpers.get_embedder(4, 0)
    qfn = lambda inputs, viewdirs, network_fn: run_nerf.run_network(
        inputs, viewdirs, network_fn, embed_fn=embed_fn, embeddirs_fn=embeddirs_fn, netchunk=1024 * 64)
    K = orc.intrinsics(cfg)
    imgs = {}
    for tag, (Pc, Pf) in (("ref", orc.scene_params()), ("target", orc.teacher_params())):
        nc, nf = reference_networks(helpers, Pc), reference_networks(helpers, Pf)
        with torch.no_grad():
            rgb, disp, acc, extras = run_nerf.render(
                cfg["H"], cfg["W"], K, chunk=1024 * 32, rays=batch, ndc=cfg["ndc"], near=cfg["near"], far=cfg["far"],
                use_viewdirs=True, network_fn=nc, network_query_fn=qfn, N_samples=64, N_importance=n_importance,
                network_fine=nf, perturb=0.0, raw_noise_std=0.0, white_bkgd=cfg["white_bkgd"], lindisp=False)
        imgs[tag] = rgb.numpy()
        imgs[tag + "_acc"] = acc.numpy()
    mse = float(((imgs["ref"] - imgs["target"]) ** 2).mean())
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, rgb_ref=imgs["ref"], target=imgs["target"], acc_ref=imgs["ref_acc"],
                        rays_checksum=checksum(batch), target_psnr_db=np.float64(orc.psnr(mse)))
    print(f"{name}: {batch.shape[1]} rays, PSNR(reference image, teacher target) = {orc.psnr(mse):.2f} dB, "
          f"mean acc {imgs['ref_acc'].mean():.3f} -> {os.path.basename(path)} ({os.path.getsize(path)} B)")


def main_round2():
    """Round-2 fixtures: BASELINE.json configs[2] (fern, NDC) end to end through the reference's render(), and the
    PSNR-gate images of the lego-like and fern-like workloads."""
    run_nerf, helpers = load_reference()
    Pc, Pf = orc.scene_params()
    nets = (reference_networks(helpers, Pc), reference_networks(helpers, Pf))
    target = torch.tensor(np.random.RandomState(99).rand(N_RAYS, 3), dtype=torch.float32)
    cfg = orc.FERN
    rcfg = dict(H=cfg["H"], W=cfg["W"], K=orc.intrinsics(cfg), ndc=True, near=0.0, far=1.0)
    # 5. configs/fern.txt through render(): NDC rays, near=0/far=1, raw_noise_std=1, perturb=1, no white_bkgd, 64+128
    run_case("fern_ndc_train", run_nerf, helpers, orc.fern_batch(N_RAYS, seed=3), nets, Pc, Pf, target, seed=77,
             render=rcfg, kw=dict(perturb=1.0, raw_noise_std=1.0, white_bkgd=False, N_importance=128))
    # 6. the same through render() without NDC (lego): pins the rays=... branch of the boundary for configs[1]
    lcfg = orc.LEGO
    run_case("lego_render_train", run_nerf, helpers, orc.lego_batch(N_RAYS, seed=7), nets, Pc, Pf, target, seed=123,
             render=dict(H=lcfg["H"], W=lcfg["W"], K=orc.intrinsics(lcfg), ndc=False, near=2.0, far=6.0),
             kw=dict(perturb=1.0))
    gate_case("gate_lego", run_nerf, helpers, orc.lego_batch(1024, seed=31), orc.LEGO)
    gate_case("gate_fern", run_nerf, helpers, orc.fern_batch(1024, seed=32), orc.FERN)


def main_round3():
    """Round-3 fixtures: architectures outside the fused kernels, produced by the REAL reference through render() -- the command
    line's default (no --use_viewdirs: output_linear with 5 channels) and --netdepth 6 --netwidth 128 --multires 6
    --multires_views 2 with view directions.  Stored per case: the reference's outputs for 128 lego-like rays (24 + 40 samples,
    jitter, density noise, white background), its loss, a digest of every gradient, and per quantity the reference's own
    fp32-vs-fp64 distance (the oracle in fp64; the oracle is pinned bit-identical to the reference for these architectures)."""
    run_nerf, helpers = load_reference()
    lcfg = orc.LEGO
    for name in orc.DENSE_CASES:
        arch, Pc, Pf, batch, target, n_c, n_f = orc.dense_case(name)
        n = batch.shape[1]
        A = {k: arch[k] for k in ("D", "W", "input_ch", "input_ch_views", "output_ch", "skips", "use_viewdirs")}
        Ps = [Pc, Pf]
        nets = []
        for P in Ps:
            net = helpers.NeRF(**A)
            net.load_state_dict({k: v.clone() for k, v in P.items()})
            nets.append(net)
        i_embed = -1 if arch["multires"] < 0 else 0
        e_fn, _ = helpers.get_embedder(arch["multires"], i_embed)
        ed_fn = helpers.get_embedder(arch["multires_views"], i_embed)[0] if arch["use_viewdirs"] else None
        qfn = lambda inputs, viewdirs, network_fn: run_nerf.run_network(inputs, viewdirs, network_fn, embed_fn=e_fn, embeddirs_fn=ed_fn,
                                                                         netchunk=1024 * 64)
        torch.manual_seed(55)
        rgb, disp, acc, extras = run_nerf.render(lcfg["H"], lcfg["W"], orc.intrinsics(lcfg), chunk=1024, rays=batch, ndc=False, near=2.0,
                                                 far=6.0, use_viewdirs=arch["use_viewdirs"], network_fn=nets[0], network_query_fn=qfn,
                                                 N_samples=n_c, N_importance=n_f, network_fine=nets[1], perturb=1.0, white_bkgd=True,
                                                 raw_noise_std=0.5, retraw=True)
        loss = helpers.img2mse(rgb, target) + helpers.img2mse(extras["rgb0"], target)
        loss.backward()
        torch.manual_seed(55)       # the stream the reference consumed (run_nerf.py:371, :285, helpers:208, :285)
        rnd = dict(t_rand=torch.rand(n, n_c), noise_c=torch.randn(n, n_c), u=torch.rand(n, n_f), noise_f=torch.randn(n, n_c + n_f))
        flat = orc.assemble_render_rays(lcfg["H"], lcfg["W"], orc.intrinsics(lcfg), batch[0], batch[1], False, 2.0, 6.0)
        rr = flat if arch["use_viewdirs"] else flat[:, :8]
        P64 = [{k: v.double().requires_grad_(True) for k, v in P.items()} for P in Ps]
        o64 = orc.trace_rays(rr.double(), P64[0], P64[1], n_c, n_f, perturb=1.0, white_bkgd=True, raw_noise_std=0.5, retraw=True, arch=arch,
                             **{k: v.double() for k, v in rnd.items()})
        (orc.mse(o64["rgb_map"], target.double()) + orc.mse(o64["rgb0"], target.double())).backward()
        ref = dict(rgb_map=rgb, disp_map=disp, acc_map=acc, rgb0=extras["rgb0"], acc0=extras["acc0"], z_std=extras["z_std"], raw=extras["raw"])
        save = {"rays_checksum": np.float64(checksum(batch)), "params_checksum": np.float64(sum(checksum(v) for P in Ps for v in P.values())),
                "loss": np.float64(loss.item())}
        for k, v in ref.items():
            save[k] = v.detach().numpy()
            save[k + "/noise"] = (v.detach().double() - o64[k].detach()).abs().numpy()
        for tag, net, P in (("c", nets[0], P64[0]), ("f", nets[1], P64[1])):
            g32 = {k: p.grad for k, p in net.state_dict(keep_vars=True).items() if p.grad is not None}
            g64 = {k: P[k].grad for k in g32}
            for k, v in grad_digest(g32, g64).items():
                save[f"grad_{tag}/{k}"] = v
        path = os.path.join(HERE, name + ".npz")
        np.savez_compressed(path, **save)
        print(f"{name}: loss {loss.item():.6f}, mean acc {float(acc.detach().mean()):.3f}, rgb noise {save['rgb_map/noise'].max():.2e} -> "
              f"{os.path.basename(path)} ({os.path.getsize(path)} B)")


def main_round4():
    """Round-4 fixtures at BASELINE.json's OWN batch sizes, through the reference's render() (run_nerf.py:69-134 -> :54-66 ->
    :308-418): configs[1] = one 4096-ray lego training step, configs[2] = one 4096-ray fern / NDC training step (outputs per
    ray, loss, a digest of every gradient, the reference's fp32-vs-fp64 noise per quantity), and configs[3]'s 32,768-ray batch
    as ONE chunk, forward only (maps + loss) -- the shape npa.render(chunk=32768) renders in resident sub-chunks."""
    import time
    run_nerf, helpers = load_reference()
    Pc, Pf = orc.scene_params()
    nets = (reference_networks(helpers, Pc), reference_networks(helpers, Pf))
    n = 4096
    target = torch.tensor(np.random.RandomState(98).rand(n, 3), dtype=torch.float32)
    lcfg, fcfg = orc.LEGO, orc.FERN
    t0 = time.time()
    run_case("lego_cfg2_train", run_nerf, helpers, orc.lego_batch(n, seed=17), nets, Pc, Pf, target, seed=1234, raw_ray_stride=16,
             render=dict(H=lcfg["H"], W=lcfg["W"], K=orc.intrinsics(lcfg), ndc=False, near=2.0, far=6.0), kw=dict(perturb=1.0))
    print(f"  ({time.time() - t0:.0f} s)", flush=True)
    run_case("fern_cfg3_train", run_nerf, helpers, orc.fern_batch(n, seed=13), nets, Pc, Pf, target, seed=4321, raw_ray_stride=16,
             render=dict(H=fcfg["H"], W=fcfg["W"], K=orc.intrinsics(fcfg), ndc=True, near=0.0, far=1.0),
             kw=dict(perturb=1.0, raw_noise_std=1.0, white_bkgd=False, N_importance=128))
    print(f"  ({time.time() - t0:.0f} s)", flush=True)
    # configs[3]: 32,768 rays as one chunk, forward only
    n = 32768
    batch = orc.lego_batch(n, seed=19)
    target = torch.tensor(np.random.RandomState(97).rand(n, 3), dtype=torch.float32)
    embed_fn, _ = helpers.get_embedder(10, 0)
    embeddirs_fn, _ = helpers.get_embedder(4, 0)
    qfn = lambda inputs, viewdirs, network_fn: run_nerf.run_network(
        inputs, viewdirs, network_fn, embed_fn=embed_fn, embeddirs_fn=embeddirs_fn, netchunk=1024 * 64)
    torch.manual_seed(2024)
    with torch.no_grad():
        rgb, disp, acc, extras = run_nerf.render(800, 800, orc.intrinsics(dict(lcfg, H=800, W=800, focal=1111.0)), chunk=1024 * 32, rays=batch,
                                                 ndc=False, near=2.0, far=6.0, use_viewdirs=True, network_fn=nets[0], network_query_fn=qfn,
                                                 N_samples=64, N_importance=128, network_fine=nets[1], perturb=1.0, white_bkgd=True,
                                                 raw_noise_std=0.0, lindisp=False)
        loss = helpers.img2mse(rgb, target) + helpers.img2mse(extras["rgb0"], target)
    rec = dict(loss=np.float64(loss.item()), rays_checksum=checksum(batch), rgb_map=rgb.numpy(), disp_map=disp.numpy(), acc_map=acc.numpy(),
               rgb0=extras["rgb0"].numpy(), acc0=extras["acc0"].numpy(), z_std=extras["z_std"].numpy())
    path = os.path.join(HERE, "lego_cfg4_forward.npz")
    np.savez_compressed(path, **rec)
    print(f"lego_cfg4_forward: {n} rays in one chunk, loss {loss.item():.6f}, mean acc {float(acc.mean()):.3f} -> {os.path.basename(path)} "
          f"({os.path.getsize(path)} B)  ({time.time() - t0:.0f} s)")


if __name__ == "__main__":
    if "--round4" in sys.argv:
        main_round4()
    elif "--round3" in sys.argv:
        main_round3()
    elif "--round2" in sys.argv:
        main_round2()
    else:
        main()
