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


def oracle_fp64(rays, Pc, Pf, target, kw, rnd):
    """The same computation in float64 (oracle == reference bit for bit in fp32, see pin_against_reference):
    its distance from the fp32 reference is the reference's own rounding noise on these inputs."""
    P64c = {k: v.double().requires_grad_(True) for k, v in Pc.items()}
    P64f = {k: v.double().requires_grad_(True) for k, v in Pf.items()}
    n_f = kw["N_importance"]
    out = orc.trace_rays(rays.double(), P64c, P64f if n_f > 0 else None, 64, n_f, perturb=kw["perturb"],
                         lindisp=kw["lindisp"], white_bkgd=kw["white_bkgd"], raw_noise_std=kw["raw_noise_std"],
                         retraw=True, **{k: v.double() for k, v in rnd.items()})
    loss = orc.mse(out["rgb_map"], target.double())
    if "rgb0" in out:
        loss = loss + orc.mse(out["rgb0"], target.double())
    loss.backward()
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
    out = run_nerf.render_rays(rays, **kw)
    loss = helpers.img2mse(out["rgb_map"], target)
    if "rgb0" in out:
        loss = loss + helpers.img2mse(out["rgb0"], target)
    loss.backward()
    out64, g64c, g64f = oracle_fp64(rays, Pc, Pf, target, kw, draw_randoms(seed, kw))
    rec = {"loss": np.float64(loss.item()), "rays_checksum": checksum(rays),
           "params_checksum": checksum(torch.cat([v.reshape(-1) for v in Pc.values()])) +
           checksum(torch.cat([v.reshape(-1) for v in Pf.values()]))}
    for k, v in out.items():
        v = v.detach()
        d = (v.double() - out64[k].detach())
        both_nan = torch.isnan(v) & torch.isnan(out64[k].detach())
        rec["noise/" + k] = np.float64(d.abs().masked_fill(both_nan, 0.0).max())
        rec[k] = (v[:, ::8] if k == "raw" else v).numpy()
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


if __name__ == "__main__":
    main()
