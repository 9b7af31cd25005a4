"""One training launch of 28,672 rays x 192 samples (5.5 M points, 2.8 GB per 256-wide region: past 2^31 bytes) under a raised
NERF_SAVE_BUDGET against the same rays as two half launches: outputs bit-identical, gradients equal to 6e-7 (round 5; the default
budget's largest launch, 21,504 rays, is tests/test_gpu_golden_cfg.py::test_largest_single_launch_matches_two_half_launches)."""
import sys, os, torch
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "oracle"))
import nerf_pytorch_amd as npa, nerf_oracle as orc, workloads as wl
hb = npa.hip_backend
dev = torch.device("cuda", 0)
hb.SAVE_BUDGET_BYTES = 80 << 30
kw = dict(D=8, W=256, input_ch=63, output_ch=5, skips=[4], input_ch_views=27, use_viewdirs=True)
Pc, Pf = wl.scene_params()
nc, nf = npa.NeRF(**kw).to(dev), npa.NeRF(**kw).to(dev)
nc.load_state_dict(Pc); nf.load_state_dict(Pf)
npa.set_precision("fp16x3")
n = 28672
print("max rays per launch", hb.max_saved_rays(64, 128, "fp16x3"), "region bytes", n * 192 * 512)
rays = orc.synthetic_rays(n, seed=5).to(dev)
g = torch.Generator().manual_seed(9)
rnd = {"t_rand": torch.rand(n, 64, generator=g).to(dev), "u": torch.rand(n, 128, generator=g).to(dev)}
target = torch.rand(n, 3, generator=g).to(dev)
kwr = dict(N_samples=64, N_importance=128, network_fine=nf, white_bkgd=True, perturb=1.0, retraw=False)
rm = sys.modules[npa.parallel.__name__.rsplit(".", 1)[0] + ".render"]
def run(lo, hi):
    for m in (nc, nf): m.zero_grad()
    out = npa.render_rays(rays[lo:hi], nc, None, randoms={k: v[lo:hi] for k, v in rnd.items()}, **kwr)
    plan = rm.LAST_BACKWARD_PLAN
    (((out["rgb_map"] - target[lo:hi]) ** 2).sum() + ((out["rgb0"] - target[lo:hi]) ** 2).sum()).backward()
    torch.cuda.synchronize()
    return out["rgb_map"].detach().clone(), torch.cat([nc.last_flat_grad, nf.last_flat_grad]).double(), plan
rgb, g_all, plan = run(0, n)
print(plan)
r1, g1, _ = run(0, n // 2); r2, g2, _ = run(n // 2, n)
print("outputs equal", torch.equal(rgb, torch.cat([r1, r2])), "grad rel", float((g_all - (g1 + g2)).norm() / (g1 + g2).norm()), "finite", bool(torch.isfinite(g_all).all()))
