"""Quick GPU timing probe (not the contract bench): HIP path vs the oracle run as eager PyTorch-ROCm."""
import os, sys, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import torch
import nerf_oracle as orc
import nerf_pytorch_amd as npa

dev = torch.device("cuda", 0)
N = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
print("device", torch.cuda.get_device_name(0), "cpus", os.cpu_count(), "torch", torch.__version__, "hip", torch.version.hip)
Pc, Pf = orc.scene_params()
kw = dict(D=8, W=256, input_ch=63, output_ch=5, skips=[4], input_ch_views=27, use_viewdirs=True)
nc, nf = npa.NeRF(**kw).to(dev), npa.NeRF(**kw).to(dev)
nc.load_state_dict(Pc); nf.load_state_dict(Pf)
rays = orc.synthetic_rays(N, seed=1).to(dev)
target = torch.rand(N, 3, device=dev)
opt = torch.optim.Adam(list(nc.parameters()) + list(nf.parameters()), lr=5e-4)

def timeit(fn, warm=2, reps=5):
    for _ in range(warm): fn()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t) / reps

def infer():
    with torch.no_grad():
        npa.render_rays(rays, nc, None, 64, N_importance=128, network_fine=nf, white_bkgd=True)
def train():
    opt.zero_grad()
    out = npa.render_rays(rays, nc, None, 64, N_importance=128, network_fine=nf, white_bkgd=True, perturb=1.0)
    loss = npa.img2mse(out["rgb_map"], target) + npa.img2mse(out["rgb0"], target)
    loss.backward(); opt.step()
res = {}
res["hip_infer_s"] = timeit(infer); res["hip_train_s"] = timeit(train)
# stage timings
packed = nf.packed_params()
z = torch.sort(torch.rand(N, 192, device=dev) * 4 + 2, -1)[0]
res["field_fwd_192_s"] = timeit(lambda: npa.hip_backend.field_fwd(packed, rays, z, False))
res["field_fwd_192_save_s"] = timeit(lambda: npa.hip_backend.field_fwd(packed, rays, z, True))
packed3 = nf.packed_params("bf16x3")
res["field_fwd3_192_s"] = timeit(lambda: npa.hip_backend.field_fwd(packed3, rays, z, False, precision="bf16x3"))
res["field_fwd3_192_save_s"] = timeit(lambda: npa.hip_backend.field_fwd(packed3, rays, z, True, precision="bf16x3"))
npa.set_precision("bf16x3"); res["hip_infer_bf16x3_s"] = timeit(infer); npa.set_precision("fp32")
raw, act = npa.hip_backend.field_fwd(packed, rays, z, True)
d_raw = torch.randn(N, 192, 4, device=dev); grad = torch.empty(595844, device=dev)
res["field_bwd_192_s"] = timeit(lambda: npa.hip_backend.field_bwd(packed, act, d_raw, grad, False))
del act
# eager reference (oracle ops on the GPU) -- the "PyTorch-ROCm eager" denominator
Pcg = {k: v.to(dev).requires_grad_(True) for k, v in Pc.items()}; Pfg = {k: v.to(dev).requires_grad_(True) for k, v in Pf.items()}
opt2 = torch.optim.Adam(list(Pcg.values()) + list(Pfg.values()), lr=5e-4)
torch.set_default_device(dev)
def eager_train():
    opt2.zero_grad()
    t_rand = torch.rand(N, 64); u = torch.rand(N, 128)
    out = orc.trace_rays(rays, Pcg, Pfg, 64, 128, perturb=1.0, white_bkgd=True, t_rand=t_rand, u=u)
    loss = orc.mse(out["rgb_map"], target) + orc.mse(out["rgb0"], target)
    loss.backward(); opt2.step()
def eager_infer():
    with torch.no_grad():
        orc.trace_rays(rays, Pcg, Pfg, 64, 128, perturb=0.0, white_bkgd=True)
if "--eager" in sys.argv:
    res["eager_infer_s"] = timeit(eager_infer, 1, 3); res["eager_train_s"] = timeit(eager_train, 1, 3)
for k in list(res):
    res[k.replace("_s", "_rays_per_s")] = N / res[k]
flop_fwd = 303.82e6 * N; flop_train = 893.19e6 * N
res["hip_infer_TFLOPs"] = flop_fwd / res["hip_infer_s"] / 1e12
res["hip_train_TFLOPs"] = flop_train / res["hip_train_s"] / 1e12
res["field_fwd_192_TFLOPs"] = 1186816 * N * 192 / res["field_fwd_192_s"] / 1e12
res["field_fwd3_192_TFLOPs_equiv"] = 1186816 * N * 192 / res["field_fwd3_192_s"] / 1e12
res["field_bwd_192_TFLOPs"] = 2 * (557696 + 593408) * N * 192 / res["field_bwd_192_s"] / 1e12
print(json.dumps(res, indent=1))
