"""Inference forward, three-term fp16 vs the reduced class (fp16 main + fp8 corrections): ms per launch for the coarse- and
fine-sized launches of a 4096-ray batch, and whole no_grad render() calls."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import workloads as wl
import nerf_pytorch_amd as npa
hb = npa.hip_backend
dev = torch.device("cuda", 0)
L = hb.lib()
Pc, Pf = wl.scene_params()
kw = dict(D=8, W=256, input_ch=63, output_ch=5, skips=[4], input_ch_views=27, use_viewdirs=True)
nc, nf = npa.NeRF(**kw).to(dev), npa.NeRF(**kw).to(dev)
nc.load_state_dict(Pc); nf.load_state_dict(Pf)
p3, p8 = nf.packed_params("fp16x3"), nf.packed_params("fp16_fp8c")
s = torch.cuda.current_stream().cuda_stream
N = 4096
rays = wl.synthetic_rays(N, seed=1).to(dev)
for S in (64, 192):
    z = torch.sort(torch.rand(N, S, device=dev) * 4 + 2, -1)[0]
    raw = torch.empty(N, S, 4, device=dev)
    for name, p, split in (("fp16x3", p3, 1), ("fp16+fp8c", p8, 2), ("fp16x3", p3, 1), ("fp16+fp8c", p8, 2)):
        fn = lambda: L.nerf_field_fwd_split(p.data_ptr(), rays.data_ptr(), 11, z.data_ptr(), N, S, raw.data_ptr(), None, split, s)
        for _ in range(5): assert fn() == 0
        torch.cuda.synchronize(); t = time.perf_counter()
        for _ in range(40): fn()
        torch.cuda.synchronize(); t = (time.perf_counter() - t) / 40
        print(f"S={S:3d} {name:10s}: {t * 1e3:.3f} ms per launch ({N * S / t / 1e6:.1f} M points/s)", flush=True)
batch = wl.lego_batch(N, seed=3).to(dev)
cfg = wl.LEGO
args = dict(chunk=1 << 15, ndc=False, near=2.0, far=6.0, use_viewdirs=True, network_fn=nc, network_query_fn=None, N_samples=64, N_importance=128,
            network_fine=nf, perturb=0.0, white_bkgd=True, raw_noise_std=0.0)
for prec in ("fp16x3", "fp16_fp8c", "fp16x3", "fp16_fp8c"):
    npa.set_precision(prec)
    with torch.no_grad():
        for _ in range(3): npa.render(cfg["H"], cfg["W"], wl.intrinsics(cfg), rays=batch, **args)
        torch.cuda.synchronize(); t = time.perf_counter()
        for _ in range(20): npa.render(cfg["H"], cfg["W"], wl.intrinsics(cfg), rays=batch, **args)
        torch.cuda.synchronize(); t = (time.perf_counter() - t) / 20
    print(f"render() no_grad, {prec}: {t * 1e3:.3f} ms per 4096-ray batch = {N / t / 1e6:.3f} M rays/s", flush=True)
npa.set_precision("fp32")
