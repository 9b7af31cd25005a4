"""Times the ring forward (inference and bf16-saving, fine launch 4096 x 192 and coarse 4096 x 64) of every
nerf-pytorch_amd/libexp_*.so built by tools/ring_variants.py.  Results of most variants are wrong on purpose."""
import ctypes, glob, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import workloads as wl
import nerf_pytorch_amd as npa

hb = npa.hip_backend
dev = torch.device("cuda", 0)
N = 4096
Pc, Pf = wl.scene_params()
kw = dict(D=8, W=256, input_ch=63, output_ch=5, skips=[4], input_ch_views=27, use_viewdirs=True)
nf = npa.NeRF(**kw).to(dev)
nf.load_state_dict(Pf)
p3 = nf.packed_params("bf16x3")
s = torch.cuda.current_stream().cuda_stream
rays = wl.synthetic_rays(N, seed=1).to(dev)
zs = {S: torch.sort(torch.rand(N, S, device=dev) * 4 + 2, -1)[0] for S in (64, 192)}
raw = torch.empty(N, 192, 4, device=dev)
act = torch.empty(hb.act_floats(N, 192), device=dev)
only = sys.argv[1:]
libs = sorted(glob.glob(os.path.join(ROOT, "nerf-pytorch_amd", "libexp_*.so")))
d_raw = torch.randn(N, 192, 4, device=dev)
delta = torch.empty(hb.lib().nerf_delta_floats(N, 192), device=dev)
hb.lib().nerf_field_fwd16r_bf16x3(p3.data_ptr(), rays.data_ptr(), 11, zs[192].data_ptr(), N, 192, raw.data_ptr(), act.data_ptr(), s)
print("variant          infer192  save192  infer64  save64  dgrad192(bf16 out)  (ms, min of 2 x 10 launches)", flush=True)
for path in libs:
    name = os.path.basename(path)[7:-3]
    if only and name not in only:
        continue
    L = ctypes.CDLL(path)
    hb._declare(L)
    out = []
    for S in (192, 64):
        z = zs[S]
        for save in (False, True):
            a = act.data_ptr() if save else None
            best = 1e9
            for rep in range(2):
                for _ in range(2):
                    L.nerf_field_fwd16r_bf16x3(p3.data_ptr(), rays.data_ptr(), 11, z.data_ptr(), N, S, raw.data_ptr(), a, s)
                e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(10):
                    L.nerf_field_fwd16r_bf16x3(p3.data_ptr(), rays.data_ptr(), 11, z.data_ptr(), N, S, raw.data_ptr(), a, s)
                e1.record(); torch.cuda.synchronize()
                best = min(best, e0.elapsed_time(e1) / 10)
            out.append(best)
    best = 1e9
    assert L.nerf_field_fwd16r_bf16x3(p3.data_ptr(), rays.data_ptr(), 11, zs[192].data_ptr(), N, 192, raw.data_ptr(), act.data_ptr(), s) == 0
    assert L.nerf_field_dgrad3r_bf16x3(p3.data_ptr(), act.data_ptr(), d_raw.data_ptr(), N, 192, delta.data_ptr(), 1, s) == 0, L.nerf_last_error()
    for rep in range(2):
        for _ in range(2):
            L.nerf_field_dgrad3r_bf16x3(p3.data_ptr(), act.data_ptr(), d_raw.data_ptr(), N, 192, delta.data_ptr(), 1, s)
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            L.nerf_field_dgrad3r_bf16x3(p3.data_ptr(), act.data_ptr(), d_raw.data_ptr(), N, 192, delta.data_ptr(), 1, s)
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / 10)
    print(f"{name:16s} {out[0]:8.4f} {out[1]:8.4f} {out[2]:8.4f} {out[3]:8.4f} {best:8.4f}", flush=True)
