"""A/B of the 16-point forward on the double-buffered weight stream (field_fwd16_kernel) and on the weight ring
(field_fwd16r_kernel): bit-identity of raw and of the whole save buffer, then timings (HIP events on the launch stream).
Usage: python tools/exp_ring.py [n_rays]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import workloads as wl
import nerf_pytorch_amd as npa

hb = npa.hip_backend
dev = torch.device("cuda", 0)
import ctypes
N = 4096
LR = None           # library whose ring forward is tested (default: the product library)
for a in sys.argv[1:]:
    if a.endswith(".so"):
        LR = ctypes.CDLL(os.path.join(ROOT, "nerf-pytorch_amd", a) if not os.path.isabs(a) else a)
        hb._declare(LR)
    else:
        N = int(a)
Pc, Pf = wl.scene_params()
kw = dict(D=8, W=256, input_ch=63, output_ch=5, skips=[4], input_ch_views=27, use_viewdirs=True)
nf = npa.NeRF(**kw).to(dev)
nf.load_state_dict(Pf)
p3 = nf.packed_params("bf16x3")
L = hb.lib()
s = torch.cuda.current_stream().cuda_stream


def run(kind, rays, z, raw, act):
    n, S = z.shape
    a = act.data_ptr() if act is not None else None
    if kind == "old":
        rc = L.nerf_field_fwd16_bf16x3(p3.data_ptr(), rays.data_ptr(), 11, z.data_ptr(), n, S, raw.data_ptr(), a, 1, s)
    else:
        rc = (LR or L).nerf_field_fwd16r_bf16x3(p3.data_ptr(), rays.data_ptr(), 11, z.data_ptr(), n, S, raw.data_ptr(), a, s)
    assert rc == 0, L.nerf_last_error()


ok = True
for n, S in ((37, 5), (129, 64), (1024, 192), (333, 77)):
    rays = wl.synthetic_rays(n, seed=3).to(dev)
    z = torch.sort(torch.rand(n, S, device=dev) * 4 + 2, -1)[0]
    outs = {}
    for kind in ("old", "ring"):
        for save in (False, True):
            raw = torch.zeros(n, S, 4, device=dev)
            act = torch.zeros(hb.act_floats(n, S), device=dev) if save else None
            run(kind, rays, z, raw, act)
            torch.cuda.synchronize()
            outs[(kind, save)] = (raw, act)
    for save in (False, True):
        r0, a0 = outs[("old", save)]
        r1, a1 = outs[("ring", save)]
        same_raw = torch.equal(r0.view(torch.int32), r1.view(torch.int32))
        same_act = True if not save else torch.equal(a0.view(torch.int32), a1.view(torch.int32))
        nz = 0 if not save else int((a1 != 0).sum())
        print(f"n={n} S={S} save={save}: raw identical {same_raw}, act identical {same_act} (nonzero act words {nz}), "
              f"max|raw| {r1.abs().max().item():.3f}", flush=True)
        ok &= same_raw and same_act
print("BIT-IDENTICAL" if ok else "MISMATCH", flush=True)

for S in (64, 192):
    rays = wl.synthetic_rays(N, seed=1).to(dev)
    z = torch.sort(torch.rand(N, S, device=dev) * 4 + 2, -1)[0]
    raw = torch.empty(N, S, 4, device=dev)
    act = torch.empty(hb.act_floats(N, S), device=dev)
    for save in (False, True):
        res = {}
        for rep in range(2):
            for kind in ("old", "ring"):
                for _ in range(3):
                    run(kind, rays, z, raw, act if save else None)
                e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(10):
                    run(kind, rays, z, raw, act if save else None)
                e1.record(); torch.cuda.synchronize()
                res.setdefault(kind, []).append(e0.elapsed_time(e1) / 10)
        print(f"N={N} S={S} save={save}: old {min(res['old']):.4f} ms  ring {min(res['ring']):.4f} ms  "
              f"speedup {min(res['old']) / min(res['ring']):.3f}  ({res})", flush=True)
