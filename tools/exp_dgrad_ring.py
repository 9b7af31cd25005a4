"""A/B of the split-bf16 delta chain on the double-buffered weight stream (field_dgrad3_kernel) and on the weight ring
(field_dgrad3r_kernel): bit-identity of the delta buffer, then timings.  Usage: python tools/exp_dgrad_ring.py [lib.so]"""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import workloads as wl
import nerf_pytorch_amd as npa

hb = npa.hip_backend
dev = torch.device("cuda", 0)
L = hb.lib()
LR = L
for a in sys.argv[1:]:
    if a.endswith(".so"):
        LR = ctypes.CDLL(os.path.join(ROOT, "nerf-pytorch_amd", a) if not os.path.isabs(a) else a)
        hb._declare(LR)
Pc, Pf = wl.scene_params()
kw = dict(D=8, W=256, input_ch=63, output_ch=5, skips=[4], input_ch_views=27, use_viewdirs=True)
nf = npa.NeRF(**kw).to(dev)
nf.load_state_dict(Pf)
p3 = nf.packed_params("bf16x3")
s = torch.cuda.current_stream().cuda_stream
ok = True
for n, S in ((37, 5), (129, 64), (512, 192), (333, 77)):
    rays = wl.synthetic_rays(n, seed=3).to(dev)
    z = torch.sort(torch.rand(n, S, device=dev) * 4 + 2, -1)[0]
    d_raw = torch.randn(n, S, 4, device=dev)
    raw = torch.empty(n, S, 4, device=dev)
    act = torch.zeros(hb.act_floats(n, S), device=dev)
    assert L.nerf_field_fwd16r_bf16x3(p3.data_ptr(), rays.data_ptr(), 11, z.data_ptr(), n, S, raw.data_ptr(), act.data_ptr(), s) == 0
    for b16 in (1, 0):
        outs = []
        for fn in (L.nerf_field_dgrad_bf16x3, LR.nerf_field_dgrad3r_bf16x3):
            delta = torch.zeros(L.nerf_delta_floats(n, S), device=dev)
            assert fn(p3.data_ptr(), act.data_ptr(), d_raw.data_ptr(), n, S, delta.data_ptr(), b16, s) == 0
            outs.append(delta)
        torch.cuda.synchronize()
        same = torch.equal(outs[0].view(torch.int32), outs[1].view(torch.int32))
        print(f"n={n} S={S} bf16_out={b16}: delta identical {same} (nonzero words {int((outs[1] != 0).sum())})", flush=True)
        ok &= same
print("BIT-IDENTICAL" if ok else "MISMATCH", flush=True)
N = 4096
for S in (64, 192):
    rays = wl.synthetic_rays(N, seed=1).to(dev)
    z = torch.sort(torch.rand(N, S, device=dev) * 4 + 2, -1)[0]
    d_raw = torch.randn(N, S, 4, device=dev)
    raw = torch.empty(N, S, 4, device=dev)
    act = torch.empty(hb.act_floats(N, S), device=dev)
    delta = torch.empty(L.nerf_delta_floats(N, S), device=dev)
    L.nerf_field_fwd16r_bf16x3(p3.data_ptr(), rays.data_ptr(), 11, z.data_ptr(), N, S, raw.data_ptr(), act.data_ptr(), s)
    for b16 in (1, 0):
        res = {}
        for rep in range(2):
            for name, fn in (("stream", L.nerf_field_dgrad_bf16x3), ("ring", LR.nerf_field_dgrad3r_bf16x3)):
                for _ in range(3):
                    fn(p3.data_ptr(), act.data_ptr(), d_raw.data_ptr(), N, S, delta.data_ptr(), b16, s)
                e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(10):
                    fn(p3.data_ptr(), act.data_ptr(), d_raw.data_ptr(), N, S, delta.data_ptr(), b16, s)
                e1.record(); torch.cuda.synchronize()
                res.setdefault(name, []).append(e0.elapsed_time(e1) / 10)
        print(f"N={N} S={S} bf16_out={b16}: stream {min(res['stream']):.4f} ms  ring {min(res['ring']):.4f} ms  "
              f"speedup {min(res['stream']) / min(res['ring']):.3f}", flush=True)
