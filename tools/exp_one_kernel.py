"""Runs ONE field kernel of a library (rocprofv3 target): python tools/exp_one_kernel.py <lib.so|-> <fwd|fwdsave|dgrad> [reps]"""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import workloads as wl
import nerf_pytorch_amd as npa
hb = npa.hip_backend
dev = torch.device("cuda", 0)
L = hb.lib()
if sys.argv[1] != "-":
    L = ctypes.CDLL(os.path.join(ROOT, "nerf-pytorch_amd", sys.argv[1]))
    hb._declare(L)
what, reps = sys.argv[2], int(sys.argv[3]) if len(sys.argv) > 3 else 10
N, S = 4096, 192
Pc, Pf = wl.scene_params()
nf = npa.NeRF(D=8, W=256, input_ch=63, output_ch=5, skips=[4], input_ch_views=27, use_viewdirs=True).to(dev)
nf.load_state_dict(Pf)
p3 = nf.packed_params("bf16x3")
s = torch.cuda.current_stream().cuda_stream
rays = wl.synthetic_rays(N, seed=1).to(dev)
z = torch.sort(torch.rand(N, S, device=dev) * 4 + 2, -1)[0]
raw = torch.empty(N, S, 4, device=dev)
act = torch.empty(hb.act_floats(N, S), device=dev)
d_raw = torch.randn(N, S, 4, device=dev)
delta = torch.empty(L.nerf_delta_floats(N, S), device=dev)
assert L.nerf_field_fwd16r_bf16x3(p3.data_ptr(), rays.data_ptr(), 11, z.data_ptr(), N, S, raw.data_ptr(), act.data_ptr(), s) == 0
for _ in range(reps):
    if what == "fwd":
        L.nerf_field_fwd16r_bf16x3(p3.data_ptr(), rays.data_ptr(), 11, z.data_ptr(), N, S, raw.data_ptr(), None, s)
    elif what == "fwdsave":
        L.nerf_field_fwd16r_bf16x3(p3.data_ptr(), rays.data_ptr(), 11, z.data_ptr(), N, S, raw.data_ptr(), act.data_ptr(), s)
    else:
        assert L.nerf_field_dgrad3r_bf16x3(p3.data_ptr(), act.data_ptr(), d_raw.data_ptr(), N, S, delta.data_ptr(), 1, s) == 0
torch.cuda.synchronize()
