import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import workloads as wl
import nerf_pytorch_amd as npa
hb = npa.hip_backend
dev = torch.device("cuda", 0); N = 4096
Pc, Pf = wl.scene_params()
kw = dict(D=8, W=256, input_ch=63, output_ch=5, skips=[4], input_ch_views=27, use_viewdirs=True)
nf = npa.NeRF(**kw).to(dev); nf.load_state_dict(Pf)
rays = wl.synthetic_rays(N, seed=1).to(dev)
z = torch.sort(torch.rand(N, 192, device=dev) * 4 + 2, -1)[0]
p3 = nf.packed_params("bf16x3")
act = torch.empty(hb.act_floats(N, 192), device=dev); raw = torch.empty(N, 192, 4, device=dev)
L = hb.lib(); s = torch.cuda.current_stream().cuda_stream
f16 = lambda a, bf: L.nerf_field_fwd16_bf16x3(p3.data_ptr(), rays.data_ptr(), 11, z.data_ptr(), N, 192, raw.data_ptr(), a, bf, s)
d_raw = torch.randn(N, 192, 4, device=dev); delta = torch.empty(L.nerf_delta_floats(N, 192), device=dev)
g = lambda: L.nerf_field_dgrad_bf16x3(p3.data_ptr(), act.data_ptr(), d_raw.data_ptr(), N, 192, delta.data_ptr(), 0, s)
for _ in range(6):
    f16(None, 0); f16(act.data_ptr(), 0); f16(act.data_ptr(), 1); g()
torch.cuda.synchronize()
