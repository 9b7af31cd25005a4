"""Times the split-bf16 weight-gradient GEMM of a library variant (tools/build_variants.py) and prints a digest of the
gradient it produces, so that variants can be compared with the product library.  usage: exp_wgrad.py [libname|-]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import workloads as wl
import nerf_pytorch_amd as npa
hb = npa.hip_backend
name = sys.argv[1] if len(sys.argv) > 1 else "-"
B16 = int("--bf16" in sys.argv)         # bf16 operand storage (wgrad1_kernel) instead of fp32 (wgrad3_256_kernel)
if name != "-":
    npa.build.LIB_PATH = os.path.join(ROOT, "nerf-pytorch_amd", name); hb._LIB = None
dev = torch.device("cuda", 0); N = 4096
Pc, Pf = wl.scene_params()
kw = dict(D=8, W=256, input_ch=63, output_ch=5, skips=[4], input_ch_views=27, use_viewdirs=True)
nf = npa.NeRF(**kw).to(dev); nf.load_state_dict(Pf)
rays = wl.synthetic_rays(N, seed=1).to(dev)
torch.manual_seed(0)
def timeit(fn, warm=3, reps=10):
    for _ in range(warm): fn()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t) / reps * 1e3
p3 = nf.packed_params("bf16x3")
L = hb.lib(); s = torch.cuda.current_stream().cuda_stream
out = []
for S in (64, 192):
    z = torch.sort(torch.rand(N, S, device=dev) * 4 + 2, -1)[0]
    act = torch.empty(hb.act_floats(N, S), device=dev); raw = torch.empty(N, S, 4, device=dev)
    L.nerf_field_fwd16_bf16x3(p3.data_ptr(), rays.data_ptr(), 11, z.data_ptr(), N, S, raw.data_ptr(), act.data_ptr(), B16, s)
    d_raw = torch.randn(N, S, 4, device=dev); delta = torch.empty(L.nerf_delta_floats(N, S), device=dev)
    L.nerf_field_dgrad_bf16x3(p3.data_ptr(), act.data_ptr(), d_raw.data_ptr(), N, S, delta.data_ptr(), B16, s)
    partial = torch.empty(L.nerf_wgrad_partial_floats(N, S), device=dev); grad = torch.zeros(595844, device=dev)
    w = lambda ph: L.nerf_field_wgrad_phase(act.data_ptr(), delta.data_ptr(), d_raw.data_ptr(), N, S, partial.data_ptr(), grad.data_ptr(), 0, 4 if B16 else 3, ph, nf.flat_params().data_ptr(), s)
    t1, t4 = timeit(lambda: w(3)), timeit(lambda: w(4))
    w(7); torch.cuda.synchronize()
    out.append("S=%d wgrad %.3f ms reduce %.3f ms digest %.9e %.9e" % (S, t1, t4, grad.double().abs().sum().item(), grad.double().pow(2).sum().item()))
print(name, "bf16 operands" if B16 else "fp32 operands", "|", " | ".join(out), flush=True)
