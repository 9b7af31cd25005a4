import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import torch
import nerf_oracle as orc
import nerf_pytorch_amd as npa
hb = npa.hip_backend
if len(sys.argv) > 1 and sys.argv[1] != "-":
    npa.build.LIB_PATH = os.path.join(ROOT, "nerf-pytorch_amd", sys.argv[1]); hb._LIB = None
dev = torch.device("cuda", 0); N = 4096
Pc, Pf = orc.scene_params()
kw = dict(D=8, W=256, input_ch=63, output_ch=5, skips=[4], input_ch_views=27, use_viewdirs=True)
nf = npa.NeRF(**kw).to(dev); nf.load_state_dict(Pf)
rays = orc.synthetic_rays(N, seed=1).to(dev)
z = torch.sort(torch.rand(N, 192, device=dev) * 4 + 2, -1)[0]
def timeit(fn, warm=3, reps=10):
    for _ in range(warm): fn()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t) / reps * 1e3
p3 = nf.packed_params("bf16x3")
act = torch.empty(hb.act_floats(N, 192), device=dev); raw = torch.empty(N, 192, 4, device=dev)
L = hb.lib(); s = torch.cuda.current_stream().cuda_stream
f = lambda a: L.nerf_field_fwd_bf16x3(p3.data_ptr(), rays.data_ptr(), 11, z.data_ptr(), N, 192, raw.data_ptr(), a, s)
print(sys.argv[1:], "fwd3 nosave %.3f ms | save %.3f ms" % (timeit(lambda: f(None)), timeit(lambda: f(act.data_ptr()))), flush=True)
f16 = lambda a, bf: L.nerf_field_fwd16_bf16x3(p3.data_ptr(), rays.data_ptr(), 11, z.data_ptr(), N, 192, raw.data_ptr(), a, bf, s)
print("   fwd16 nosave %.3f ms | save %.3f ms | save bf16 %.3f ms" % (timeit(lambda: f16(None, 0)), timeit(lambda: f16(act.data_ptr(), 0)), timeit(lambda: f16(act.data_ptr(), 1))), flush=True)
if "--bwd" in sys.argv:
    d_raw = torch.randn(N, 192, 4, device=dev); delta = torch.empty(L.nerf_delta_floats(N, 192), device=dev)
    g = lambda b: L.nerf_field_dgrad_bf16x3(p3.data_ptr(), act.data_ptr(), d_raw.data_ptr(), N, 192, delta.data_ptr(), b, s)
    f(act.data_ptr()); print("   dgrad3 %.3f ms | bf16 deltas %.3f ms" % (timeit(lambda: g(0)), timeit(lambda: g(1))), flush=True)
    partial = torch.empty(L.nerf_wgrad_partial_floats(N, 192), device=dev); grad = torch.empty(595844, device=dev)
    w = lambda ph: L.nerf_field_wgrad_phase(act.data_ptr(), delta.data_ptr(), d_raw.data_ptr(), N, 192, partial.data_ptr(), grad.data_ptr(), 0, 3, ph, nf.flat_params().data_ptr(), s)
    print("   wgrad3 %.3f ms  (+reduce %.3f ms)" % (timeit(lambda: w(1)), timeit(lambda: w(4))), flush=True)

if "--mixed" in sys.argv:
    fm = lambda: L.nerf_field_fwd_mixed(p3.data_ptr(), rays.data_ptr(), 11, z.data_ptr(), N, 192, raw.data_ptr(), act.data_ptr(), s)
    print("   mixed: fwd<save bf16> %.3f ms" % timeit(fm), flush=True)
    gm = lambda: L.nerf_field_dgrad_mixed(p3.data_ptr(), act.data_ptr(), d_raw.data_ptr(), N, 192, delta.data_ptr(), 0, s)
    print("   mixed: dgrad %.3f ms" % timeit(gm), flush=True)
    wm = lambda ph: L.nerf_field_wgrad_phase(act.data_ptr(), delta.data_ptr(), d_raw.data_ptr(), N, 192, partial.data_ptr(), grad.data_ptr(), 0, 2, ph, nf.flat_params().data_ptr(), s)
    print("   mixed: wgrad1 %.3f ms" % timeit(lambda: wm(1)), flush=True)
