"""Board power and clocks while one kernel of the fp16x3 datapath runs back to back (rocm-smi sampled from a second thread): what clock
does the chip grant under each heavy kernel, and at what package power?  Usage: python tools/exp_power.py"""
import os, subprocess, sys, threading, time
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
net = npa.NeRF(**kw).to(dev)
net.load_state_dict(Pf)
p3 = net.packed_params("fp16x3")
flat = net.flat_params()
p8 = net.packed_params("fp16_fp8c")
N, S = 4096, 192
rays = wl.synthetic_rays(N, seed=1).to(dev)
z = torch.sort(torch.rand(N, S, device=dev) * 4 + 2, -1)[0]
d_raw = torch.randn(N, S, 4, device=dev) * 1e-3
raw = torch.empty(N, S, 4, device=dev)
act = torch.empty(hb.act_floats(N, S), device=dev)
delta = torch.empty(L.nerf_delta_floats(N, S), device=dev)
partial = torch.empty(L.nerf_wgrad_partial_floats(N, S), device=dev)
grad = torch.empty(hb.N_PARAMS, device=dev)
st = torch.cuda.current_stream().cuda_stream
K = {
    "idle": None,
    "forward (inference)": lambda: L.nerf_field_fwd_split(p3.data_ptr(), rays.data_ptr(), 11, z.data_ptr(), N, S, raw.data_ptr(), None, 1, st),
    "forward (inference, fp16 + fp8c)": lambda: L.nerf_field_fwd_split(p8.data_ptr(), rays.data_ptr(), 11, z.data_ptr(), N, S, raw.data_ptr(), None, 2, st),
    "forward (saving)": lambda: L.nerf_field_fwd_split(p3.data_ptr(), rays.data_ptr(), 11, z.data_ptr(), N, S, raw.data_ptr(), act.data_ptr(), 1, st),
    "dgrad": lambda: L.nerf_field_dgrad_split(p3.data_ptr(), act.data_ptr(), d_raw.data_ptr(), N, S, delta.data_ptr(), 1, st),
    "wgrad": lambda: L.nerf_field_wgrad_phase(act.data_ptr(), delta.data_ptr(), d_raw.data_ptr(), N, S, partial.data_ptr(), grad.data_ptr(), 0, -1, 3,
                                              flat.data_ptr(), st),
}
K["forward (saving)"](); K["dgrad"](); torch.cuda.synchronize()


def smi():
    out = subprocess.run(["rocm-smi", "--showpower", "--showclocks", "--showmaxpower", "--csv"], capture_output=True, text=True)
    return out.stdout.strip().splitlines()


print("\n".join(smi()[:3]), flush=True)
for name, fn in K.items():
    samples, stop = [], False

    def sampler():
        while not stop:
            samples.append(smi())
            time.sleep(0.05)
    th = threading.Thread(target=sampler)
    t0 = time.time()
    th.start()
    n = 0
    while time.time() - t0 < 4.0:
        if fn is None:
            time.sleep(0.1)
        else:
            for _ in range(50):
                fn()
            torch.cuda.synchronize()
            n += 50
    stop = True
    th.join()
    dt = time.time() - t0
    rows = [s[1] for s in samples[len(samples) // 2:] if len(s) > 1]          # second half: settled
    print(f"== {name}: {n} launches in {dt:.2f} s ({1e3 * dt / max(n, 1):.3f} ms each)")
    for r in rows[-3:]:
        print("   ", r)
