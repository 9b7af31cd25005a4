"""Why do the SAVING field kernels take 15-20 % more time per point in the fine launch (786k points) than in the coarse one
(262k)?  Same kernel, same work per point.  This times (a) coarse-sized launches back to back, each bracketed by events,
(b) one fine-sized launch, (c) a coarse-sized launch after an idle gap -- if (a) slows down from the second launch on, it is
the chip's power management (sustained load), not the launch length or the cache."""
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
nf = npa.NeRF(D=8, W=256, input_ch=63, output_ch=5, skips=[4], input_ch_views=27, use_viewdirs=True).to(dev)
nf.load_state_dict(Pf)
p3 = nf.packed_params("fp16x3")
s = torch.cuda.current_stream().cuda_stream
N = 4096
rays = wl.synthetic_rays(N, seed=1).to(dev)
bufs = {}
for S in (64, 192):
    bufs[S] = (torch.sort(torch.rand(N, S, device=dev) * 4 + 2, -1)[0], torch.empty(N, S, 4, device=dev), torch.empty(hb.act_floats(N, S), device=dev))
def fwd(S, save=True):
    z, raw, act = bufs[S]
    assert L.nerf_field_fwd_split(p3.data_ptr(), rays.data_ptr(), 11, z.data_ptr(), N, S, raw.data_ptr(), act.data_ptr() if save else None, 1, s) == 0
def timed(seq):
    evs = [torch.cuda.Event(enable_timing=True) for _ in range(len(seq) + 1)]
    evs[0].record()
    for i, (S, save) in enumerate(seq):
        fwd(S, save)
        evs[i + 1].record()
    torch.cuda.synchronize()
    return [evs[i].elapsed_time(evs[i + 1]) for i in range(len(seq))]
for _ in range(3): fwd(64); fwd(192)
torch.cuda.synchronize()
for save in (True, False):
    time.sleep(0.5)
    t = timed([(64, save)] * 12)
    print(f"save={save}: 12 coarse-sized launches back to back after 0.5 s idle (ms): " + " ".join(f"{x:.3f}" for x in t) + f"   ns/pt first {t[0] * 1e6 / (N * 64):.2f} last {t[-1] * 1e6 / (N * 64):.2f}", flush=True)
    time.sleep(0.5)
    t = timed([(192, save)] * 4)
    print(f"save={save}: 4 fine-sized launches back to back after 0.5 s idle (ms): " + " ".join(f"{x:.3f}" for x in t) + f"   ns/pt first {t[0] * 1e6 / (N * 192):.2f} last {t[-1] * 1e6 / (N * 192):.2f}", flush=True)
    time.sleep(0.5)
    t = timed([(64, save), (192, save), (64, save), (192, save), (64, save)])
    print(f"save={save}: coarse, fine, coarse, fine, coarse (ms): " + " ".join(f"{x:.3f}" for x in t), flush=True)
