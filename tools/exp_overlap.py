"""Do the HBM-bound weight-gradient GEMM and the power-bound MFMA kernels run faster side by side (two streams) than back
to back?  Times each kernel alone, then pairs of them on two streams, and prints the concurrent wall time next to the sum.
(Timing only: the buffers are the right shapes, the concurrent pairs are not data-dependent.)"""
import os, sys
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
p3 = net.packed_params("bf16x3")
flat = net.flat_params()
N = 4096
s0 = torch.cuda.current_stream()
B = {}
for S in (64, 192):
    rays = wl.synthetic_rays(N, seed=1).to(dev)
    z = torch.sort(torch.rand(N, S, device=dev) * 4 + 2, -1)[0]
    b = dict(rays=rays, z=z, d_raw=torch.randn(N, S, 4, device=dev) * 1e-3, raw=torch.empty(N, S, 4, device=dev),
             act=torch.empty(hb.act_floats(N, S), device=dev), delta=torch.empty(L.nerf_delta_floats(N, S), device=dev),
             partial=torch.empty(L.nerf_wgrad_partial_floats(N, S), device=dev), grad=torch.empty(hb.N_PARAMS, device=dev), S=S)
    B[S] = b
    st = s0.cuda_stream
    assert L.nerf_field_fwd16r_bf16x3(p3.data_ptr(), rays.data_ptr(), 11, z.data_ptr(), N, S, b["raw"].data_ptr(), b["act"].data_ptr(), st) == 0
    assert L.nerf_field_dgrad3r_bf16x3(p3.data_ptr(), b["act"].data_ptr(), b["d_raw"].data_ptr(), N, S, b["delta"].data_ptr(), 1, st) == 0
torch.cuda.synchronize()


def fwd(S, save=True):
    b = B[S]
    return lambda st: L.nerf_field_fwd16r_bf16x3(p3.data_ptr(), b["rays"].data_ptr(), 11, b["z"].data_ptr(), N, S, b["raw"].data_ptr(),
                                                 b["act"].data_ptr() if save else None, st)


def dgrad(S):
    b = B[S]
    return lambda st: L.nerf_field_dgrad3r_bf16x3(p3.data_ptr(), b["act"].data_ptr(), b["d_raw"].data_ptr(), N, S, b["delta"].data_ptr(), 1, st)


def wgrad(S):
    b = B[S]
    return lambda st: L.nerf_field_wgrad_phase(b["act"].data_ptr(), b["delta"].data_ptr(), b["d_raw"].data_ptr(), N, S, b["partial"].data_ptr(),
                                               b["grad"].data_ptr(), 0, -1, 3, flat.data_ptr(), st)


def alone(fn, reps):
    for _ in range(3):
        assert fn(s0.cuda_stream) == 0
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn(s0.cuda_stream)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


import ctypes
HIP = ctypes.CDLL("libamdhip64.so")


def masked_stream(bits):
    """a stream restricted to the CUs whose bit is set (hipExtStreamCreateWithCUMask; 256 CUs = 8 words)"""
    words = (ctypes.c_uint32 * 8)(*[sum(1 << i for i in range(32) if bits[32 * w + i]) for w in range(8)])
    st = ctypes.c_void_p()
    rc = HIP.hipExtStreamCreateWithCUMask(ctypes.byref(st), 8, words)
    assert rc == 0, rc
    return torch.cuda.ExternalStream(st.value)


def together(fa, ra, fb, rb, sa=None, sb=None):
    sa = sa or torch.cuda.Stream()
    sb = sb or torch.cuda.Stream()
    best = None
    for _ in range(3):
        start = torch.cuda.Event(enable_timing=True)
        ea, eb = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        start.record(s0)
        sa.wait_event(start)
        sb.wait_event(start)
        # interleave the enqueues so that neither queue runs dry
        ia = ib = 0
        while ia < ra or ib < rb:
            if ia < ra and ia * rb <= ib * ra:
                fa(sa.cuda_stream); ia += 1
            else:
                fb(sb.cuda_stream); ib += 1
        ea.record(sa)
        eb.record(sb)
        torch.cuda.synchronize()
        t = max(start.elapsed_time(ea), start.elapsed_time(eb))
        best = t if best is None else min(best, t)
    return best


K = {"fwd_save_192": fwd(192), "fwd_save_64": fwd(64), "fwd_infer_192": fwd(192, False), "dgrad_192": dgrad(192), "dgrad_64": dgrad(64),
     "wgrad_192": wgrad(192), "wgrad_64": wgrad(64)}
T = {k: alone(f, 10) for k, f in K.items()}
for k, t in T.items():
    print(f"alone  {k:14s} {t:.4f} ms", flush=True)
for a, b in (("dgrad_192", "wgrad_64"), ("dgrad_64", "wgrad_192"), ("dgrad_192", "wgrad_192"), ("fwd_save_192", "wgrad_192"),
             ("fwd_save_192", "wgrad_64"), ("fwd_infer_192", "wgrad_192"), ("dgrad_192", "fwd_save_64")):
    # equal amounts of standalone time in both queues (~10 ms each)
    ra, rb = max(1, round(10.0 / T[a])), max(1, round(10.0 / T[b]))
    seq = ra * T[a] + rb * T[b]
    con = together(K[a], ra, K[b], rb)
    print(f"pair   {a:14s} x{ra:<3d} || {b:10s} x{rb:<3d}: back to back {seq:.3f} ms, two streams {con:.3f} ms  ({con / seq:.3f})", flush=True)

# ---- the same pairs with the CUs partitioned between the two streams
for name, mask_b in (("64 CUs: bits 192..255", [i >= 192 for i in range(256)]), ("64 CUs: every 4th bit", [i % 4 == 3 for i in range(256)]),
                     ("32 CUs: every 8th bit", [i % 8 == 7 for i in range(256)]), ("128 CUs: every 2nd bit", [i % 2 == 1 for i in range(256)])):
    sa, sb = masked_stream([not m for m in mask_b]), masked_stream(mask_b)
    print(f"-- MFMA kernel on the other CUs, GEMM on {name}", flush=True)
    for a, b in (("dgrad_192", "wgrad_192"), ("fwd_save_192", "wgrad_192")):
        ta = together(K[a], 5, K[b], 0, sa, sb) / 5
        tb = together(K[a], 0, K[b], 5, sa, sb) / 5
        ra, rb = max(1, round(10.0 / ta)), max(1, round(10.0 / tb))
        con = together(K[a], ra, K[b], rb, sa, sb)
        print(f"   {a} alone on its CUs {ta:.3f} ms, {b} alone on its CUs {tb:.3f} ms; x{ra} || x{rb}: {con:.3f} ms "
              f"= {con / (ra * T[a] + rb * T[b]):.3f} of back to back on the whole chip", flush=True)
