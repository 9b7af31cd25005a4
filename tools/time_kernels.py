"""Per-launch time of the heavy kernels at BASELINE's launch sizes (4096 rays x 64 / 192 samples), HIP events around batches of
back-to-back launches on torch's current stream.  For A/B timing of two builds on ONE box (boxes differ by 2-4 %):

    NERF_HIP_LIB=/path/to/other/libnerf_hip.so python tools/time_kernels.py [--precision fp16x3] [--reps 20] [--rounds 3]

prints one JSON line: {kernel: {"coarse_ms": .., "fine_ms": ..}}.  tools/ab.sh alternates two libraries."""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import nerf_pytorch_amd as npa  # noqa: E402
import workloads as wl  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--precision", default="fp16x3")
    ap.add_argument("--reps", type=int, default=20)
    ap.add_argument("--rounds", type=int, default=3)
    ap.add_argument("--rays", type=int, default=4096)
    ap.add_argument("--only", default="")
    args = ap.parse_args()
    hb = npa.hip_backend
    L = hb.lib()
    dev = torch.device("cuda", 0)
    prec = args.precision
    Pc, Pf = wl.scene_params()
    net = npa.NeRF(D=8, W=256, input_ch=63, output_ch=5, skips=[4], input_ch_views=27, use_viewdirs=True).to(dev)
    net.load_state_dict(Pf)
    packed = net.packed_params(prec)
    n = args.rays
    rays = wl.synthetic_rays(n, seed=1).to(dev)
    out = {}

    def timed(fn):
        best = None
        for _ in range(args.rounds):
            for _ in range(3):
                fn()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(args.reps):
                fn()
            e1.record()
            torch.cuda.synchronize()
            t = e0.elapsed_time(e1) / args.reps
            best = t if best is None else min(best, t)
        return round(best, 4)

    for tag, S in (("coarse", 64), ("fine", 192)):
        z = torch.sort(torch.rand(n, S, device=dev) * 4 + 2, -1)[0]
        d_raw = torch.randn(n, S, 4, device=dev) * 1e-4
        raw = torch.empty(n, S, 4, device=dev)
        act = torch.empty(max(hb.act_floats(n, S), hb.act_floats(n, S, prec)), device=dev)
        delta = torch.empty(max(L.nerf_delta_floats(n, S), hb.delta_floats(n, S, prec)), device=dev)
        partial = torch.empty(L.nerf_wgrad_partial_floats(n, S), device=dev)
        grad = torch.empty(hb.N_PARAMS, device=dev)
        s = torch.cuda.current_stream().cuda_stream
        split = {"bf16x3": 0, "fp16x3": 1, "fp16x3w": 5}.get(prec)       # (5: two-word saves; the inference forward is split 1's)
        flat = net.flat_params()
        if split is None:
            fwd_i = lambda: L.nerf_field_fwd(packed.data_ptr(), rays.data_ptr(), 11, z.data_ptr(), n, S, raw.data_ptr(), None, s)
            fwd_s = lambda: L.nerf_field_fwd(packed.data_ptr(), rays.data_ptr(), 11, z.data_ptr(), n, S, raw.data_ptr(), act.data_ptr(), s)
            dgrad = lambda: L.nerf_field_dgrad(packed.data_ptr(), act.data_ptr(), d_raw.data_ptr(), n, S, delta.data_ptr(), s)
        else:
            fwd_i = lambda: L.nerf_field_fwd_split(packed.data_ptr(), rays.data_ptr(), 11, z.data_ptr(), n, S, raw.data_ptr(), None, split, s)
            fwd_s = lambda: L.nerf_field_fwd_split(packed.data_ptr(), rays.data_ptr(), 11, z.data_ptr(), n, S, raw.data_ptr(), act.data_ptr(), split, s)
            dgrad = lambda: L.nerf_field_dgrad_split(packed.data_ptr(), act.data_ptr(), d_raw.data_ptr(), n, S, delta.data_ptr(), split, s)
        wargs = (act.data_ptr(), delta.data_ptr(), d_raw.data_ptr(), n, S, partial.data_ptr(), grad.data_ptr(), 0, -1)
        wgemm = lambda: L.nerf_field_wgrad_phase(*wargs, 3, flat.data_ptr(), s)
        wred = lambda: L.nerf_field_wgrad_phase(*wargs, 4, flat.data_ptr(), s)
        assert fwd_s() == 0 and dgrad() == 0 and wgemm() == 0 and wred() == 0, L.nerf_last_error()
        for name, fn in (("fwd_infer", fwd_i), ("fwd_save", fwd_s), ("dgrad", dgrad), ("wgrad_gemm", wgemm), ("wgrad_reduce", wred)):
            if args.only and name not in args.only.split(","):
                continue
            out.setdefault(name, {})[tag + "_ms"] = timed(fn)
    print(json.dumps({"lib": os.environ.get("NERF_HIP_LIB", "in-tree"), "precision": prec, "kernels": out}))


if __name__ == "__main__":
    main()
