"""Round 6: what the save path of the saving forward and of the delta chain costs, ingredient by ingredient.
One library per process (NERF_HIP_LIB selects a `tools/build_variant.sh <name> -DNERF_ABL_SAVE=k` build); each kernel runs back to
back for --seconds on the fine launch (4096 rays x 192 samples) while board power and the granted shader clock are sampled.
Prints one JSON line; tools/exp_save_ablation.sh alternates the builds on ONE box and writes profiles/r06_save_ablation.txt."""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import nerf_pytorch_amd as npa  # noqa: E402
import workloads as wl  # noqa: E402
from bench_support import PowerSampler  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=2.0)
    ap.add_argument("--kernels", default="fwd_infer,fwd_save,dgrad")
    ap.add_argument("--samples", type=int, default=192)
    args = ap.parse_args()
    hb = npa.hip_backend
    L = hb.lib()
    dev = torch.device("cuda", 0)
    Pc, Pf = wl.scene_params()
    net = npa.NeRF(D=8, W=256, input_ch=63, output_ch=5, skips=[4], input_ch_views=27, use_viewdirs=True).to(dev)
    net.load_state_dict(Pf)
    packed = net.packed_params("fp16x3")
    n, S = 4096, args.samples
    rays = wl.synthetic_rays(n, seed=1).to(dev)
    z = torch.sort(torch.rand(n, S, device=dev) * 4 + 2, -1)[0]
    d_raw = torch.randn(n, S, 4, device=dev) * 1e-4
    raw = torch.empty(n, S, 4, device=dev)
    act = torch.empty(hb.act_floats(n, S), device=dev)
    delta = torch.empty(L.nerf_delta_floats(n, S), device=dev)
    s = torch.cuda.current_stream().cuda_stream
    K = {
        "fwd_infer": lambda: L.nerf_field_fwd_split(packed.data_ptr(), rays.data_ptr(), 11, z.data_ptr(), n, S, raw.data_ptr(), None, 1, s),
        "fwd_save": lambda: L.nerf_field_fwd_split(packed.data_ptr(), rays.data_ptr(), 11, z.data_ptr(), n, S, raw.data_ptr(), act.data_ptr(), 1, s),
        "dgrad": lambda: L.nerf_field_dgrad_split(packed.data_ptr(), act.data_ptr(), d_raw.data_ptr(), n, S, delta.data_ptr(), 1, s),
    }
    assert K["fwd_save"]() == 0 and K["dgrad"]() == 0, L.nerf_last_error()
    torch.cuda.synchronize()
    out = {}
    for name in args.kernels.split(","):
        fn = K[name]
        for _ in range(20):
            fn()
        torch.cuda.synchronize()
        with PowerSampler(0.02) as ps:
            t0 = time.perf_counter()
            ms, launches = 0.0, 0
            while time.perf_counter() - t0 < args.seconds:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(50):
                    fn()
                e1.record()
                torch.cuda.synchronize()
                ms += e0.elapsed_time(e1)
                launches += 50
            t1 = time.perf_counter()
        pw = ps.summary(t0 + 0.5 * (t1 - t0), t1)          # second half: settled
        out[name] = {"ms": round(ms / launches, 4), "w": pw.get("mean_w") and round(pw["mean_w"]), "mhz": pw.get("sclk_mhz_mean") and round(pw["sclk_mhz_mean"])}
    print(json.dumps({"lib": os.path.basename(os.environ.get("NERF_HIP_LIB", "in-tree")), "points": n * S, "kernels": out}))


if __name__ == "__main__":
    main()
