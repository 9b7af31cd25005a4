"""Contract benchmark: rays/s of the volumetric-rendering training step on MI355X.

    python bench.py --gpus N --steps K --warmup W        (N > 1: launched through torch.distributed.run)

One "step" = one pass of the hot path over one batch of synthetic rays, exactly what train() does per
iteration around render() (run_nerf.py:760-776): render (coarse 64 + fine 128 samples, perturb=1,
white_bkgd, both networks) -> MSE(rgb)+MSE(rgb0) -> backward -> [RCCL all-reduce of the two flat
gradient buckets] -> Adam.  Workload = BASELINE.json configs[1]: lego-like, N_rand = 4096 rays per GPU,
64+128 samples (configs[3] = the same per-GPU work on 8 GPUs, global batch 32768 -> "weak" scaling).
Inputs are resident in HBM before the timed region.  Prints ONE JSON line (rank 0).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for _p in (ROOT, os.path.join(ROOT, "oracle")):
    if _p not in sys.path:
        sys.path.insert(0, _p)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

N_RAND = 4096
N_SAMPLES, N_IMPORTANCE = 64, 128
FLOP_FWD_PER_RAY = 2 * 593408 * (N_SAMPLES + N_SAMPLES + N_IMPORTANCE)               # 303.82 MFLOP
FLOP_TRAIN_PER_RAY = 2 * (593408 + 557696 + 593408) * (N_SAMPLES + N_SAMPLES + N_IMPORTANCE)  # 893.19 MFLOP
PEAK_FP32_MFMA_TFLOPS = 157.3       # MI355X_MICROARCH.md: f32-input MFMA = vector rate; exact-fp32 datapath
PEAK_BF16_MFMA_TFLOPS = 2500.0      # dense bf16 MFMA; the bf16x3 datapath issues 3 MFMA FLOPs per algorithmic FLOP


def cpu_baseline(n_rays=512):
    """The oracle (bit-identical restatement of the reference, CPU, fp32) timed on this box's host cores on a
    bounded sample of the same workload: one training step of n_rays rays x (64+128) samples."""
    import nerf_oracle as orc
    Pc, Pf = orc.scene_params()
    Pc = {k: v.requires_grad_(True) for k, v in Pc.items()}
    Pf = {k: v.requires_grad_(True) for k, v in Pf.items()}
    opt = torch.optim.Adam(list(Pc.values()) + list(Pf.values()), lr=5e-4, betas=(0.9, 0.999))
    rays = orc.synthetic_rays(n_rays, seed=1)
    target = torch.rand(n_rays, 3)

    def step():
        opt.zero_grad()
        t_rand, u = torch.rand(n_rays, N_SAMPLES), torch.rand(n_rays, N_IMPORTANCE)
        out = orc.trace_rays(rays, Pc, Pf, N_SAMPLES, N_IMPORTANCE, perturb=1.0, white_bkgd=True, t_rand=t_rand, u=u)
        loss = orc.mse(out["rgb_map"], target) + orc.mse(out["rgb0"], target)
        loss.backward()
        opt.step()
    step()
    t0 = time.perf_counter()
    reps = 0
    while reps < 2 or (time.perf_counter() - t0 < 10.0 and reps < 20):
        step()
        reps += 1
    dt = (time.perf_counter() - t0) / reps
    return {"value": n_rays / dt, "unit": "rays/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": f"{reps} training steps of {n_rays} rays x (64+128) samples (oracle = bit-identical "
                      f"restatement of the reference, torch CPU fp32, {torch.get_num_threads()} threads of "
                      f"{os.cpu_count()} host CPUs)"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--mode", choices=["train", "infer"], default="train")
    ap.add_argument("--rays", type=int, default=N_RAND, help="rays per GPU per step")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--precision", choices=["fp32", "bf16x3"], default=os.environ.get("NERF_BENCH_PRECISION", "fp32"),
                    help="field datapath: exact fp32 MFMA or split-bf16 (3 bf16 MFMAs per product, fp32 accumulate)")
    args = ap.parse_args()

    import nerf_oracle as orc
    import nerf_pytorch_amd as npa
    from nerf_pytorch_amd import parallel

    npa.set_precision(args.precision)
    rank, world, dev = parallel.init_distributed()
    assert world == args.gpus or world == 1, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    assert torch.cuda.is_available(), "bench.py needs an MI355X (the render hot path has no CPU fallback)"
    n = args.rays

    Pc, Pf = orc.scene_params()
    kw = dict(D=8, W=256, input_ch=63, output_ch=5, skips=[4], input_ch_views=27, use_viewdirs=True)
    net_c, net_f = npa.NeRF(**kw).to(dev), npa.NeRF(**kw).to(dev)
    net_c.load_state_dict(Pc)
    net_f.load_state_dict(Pf)
    parallel.broadcast_parameters([net_c, net_f])
    opt = torch.optim.Adam(list(net_c.parameters()) + list(net_f.parameters()), lr=5e-4, betas=(0.9, 0.999))

    # synthetic data, resident in HBM: a pool of ray batches (rank-dependent seeds) + targets
    pool = 8
    rays = [orc.synthetic_rays(n, seed=1000 * rank + i).to(dev) for i in range(pool)]
    gen = torch.Generator().manual_seed(77 + rank)
    targets = [torch.rand(n, 3, generator=gen).to(dev) for _ in range(pool)]
    render_kw = dict(N_samples=N_SAMPLES, N_importance=N_IMPORTANCE, network_fine=net_f, white_bkgd=True,
                     raw_noise_std=0., retraw=True)

    def train_step(i):
        opt.zero_grad()
        out = npa.render_rays(rays[i % pool], net_c, None, perturb=1.0, **render_kw)
        t = targets[i % pool]
        loss = npa.img2mse(out["rgb_map"], t) + npa.img2mse(out["rgb0"], t)
        loss.backward()
        parallel.allreduce_gradients([net_c, net_f])
        opt.step()

    def infer_step(i):
        with torch.no_grad():
            npa.render_rays(rays[i % pool], net_c, None, perturb=0., **render_kw)

    step = train_step if args.mode == "train" else infer_step

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for i in range(args.warmup):
        step(i)
    timer = npa.hip_backend.KernelTimer()
    npa.hip_backend.TIMER = timer
    barrier()
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(i)
    barrier()
    elapsed = time.perf_counter() - t0
    npa.hip_backend.TIMER = None
    kern = timer.summary()
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # secondary number: inference rays/s on the same batch shape (not the headline)
    other = None
    if args.mode == "train":
        for i in range(2):
            infer_step(i)
        barrier()
        t1 = time.perf_counter()
        for i in range(max(5, args.steps // 2)):
            infer_step(i)
        barrier()
        other = n * world * max(5, args.steps // 2) / (time.perf_counter() - t1)

    if rank == 0:
        total_rays = n * world * args.steps
        value = total_rays / elapsed
        flop_per_ray = FLOP_TRAIN_PER_RAY if args.mode == "train" else FLOP_FWD_PER_RAY
        dom_name, dom = max(kern.items(), key=lambda kv: kv[1]["ms"]) if kern else (None, None)
        kernels = {k: {"launches": v["launches"], "avg_ms": v["ms"] / v["launches"],
                       "tflops": v["flops"] / (v["ms"] * 1e-3) / 1e12} for k, v in kern.items()}
        roofline = None
        if dom is not None:
            ach = dom["flops"] / (dom["ms"] * 1e-3) / 1e12
            if args.precision == "fp32":
                peak, issued = PEAK_FP32_MFMA_TFLOPS, 1.0
            else:       # every algorithmic FLOP is issued three times on the bf16 pipe
                peak, issued = PEAK_BF16_MFMA_TFLOPS, 3.0
            roofline = {"bound": "mfma", "kernel": dom_name, "achieved": ach * issued, "peak": peak,
                        "unit": "TFLOP/s", "frac": ach * issued / peak, "traffic": None,
                        "algorithmic_tflops": ach, "mfma_flops_per_algorithmic_flop": issued,
                        "avg_launch_ms": dom["ms"] / dom["launches"],
                        "whole_step_frac": value * flop_per_ray * issued / world / 1e12 / peak}
        line = {
            "metric": "rays/sec (coarse+fine, 64+128 samples), training step" if args.mode == "train"
                      else "rays/sec (coarse+fine, 64+128 samples), inference",
            "value": value, "unit": "rays/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32" if args.precision == "fp32" else "bf16x3 (split-bf16 MFMA products, f32 accumulate/activations/gradients)",
            "data": "synthetic",
            "config": {"workload": f"lego-like 400x400, N_rand={n} rays/GPU x (64 coarse + 128 fine) samples, "
                                   "two 8x256 networks, perturb=1, white_bkgd; step = render + MSE + backward"
                                   + (" + RCCL grad all-reduce" if world > 1 else "") + " + Adam"
                                   if args.mode == "train" else
                                   f"lego-like, {n} rays/GPU x (64+128) samples, no_grad render",
                       "global_batch_rays": n * world, "parallelism": f"ray-shard dp{world}"},
            "roofline": roofline, "kernels": kernels,
        }
        if other is not None:
            line["inference_rays_per_s"] = other
        if not args.no_cpu_baseline and world == 1:
            line["cpu_baseline"] = cpu_baseline()
        print(json.dumps(line))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
