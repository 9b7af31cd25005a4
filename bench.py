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


PEAK_HBM_GBS = 8000.0               # HBM3E spec (≈6.3 TB/s achievable, MI355X_MICROARCH.md)


def kernel_table(kern, precision):
    """per timed kernel: average launch time, algorithmic TFLOP/s and GB/s, and its fraction of both roofs"""
    issued = 1.0 if precision == "fp32" else 3.0        # bf16x3 issues every algorithmic FLOP three times on the bf16 pipe
    peak = PEAK_FP32_MFMA_TFLOPS if precision == "fp32" else PEAK_BF16_MFMA_TFLOPS
    out = {}
    for k, v in kern.items():
        sec = v["ms"] * 1e-3
        fp32_kernel = precision == "fp32" or k.startswith(("wgrad_kernel", "wgrad_reduce"))
        k_issued, k_peak = (1.0, PEAK_FP32_MFMA_TFLOPS) if fp32_kernel else (issued, peak)
        if k.startswith(("field_dgrad3_kernel<mixed>", "wgrad1_kernel")):      # single bf16 MFMA per product
            k_issued = 1.0
        tfl = v["flops"] / sec / 1e12
        gbs = v["bytes"] / sec / 1e9
        out[k] = {"launches": v["launches"], "avg_ms": v["ms"] / v["launches"], "total_ms": v["ms"],
                  "algorithmic_tflops": tfl, "mfma_frac": tfl * k_issued / k_peak, "mfma_peak_tflops": k_peak,
                  "mfma_flops_per_algorithmic_flop": k_issued,
                  "algorithmic_GBps": gbs, "hbm_frac": gbs / PEAK_HBM_GBS}
    return out


def pmc_traffic(kernel_name, precision):
    """HBM bytes per launch of `kernel_name` from the committed rocprofv3 PMC summary of this same command
    (separate --pmc passes; FETCH_SIZE doubled on gfx950 as MI355X_MICROARCH.md prescribes).  None if absent."""
    path = os.path.join(ROOT, "profiles", "r01_bf16x3_pmc_summary.csv" if precision == "bf16x3" else "r01_pmc_summary.csv")
    key = kernel_name.split("<")[0].split("(")[0]
    fetch = write = None
    try:
        for line in open(path):
            if line.startswith("#") or "," not in line:
                continue
            kn, cn, _, val = line.rstrip().rsplit(",", 3)
            base = kn.replace("void ", "").replace("nerf::", "")
            tmpl = base.split("<")[1].split(">")[0] if "<" in base else ""
            if base.split("<")[0].split("(")[0] != key:
                continue
            if key in ("field_fwd3_kernel", "field_fwd16_kernel") and (tmpl in ("1", "true")) != ("<save>" in kernel_name):
                continue
            if key == "field_dgrad3_kernel" and (tmpl in ("1", "true")) != ("<mixed>" in kernel_name):
                continue
            if True:
                if cn == "FETCH_SIZE":
                    fetch = float(val)
                elif cn == "WRITE_SIZE":
                    write = float(val)
    except OSError:
        return None, None
    if fetch is None or write is None:
        return None, None
    return (2.0 * fetch + write) * 1024.0, os.path.relpath(path, ROOT)


def roofline_of(table):
    """roofline object of the dominant kernel (largest share of the timed region), bound = the roof it sits closer to"""
    if not table:
        return None
    name, k = max(table.items(), key=lambda kv: kv[1]["total_ms"])
    if k["hbm_frac"] > k["mfma_frac"]:
        return {"bound": "hbm", "kernel": name, "achieved": k["algorithmic_GBps"], "peak": PEAK_HBM_GBS, "unit": "GB/s",
                "frac": k["hbm_frac"], "traffic": None, "avg_launch_ms": k["avg_ms"],
                "also": {"mfma_frac": k["mfma_frac"], "algorithmic_tflops": k["algorithmic_tflops"]}}
    return {"bound": "mfma", "kernel": name, "achieved": k["algorithmic_tflops"] * k["mfma_flops_per_algorithmic_flop"],
            "peak": k["mfma_peak_tflops"], "unit": "TFLOP/s", "frac": k["mfma_frac"], "traffic": None,
            "algorithmic_tflops": k["algorithmic_tflops"],
            "mfma_flops_per_algorithmic_flop": k["mfma_flops_per_algorithmic_flop"], "avg_launch_ms": k["avg_ms"],
            "also": {"hbm_frac": k["hbm_frac"], "algorithmic_GBps": k["algorithmic_GBps"]}}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--mode", choices=["train", "infer"], default="train")
    ap.add_argument("--rays", type=int, default=N_RAND, help="rays per GPU per step")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--precision", choices=["fp32", "bf16x3", "mixed"], default=os.environ.get("NERF_BENCH_PRECISION", "bf16x3"),
                    help="headline field datapath: split-bf16 (3 bf16 MFMAs per product, fp32 accumulate; PSNR delta vs the "
                         "reference 2e-5 dB, tests/test_gpu_parity.py) or exact fp32 MFMA (the parity anchor). The other "
                         "datapath is measured too (fewer steps) and reported in the same JSON line, and so is the "
                         "mixed-precision training option (bf16x3 forward + bf16 backward; 'mixed'), which is never the "
                         "default headline because its gradients are bf16-rounded.")
    ap.add_argument("--single-datapath", action="store_true", help="skip the secondary datapath measurement")
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
    # torch.optim.Adam semantics (run_nerf.py:207), fused over the two flat parameter vectors (state_dict compatible)
    opt = npa.FlatAdam(list(net_c.parameters()) + list(net_f.parameters()), lr=5e-4, betas=(0.9, 0.999))

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

    def measure(precision, steps, warmup):
        npa.set_precision(precision)
        for i in range(warmup):
            step(i)
        timer = npa.hip_backend.KernelTimer()
        npa.hip_backend.TIMER = timer
        barrier()
        t0 = time.perf_counter()
        for i in range(steps):
            step(i)
        barrier()
        el = time.perf_counter() - t0
        npa.hip_backend.TIMER = None
        kern = timer.summary()
        if world > 1:
            t = torch.tensor([el], dtype=torch.float64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            el = float(t.item())
        return el, kern

    elapsed, kern = measure(args.precision, args.steps, args.warmup)

    # secondary numbers (not the headline): inference on the same batch shape, and the other datapath
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
    second = second_mixed = None
    if not args.single_datapath:
        p2 = "bf16x3" if args.precision == "fp32" else "fp32"
        k2 = max(4, args.steps // 4)
        el2, kern2 = measure(p2, k2, 2)
        tab2 = kernel_table(kern2, p2)
        second = {"dtype": "f32" if p2 == "fp32" else "bf16x3", "value": n * world * k2 / el2, "unit": "rays/s",
                  "steps": k2, "ms_per_step": 1e3 * el2 / k2, "roofline": roofline_of(tab2),
                  "kernels": {k: {"avg_ms": v["avg_ms"], "mfma_frac": v["mfma_frac"], "hbm_frac": v["hbm_frac"]}
                              for k, v in tab2.items()}}
        if args.mode == "train" and args.precision != "mixed":
            k3 = max(4, args.steps // 2)
            el3, kern3 = measure("mixed", k3, 2)
            tab3 = kernel_table(kern3, "mixed")
            second_mixed = {"dtype": "bf16x3 forward (outputs identical to the headline datapath) + bf16 backward (saved activations / "
                                     "deltas rounded to bf16, one bf16 MFMA per product, f32 accumulate)",
                            "value": n * world * k3 / el3, "unit": "rays/s", "steps": k3, "ms_per_step": 1e3 * el3 / k3,
                            "kernels": {k: {"avg_ms": v["avg_ms"], "mfma_frac": v["mfma_frac"], "hbm_frac": v["hbm_frac"]}
                                        for k, v in tab3.items()}}
        npa.set_precision(args.precision)

    if rank == 0:
        total_rays = n * world * args.steps
        value = total_rays / elapsed
        flop_per_ray = FLOP_TRAIN_PER_RAY if args.mode == "train" else FLOP_FWD_PER_RAY
        kernels = kernel_table(kern, args.precision)
        roofline = roofline_of(kernels)
        if roofline is not None:
            tr, src = pmc_traffic(roofline["kernel"], args.precision)
            if tr is not None:
                k = kernels[roofline["kernel"]]
                roofline["traffic"] = tr
                roofline["traffic_note"] = (f"bytes per launch (mean over coarse+fine launches) from {src}: 2*FETCH_SIZE + WRITE_SIZE; "
                                            f"algorithmic bytes per launch here: {k['algorithmic_GBps'] * 1e9 * k['avg_ms'] * 1e-3:.4g}")
            issued = 1.0 if args.precision == "fp32" else 3.0
            peak = PEAK_FP32_MFMA_TFLOPS if args.precision == "fp32" else PEAK_BF16_MFMA_TFLOPS
            roofline["whole_step_mfma_frac"] = value * flop_per_ray * issued / world / 1e12 / peak
        line = {
            "metric": "rays/sec (coarse+fine, 64+128 samples), training step" if args.mode == "train"
                      else "rays/sec (coarse+fine, 64+128 samples), inference",
            "value": value, "unit": "rays/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None,
            "dtype": {"fp32": "f32", "bf16x3": "bf16x3 (split-bf16 MFMA products, f32 accumulate/activations/gradients)",
                      "mixed": "bf16x3 forward + bf16 backward (mixed-precision training option)"}[args.precision],
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
        if second is not None:
            line["other_datapath"] = second
        if second_mixed is not None:
            line["mixed_precision_training"] = second_mixed
        if not args.no_cpu_baseline and world == 1:
            line["cpu_baseline"] = cpu_baseline()
        print(json.dumps(line))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
