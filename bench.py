"""Contract benchmark: rays/s of the volumetric-rendering hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--config lego|fern] [--mode train|infer|render_only] [--strong]

N > 1: one process per GPU.  Launched by the driver through ``python -m torch.distributed.run ... bench.py --gpus N``
(RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* in the environment) -- or directly: ``python bench.py --gpus N`` re-executes
itself under torch.distributed.run (127.0.0.1 rendezvous) and fails loudly when the box has fewer than N GPUs.

One "step" = one pass of the hot path over one batch of synthetic rays, written exactly as the reference's train() writes
it around render() (run_nerf.py:760-784):

    rgb, disp, acc, extras = render(H, W, K, chunk=args.chunk, rays=batch_rays, verbose=False, retraw=True, **render_kwargs_train)
    optimizer.zero_grad(); loss = img2mse(rgb, target_s) + img2mse(extras['rgb0'], target_s); loss.backward()
    [RCCL all-reduce of the two flat gradient buckets]; optimizer.step()

Workloads (BASELINE.json `configs`; synthetic, seeded: workloads.py):
    --config lego  (default; configs[1], and configs[3] = the same per-GPU work on 8 GPUs): 400x400, N_rand = 4096
                   rays per GPU, 64 coarse + 128 fine samples, two 8x256 networks, perturb=1, white_bkgd, no NDC
    --config fern  (configs[2]): 504x378, focal 407.5, forward-facing rays through the NDC warp, near=0 / far=1,
                   raw_noise_std=1, perturb=1, no white_bkgd, N_rand = 4096, 64 + 128
    --mode render_only (configs[4]): one step = one 800x800 frame (640,000 rays in chunks of 32,768, no_grad,
                   perturb=0) of a pose_spherical spiral; frames are dealt round-robin over the ranks, no collective
    --strong       (configs[3] as strong scaling): the GLOBAL batch is 32,768 rays, random draws made once for the
                   full batch from a shared seed and sliced per rank
The default command (``python bench.py`` = 1 GPU, lego training step) also runs SHORT legs of the other single-GPU
configurations -- fern training step, render_only frames, the 32,768-ray batch -- and reports them under `configs`, each
with its own north-star gate (PSNR delta of our image vs the reference's image on that configuration's fixture).
Inputs are resident in HBM before the timed region.  `value` is timed with the per-kernel event timer OFF; the per-kernel
table (`kernels`, `roofline`) comes from a separate pass over the same steps.  Prints ONE JSON line (rank 0).

`roofline`: `achieved` = ALGORITHMIC FLOP of the reference's layer stack (1,186,816 per point and forward
evaluation: run_nerf_helpers.py:96-119) processed by one launch of the dominant kernel / its average launch time, `frac` =
achieved / the dense MFMA peak of the arithmetic type the datapath computes in (bf16: 2.5 PFLOP/s; f32: 157.3 TFLOP/s).
The split datapaths issue three 16-bit MFMAs per product and execute one folded layer less than the reference; the
fraction of the MFMA pipe their instructions occupy is reported next to it as `mfma_busy_frac` and is NOT the roofline
fraction.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

N_RAND = 4096
N_SAMPLES, N_IMPORTANCE = 64, 128
POINTS_PER_RAY = N_SAMPLES + N_SAMPLES + N_IMPORTANCE
MAC_FWD, MAC_DGRAD, MAC_WGRAD, MAC_FOLD = 593408, 557696, 593408, 65536        # per sample point
FLOP_FWD_PER_RAY = 2 * MAC_FWD * POINTS_PER_RAY                                  # 303.82 MFLOP
FLOP_TRAIN_PER_RAY = 2 * (MAC_FWD + MAC_DGRAD + MAC_WGRAD) * POINTS_PER_RAY      # 893.19 MFLOP
PEAK_FP32_MFMA_TFLOPS = 157.3       # MI355X_MICROARCH.md: f32-input MFMA = vector rate; exact-fp32 datapath
PEAK_BF16_MFMA_TFLOPS = 2500.0      # dense 16-bit MFMA (bf16 and fp16 run at the same rate)
PEAK_HBM_GBS = 8000.0               # HBM3E spec (~6.3 TB/s achievable, MI355X_MICROARCH.md)
DTYPE_NAME = {"fp32": "f32", "bf16x3": "bf16x3 (split-bf16 MFMA products W_hi x_hi + W_hi x_lo + W_lo x_hi in the forward and the delta chain, f32 "
                                        "accumulate / activations / deltas / gradients; the operands of the weight-gradient GEMM are stored as "
                                        "bf16 (8 significant bits) and multiplied exactly, f32 accumulate)",
              "fp16_fp8c": "fp16 main term + fp8 (e4m3) correction terms per product, inference only (gradients: fp16x3)",
              "fp16x3": "fp16x3 (split-fp16 MFMA products W_hi x_hi + W_hi x_lo + W_lo x_hi, hi = fp16(v), lo = fp16(v - hi): ~2^-22 per product, in the "
                        "forward and the delta chain, f32 accumulate / activations / deltas / gradients; the operands of the weight-gradient GEMM are the "
                        "fp16 hi words (11 significant bits), multiplied exactly, f32 accumulate; deltas scaled by a power of two per launch)",
              "fp16x3w": "fp16x3w (fp16x3 whose weight-gradient GEMM contracts TWO-WORD operands: d_hi X_hi + d_hi X_lo + d_lo X_hi on the hi and lo "
                         "words the forward and the delta chain save -- the forward's product class in all three heavy kernels; forward values "
                         "bit-identical to fp16x3's)"}


from bench_support import (CONVERGING_PAIRS, LEGO_TXT, PowerSampler, convergence_table, cpu_baseline,  # noqa: E402,F401
                           gradient_vs_fp64, rocm_eager_baseline)


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--config", choices=["lego", "fern"], default="lego")
    ap.add_argument("--mode", choices=["train", "infer", "render_only"], default="train")
    ap.add_argument("--strong", action="store_true", help="strong scaling: global batch of 32768 rays split over the ranks")
    ap.add_argument("--rays", type=int, default=N_RAND, help="rays per GPU per step (weak scaling)")
    ap.add_argument("--frame", type=int, default=800, help="render_only: frame side in pixels")
    ap.add_argument("--chunk", type=int, default=1024 * 32)
    ap.add_argument("--precision", choices=["fp32", "fp16x3", "bf16x3", "fp16_fp8c", "fp16x3w"], default=os.environ.get("NERF_BENCH_PRECISION", "fp16x3"),
                    help="headline field datapath.  fp16x3 (default) = three-term split with fp16 parts (3 MFMAs per product, ~2^-22 per product, "
                         "fp32 accumulate / activations / gradients, 11-bit operands for the weight-gradient GEMM); bf16x3 = the same with bf16 parts "
                         "(2^-17, 8-bit operands: rounds 1-3); both admitted by the north-star PSNR criterion, which this run re-measures and "
                         "prints (`precision_gate`); fp32 = exact fp32 MFMA (the parity anchor).  The other datapaths are measured "
                         "in the same run (`other_datapath`, `bf16x3_datapath`).")
    ap.add_argument("--single-datapath", action="store_true", help="skip the secondary datapath measurements")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-eager-baseline", action="store_true")
    ap.add_argument("--cpu-full", action="store_true", help="only the CPU baseline on the metric's own shape: one warm-up + one timed 4096-ray training step "
                                                             "of the oracle port on all host threads (~2 min); prints its record (commit it as profiles/r06_cpu_full_shape.json)")
    ap.add_argument("--no-gate", action="store_true")
    ap.add_argument("--no-configs", action="store_true", help="skip the short legs of the other single-GPU configurations")
    ap.add_argument("--sustained-s", type=float, default=6.0,
                    help="default run only: seconds of back-to-back training steps for `sustained` (rate over the whole interval and per "
                         "second, board power and shader clock sampled meanwhile); 0 = skip")
    ap.add_argument("--no-training-gate", action="store_true",
                    help="skip `precision_gate.training` (default run: the converging teacher / student pair of --long, 300 Adam steps of 1024 "
                         "rays on fp32, fp32 one ulp away and the headline datapath; ~25 s)")
    ap.add_argument("--long", action="store_true",
                    help="also fit a student to a teacher scene in every datapath (3 seeds x --long-steps Adam steps of 1024 rays, same "
                         "initialisation, batches and draws) and report the held-out PSNR per datapath as mean +- spread "
                         "(`precision_gate.training`): the training-equivalence evidence, ~1-2 min")
    ap.add_argument("--long-steps", type=int, default=2000)
    ap.add_argument("--long-twins", type=int, default=4, help="--long: perturbed starts of the fp32 datapath and of the fp16x3 datapath per seed (families of 1 + N runs each)")
    ap.add_argument("--backend", choices=["nccl", "gloo"], default=None, help="process-group backend (default: nccl = RCCL)")
    ap.add_argument("--bf16x3-leg", action="store_true", help="also measure the training step on the bf16 split (rounds 1-3's headline datapath)")
    ap.add_argument("--force-group", action="store_true",
                    help="build the process group even for ONE rank (NERF_FORCE_PROCESS_GROUP=1): a 1-GPU box then executes the RCCL branch as "
                         "written (ProcessGroupNCCL, async all-reduce work objects, broadcast, all-gather) and the line carries a multi_gpu block")
    ap.add_argument("--dry-run", action="store_true",
                    help="launcher / process-group plumbing only (no GPU work): used by the CPU test of the N > 1 launch path")
    return ap.parse_args(argv)


def relaunch_if_needed(args):
    """`python bench.py --gpus N` with no torchrun environment: become N ranks."""
    if args.gpus <= 1 or "WORLD_SIZE" in os.environ:
        return
    import socket
    if not args.dry_run and os.environ.get("NERF_ALLOW_SHARED_GPU") != "1":
        import torch
        have = torch.cuda.device_count()
        if have < args.gpus:
            sys.stderr.write(f"bench.py: --gpus {args.gpus} requested but this box exposes {have} GPU(s); refusing to report a "
                             f"{args.gpus}-GPU number from fewer devices\n")
            sys.exit(2)
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    os.execve(sys.executable, cmd, env)


# --------------------------------------------------------------------------------------------- per-kernel table / roofline
def _kernel_class(name):
    """(MFMAs issued per executed product, executed / algorithmic MAC ratio, arithmetic peak) of a timed kernel"""
    fwd = name.startswith(("field_fwd3_kernel", "field_fwd16_kernel", "field_fwd16r_kernel", "render_infer_kernel"))
    if name.startswith(("field_fwd_kernel", "field_dgrad_kernel", "wgrad256_kernel", "wgrad_kernel")):
        return 1.0, 1.0, PEAK_FP32_MFMA_TFLOPS                                  # exact-fp32 datapath
    if fwd:
        return (2.0 if "fp8c" in name else 3.0), (MAC_FWD - MAC_FOLD) / MAC_FWD, PEAK_BF16_MFMA_TFLOPS
    if name.startswith(("field_dgrad3_kernel", "field_dgrad3r_kernel")):
        return 3.0, (MAC_DGRAD - MAC_FOLD) / MAC_DGRAD, PEAK_BF16_MFMA_TFLOPS
    if name.startswith("wgrad1_kernel"):
        return (3.0 if "3 terms" in name else 1.0), (MAC_WGRAD - MAC_FOLD) / MAC_WGRAD, PEAK_BF16_MFMA_TFLOPS
    if name.startswith("wgrad3_256_kernel"):
        return 3.0, (MAC_WGRAD - MAC_FOLD) / MAC_WGRAD, PEAK_BF16_MFMA_TFLOPS
    return 1.0, 1.0, PEAK_FP32_MFMA_TFLOPS


def kernel_table(kern):
    """per timed kernel: average launch time; ALGORITHMIC TFLOP/s (the reference's layer stack) and its fraction of the
    arithmetic type's dense MFMA peak (`algorithmic_frac`, the roofline fraction of SURVEY 8d); the fraction of the MFMA
    pipe its issued instructions occupy (`mfma_busy_frac`: executed work x MFMAs per product); algorithmic GB/s vs HBM"""
    out = {}
    for k, v in kern.items():
        sec = v["ms"] * 1e-3
        issued, exec_ratio, peak = _kernel_class(k)
        exec_tfl = v["flops"] / sec / 1e12                      # hip_backend passes the EXECUTED flops of the launch
        alg_tfl = exec_tfl / exec_ratio
        gbs = v["bytes"] / sec / 1e9
        out[k] = {"launches": v["launches"], "avg_ms": v["ms"] / v["launches"], "total_ms": v["ms"],
                  "algorithmic_tflops": alg_tfl, "algorithmic_frac": alg_tfl / peak, "mfma_peak_tflops": peak,
                  "mfma_busy_frac": exec_tfl * issued / peak, "mfma_per_product": issued, "executed_over_algorithmic": exec_ratio,
                  "algorithmic_GBps": gbs, "hbm_frac": gbs / PEAK_HBM_GBS}
    return out


def _brief(tab):
    return {k: {"avg_ms": v["avg_ms"], "algorithmic_frac": v["algorithmic_frac"], "mfma_busy_frac": v["mfma_busy_frac"],
                "hbm_frac": v["hbm_frac"]} for k, v in tab.items()}


def _profile_row_matches(timer_name, prof_name):
    """does a kernel row of a rocprofv3 summary (`nerf::field_fwd16r_kernel<2, nerf::SplitF16>`) name the kernel instantiation the
    in-process timer calls `timer_name` (`field_fwd16r_kernel<fp16, save>`)?"""
    base, _, targs = prof_name.replace("void ", "").replace("nerf::", "").strip('"').partition("<")
    targs = targs[:targs.rfind(">")] if ">" in targs else targs        # (drop the argument list of a full signature)
    targs = [a.strip() for a in targs.split(",")] if targs else []
    key = timer_name.split("<")[0].split("(")[0]
    if base.split("(")[0] != key:
        return False
    f16 = "fp16" in timer_name
    if any(a in ("SplitF16", "SplitBF16") for a in targs) and (("SplitF16" in targs) != f16):
        return False
    first = {"false": "0", "true": "1", "": "0"}.get(targs[0] if targs else "", targs[0] if targs else "")
    if key in ("field_fwd3_kernel", "field_fwd16_kernel", "field_fwd16r_kernel", "field_fwd_kernel"):
        want = "3" if "hi+lo" in timer_name else "2" if ("<save bf16>" in timer_name or (f16 and "save" in timer_name)) else ("1" if "<save" in timer_name else "0")
        return first == want
    two = "hi+lo" in timer_name or "3 terms" in timer_name          # the two-word forms: <SP, true> / <SP, 3>
    if key in ("field_dgrad3_kernel", "field_dgrad3r_kernel", "wgrad1_kernel") and len(targs) > 1:
        return (targs[1] in ("true", "3")) == two
    return True


def pmc_traffic(kernel_name, precision):
    """HBM bytes per launch of `kernel_name` from the committed rocprofv3 PMC summary of this same command (separate --pmc
    passes; FETCH_SIZE doubled on gfx950 as MI355X_MICROARCH.md prescribes).  PMC counters cannot be read from inside the
    process, so this is the profile of the same command committed under profiles/ (None if absent)."""
    tag = {"bf16x3": "bf16x3_", "fp16x3": "fp16x3_", "fp16x3w": "fp16x3w_"}.get(precision, "")
    for rnd in ("r06", "r05", "r04", "r03", "r02", "r01"):
        path = os.path.join(ROOT, "profiles", f"{rnd}_{tag}pmc_summary.csv")
        if os.path.exists(path):
            break
    else:
        return None, None
    fetch = write = None
    for line in open(path):
        if line.startswith("#") or "," not in line:
            continue
        kn, cn, _, val = line.rstrip().rsplit(",", 3)
        if not _profile_row_matches(kernel_name, kn):
            continue
        if cn == "FETCH_SIZE":
            fetch = float(val)
        elif cn == "WRITE_SIZE":
            write = float(val)
    if fetch is None or write is None:
        return None, None
    return (2.0 * fetch + write) * 1024.0, os.path.relpath(path, ROOT)


def roofline_of(table):
    """roofline object of the dominant kernel (largest share of the timed region): bound = the roof it sits closer to.
    `frac` is ALGORITHMIC work / peak (module docstring); the pipe occupancy of the issued MFMAs is `mfma_busy_frac`."""
    if not table:
        return None
    name, k = max(table.items(), key=lambda kv: kv[1]["total_ms"])
    if k["hbm_frac"] > k["mfma_busy_frac"]:
        return {"bound": "hbm", "kernel": name, "achieved": k["algorithmic_GBps"], "peak": PEAK_HBM_GBS, "unit": "GB/s",
                "frac": k["hbm_frac"], "traffic": None, "avg_launch_ms": k["avg_ms"],
                "also": {"algorithmic_tflops": k["algorithmic_tflops"], "mfma_busy_frac": k["mfma_busy_frac"]}}
    return {"bound": "mfma", "kernel": name, "achieved": k["algorithmic_tflops"], "peak": k["mfma_peak_tflops"], "unit": "TFLOP/s",
            "frac": k["algorithmic_frac"], "traffic": None, "avg_launch_ms": k["avg_ms"],
            "mfma_busy_frac": k["mfma_busy_frac"], "mfma_per_product": k["mfma_per_product"],
            "executed_over_algorithmic": k["executed_over_algorithmic"],
            "also": {"hbm_frac": k["hbm_frac"], "algorithmic_GBps": k["algorithmic_GBps"]}}


def rccl_version():
    try:
        import torch
        v = torch.cuda.nccl.version()
        return ".".join(str(x) for x in v) if isinstance(v, (tuple, list)) else str(v)
    except Exception as e:       # version query only; never fail the bench on it
        return f"unknown ({type(e).__name__})"


def per_rank_ms(seconds, steps, world):
    """every rank's own ms per step of the timed region (all-gather of one float64 per rank): {"min", "max", "by_rank"} -- `value` is
    computed from the MAX (the contract); a slow GCD or a rank that waits in the all-reduce shows up here as the spread"""
    import torch
    import torch.distributed as dist
    ms = 1e3 * seconds / max(steps, 1)
    if world <= 1 or not dist.is_initialized():
        return {"min": ms, "max": ms, "by_rank": [ms]}
    dev = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend() == "nccl" else torch.device("cpu")
    mine = torch.tensor([ms], dtype=torch.float64, device=dev)
    out = [torch.zeros_like(mine) for _ in range(world)]
    dist.all_gather(out, mine)
    vals = [round(float(t.item()), 4) for t in out]
    return {"min": min(vals), "max": max(vals), "by_rank": vals}


# --------------------------------------------------------------------------------------------- main
def dry_run(args):
    """Launcher / process-group plumbing without GPU work (CPU test of the N > 1 path): rendezvous, sharding arithmetic
    (weak: N_rand per rank; strong: 32768 / world; render_only: 40 frames dealt round-robin), one all-reduce over a flat
    bucket of the gradient's size, the rank census and parameter-hash check of the real run, max-over-ranks timing,
    rank-0 JSON line."""
    import torch
    import torch.distributed as dist
    from nerf_pytorch_amd import parallel
    rank, world, dev = parallel.init_distributed(backend=args.backend or "gloo")
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but the process group has {world} rank(s)")
    n_global = 32768 if args.strong else args.rays * world
    lo, hi = parallel.shard_slice(n_global, rank, world)
    frames = parallel.frames_of_rank(40, rank, world)
    bucket = torch.full((595844,), float(rank + 1))
    t0 = time.perf_counter()
    if world > 1:
        dist.barrier()
        dist.all_reduce(bucket)
        dist.barrier()
    mine_s = time.perf_counter() - t0
    el = torch.tensor([mine_s], dtype=torch.float64)
    if world > 1:
        dist.all_reduce(el, op=dist.ReduceOp.MAX)
    by_rank = per_rank_ms(mine_s, 1, world)
    assert float(bucket[0]) == world * (world + 1) / 2
    seen = parallel.ranks_seen()
    same = parallel.ranks_identical([torch.arange(8.0)])
    counts = [None] * world
    if world > 1:
        dist.all_gather_object(counts, (hi - lo, len(frames)))
    else:
        counts = [(hi - lo, len(frames))]
    if rank == 0:
        print(json.dumps({"metric": "dry run (launcher plumbing only)", "value": None, "unit": "rays/s", "n_gpus": world,
                          "world_size": dist.get_world_size() if world > 1 else 1, "backend": args.backend or "gloo",
                          "scaling": "strong" if args.strong else "weak", "rays_per_rank": hi - lo,
                          "global_batch_rays": n_global, "rays_of_all_ranks": sum(c[0] for c in counts),
                          "frames_of_all_ranks": sum(c[1] for c in counts), "rccl_ranks_seen": seen, "ranks_identical": same,
                          "ms_per_step_by_rank": by_rank, "fabric_topology": parallel.fabric_topology(),
                          "dist_env": {k: os.environ.get(k) for k in ("HSA_ENABLE_IPC_MODE_LEGACY", "NERF_DIST_NO_DEVICE_ID", "NERF_DIST_TIMEOUT_S",
                                                                     "MASTER_ADDR", "MASTER_PORT")},
                          "dry_run": True}))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


class Session:
    """Everything one configuration needs: the two networks on the scene's weights, the fused optimizer, the render kwargs
    create_nerf builds (run_nerf.py:237-259), a pool of HBM-resident ray batches, and the step functions."""

    def __init__(self, config, args, rank, world, dev, n, strong):
        import numpy as np
        import torch
        import nerf_pytorch_amd as npa
        import workloads as wl
        from nerf_pytorch_amd import parallel
        self.npa, self.torch, self.wl, self.np = npa, torch, wl, np
        self.config, self.args, self.rank, self.world, self.dev, self.strong = config, args, rank, world, dev, strong
        cfg = self.cfg = wl.LEGO if config == "lego" else wl.FERN
        self.H, self.W, self.K = cfg["H"], cfg["W"], wl.intrinsics(cfg)
        if strong:
            self.n_global = 32768
            self.lo, self.hi = parallel.shard_slice(self.n_global, rank, world)
            self.n = self.hi - self.lo
        else:
            self.n = n
            self.n_global = n * world
            self.lo, self.hi = rank * n, (rank + 1) * n
        self.Pc, self.Pf = wl.scene_params()
        self.kw = dict(D=8, W=256, input_ch=63, output_ch=5, skips=[4], input_ch_views=27, use_viewdirs=True)
        self.net_c, self.net_f = npa.NeRF(**self.kw).to(dev), npa.NeRF(**self.kw).to(dev)
        self.net_c.load_state_dict(self.Pc)
        self.net_f.load_state_dict(self.Pf)
        parallel.broadcast_parameters([self.net_c, self.net_f])
        # torch.optim.Adam semantics (run_nerf.py:207), fused over the two flat parameter vectors (state_dict compatible)
        self.optimizer = npa.FlatAdam(list(self.net_c.parameters()) + list(self.net_f.parameters()), lr=5e-4, betas=(0.9, 0.999))
        self.kwargs_train = dict(network_query_fn=None, perturb=1.0, N_importance=N_IMPORTANCE, network_fine=self.net_f,
                                 N_samples=N_SAMPLES, network_fn=self.net_c, use_viewdirs=True, white_bkgd=cfg["white_bkgd"],
                                 raw_noise_std=cfg["raw_noise_std"], ndc=cfg["ndc"], lindisp=False, near=cfg["near"], far=cfg["far"])
        self.kwargs_test = dict(self.kwargs_train, perturb=False, raw_noise_std=0.)
        # synthetic data, resident in HBM: a pool of ray batches + targets.  weak: rank-dependent seeds; strong: one global
        # batch per pool slot, every rank takes its slice (and the random draws are made once for the full batch, below)
        self.pool = pool = 8 if not strong else 4
        make = wl.lego_batch if config == "lego" else wl.fern_batch
        if strong:
            self.batches = [make(self.n_global, seed=100 + i)[:, self.lo:self.hi].contiguous().to(dev) for i in range(pool)]
            self.targets = [torch.rand(self.n_global, 3, generator=torch.Generator().manual_seed(77 + i))[self.lo:self.hi].to(dev)
                            for i in range(pool)]
        else:
            self.batches = [make(self.n, seed=1000 * rank + i).to(dev) for i in range(pool)]
            gen = torch.Generator().manual_seed(77 + rank)
            self.targets = [torch.rand(self.n, 3, generator=gen).to(dev) for _ in range(pool)]
        self.strong_gen = torch.Generator(device=dev).manual_seed(4242) if strong else None
        self.sync = parallel.GradientSync([self.net_c, self.net_f]) if (world > 1 or parallel.FORCE_GROUP) else None
        # render_only (configs[4]): frames of a pose_spherical spiral (load_blender.py:75), dealt round-robin
        self.fr = args.frame
        fr_focal = cfg["focal"] * self.fr / cfg["W"]
        self.Kf = np.array([[fr_focal, 0, 0.5 * self.fr], [0, fr_focal, 0.5 * self.fr], [0, 0, 1]])
        self.spiral = [wl.pose_spherical(a, -30.0, 4.0) for a in np.linspace(-180, 180, 40 + 1)[:-1]]

    def strong_randoms(self):
        """random draws of the GLOBAL batch in the reference's order from a generator every rank seeds identically;
        each rank keeps its slice, so the N-GPU step computes exactly the 1-GPU N_rand=32768 step (SURVEY 8d-4;
        parallel.global_randoms, covered on 8 gloo ranks by tests/test_parallel_cpu.py)"""
        r = self.npa.parallel.global_randoms(self.n_global, N_SAMPLES, N_IMPORTANCE, self.cfg["raw_noise_std"], self.strong_gen, self.dev)
        return {k: v[self.lo:self.hi].contiguous() for k, v in r.items()}

    def train_step(self, i):
        npa = self.npa
        batch_rays, target_s = self.batches[i % self.pool], self.targets[i % self.pool]
        extra = {"randoms": self.strong_randoms()} if self.strong else {}
        rgb, disp, acc, extras = npa.render(self.H, self.W, self.K, chunk=self.args.chunk, rays=batch_rays, verbose=False, retraw=True,
                                            **self.kwargs_train, **extra)
        self.optimizer.zero_grad()
        loss = npa.img2mse(rgb, target_s) + npa.img2mse(extras["rgb0"], target_s)
        loss.backward()       # (the coarse bucket's all-reduce starts inside, under the fine network's backward)
        if self.sync is not None:
            self.sync.finish()
        self.optimizer.step()

    def infer_step(self, i):
        with self.torch.no_grad():
            self.npa.render(self.H, self.W, self.K, chunk=self.args.chunk, rays=self.batches[i % self.pool], retraw=True, **self.kwargs_test)

    def frame_step(self, i):
        frame_id = (self.rank + i * self.world) % len(self.spiral)      # parallel.frames_of_rank dealing: frame f -> rank f mod G
        with self.torch.no_grad():
            self.npa.render(self.fr, self.fr, self.Kf, chunk=self.args.chunk, c2w=self.spiral[frame_id][:3, :4], **self.kwargs_test)

    def step_fn(self, mode):
        return {"train": self.train_step, "infer": self.infer_step, "render_only": self.frame_step}[mode]

    def rays_per_step(self, mode):
        return self.fr * self.fr if mode == "render_only" else self.n

    def gate(self, precision, with_operands=True):
        """the north-star acceptance gate measured in this run: our image vs the image the REAL reference rendered for the
        same rays / weights (committed fixture tests/golden/gate_<config>.npz, generated by tests/golden/make_golden.py
        --round2 from /root/reference), against a teacher-scene target"""
        npa, torch, np, wl, dev = self.npa, self.torch, self.np, self.wl, self.dev
        hb = npa.hip_backend
        gpath = os.path.join(ROOT, "tests", "golden", f"gate_{self.config}.npz")
        if not os.path.exists(gpath):
            return None
        gold = np.load(gpath)
        gbatch = (wl.lego_batch(1024, seed=31) if self.config == "lego" else wl.fern_batch(1024, seed=32)).to(dev)
        # the timed steps have trained net_c / net_f: the gate is evaluated on the fixture's weights
        gate_nets = (npa.NeRF(**self.kw).to(dev), npa.NeRF(**self.kw).to(dev))
        gate_nets[0].load_state_dict(self.Pc)
        gate_nets[1].load_state_dict(self.Pf)
        kwargs = dict(self.kwargs_test, network_fn=gate_nets[0], network_fine=gate_nets[1])
        prev = npa.get_precision()
        npa.set_precision(precision)
        try:
            with torch.no_grad():
                rgb_g = npa.render(self.H, self.W, self.K, chunk=self.args.chunk, rays=gbatch, **kwargs)[0]
            gate = wl.precision_gate(rgb_g, torch.tensor(gold["rgb_ref"]), torch.tensor(gold["target"]))
            if with_operands and precision in ("bf16x3", "fp16x3", "fp16x3w"):
                # the one place these datapaths store less than fp32: the operands of the weight-gradient GEMM (the stored hi words:
                # 11 / 8 significant bits).  Gradient of the training loss against the fixture's target, this datapath vs the EXACT-fp32
                # datapath (the distance also contains the hierarchical sampling's sensitivity to forward rounding; the isolated
                # operand-rounding measurement is tests/test_gpu_fp16x3.py::test_split_backward_under_a_training_losss_upstream_gradient:
                # 1.9e-5 fp16 / 1.6e-4 bf16 of the gradient vs fp64)
                tgt = torch.tensor(gold["target"]).to(dev)

                def grads_with(prec):
                    npa.set_precision(prec)
                    try:
                        for m in gate_nets:
                            m.zero_grad()
                        rgb, _, _, ex = npa.render(self.H, self.W, self.K, chunk=self.args.chunk, rays=gbatch, **kwargs)
                        (npa.img2mse(rgb, tgt) + npa.img2mse(ex["rgb0"], tgt)).backward()
                        return torch.cat([gate_nets[0].last_flat_grad, gate_nets[1].last_flat_grad]).double()
                    finally:
                        npa.set_precision(precision)
                g16, g32 = grads_with(precision), grads_with("fp32")
                gate["gradient"] = {
                    **gradient_vs_fp64(dev, precision),
                    "rel_l2_vs_fp32_datapath": float((g16 - g32).norm() / g32.norm()),
                    "cosine_deficit": 1.0 - float((g16 * g32).sum() / (g16.norm() * g32.norm())), "rays": 1024,
                    "what": "training-loss gradient of both networks on the gate fixture, this datapath vs the exact-fp32 datapath"}
        finally:
            npa.set_precision(prev)
        gate.update(datapath=precision, rays=1024, bar_psnr_delta_db=0.01,
                    passed=bool(gate["psnr_delta_db"] < 0.01 and gate["target_psnr_db"] >= 30.0),
                    what="PSNR of our image vs the reference's image (real reference, CPU fp32, fixture gate_%s.npz) against a "
                         "teacher-scene target at target_psnr_db; north_star: psnr_delta_db < 0.01" % self.config)
        return gate

    def close(self):
        if self.sync is not None:
            self.sync.close()


def _guarded(errors, name, fn):
    """Run a SECONDARY measurement (a `configs` leg, a baseline): the headline has been measured by then and must be printed
    whatever happens here; a failure is recorded in the line under `errors` instead of ending the run."""
    t0 = time.perf_counter()
    try:
        return fn()
    except Exception as e:      # noqa: BLE001 (deliberately broad: out of memory, a missing fixture, a host without rocm tools ...)
        errors[name] = f"{type(e).__name__}: {e}"[:400]
        try:
            import torch
            import nerf_pytorch_amd as npa
            npa.hip_backend.WORKSPACE.clear()
            torch.cuda.empty_cache()
        except Exception:       # noqa: BLE001
            pass
        return None
    finally:
        LEG_SECONDS[name] = round(time.perf_counter() - t0, 2)


T_START = time.perf_counter()
LAST_LOCAL_SECONDS = [0.0]      # seconds of the last measure() call on THIS rank (before the max over ranks)
LEG_SECONDS = {}        # wall time of every secondary leg of this run (reported in the line: the default command has a time budget)


def main():
    global T_START
    T_START = time.perf_counter()
    args = parse_args()
    relaunch_if_needed(args)
    if args.dry_run:
        return dry_run(args)
    if args.cpu_full:
        print(json.dumps(cpu_baseline(args.config, n_rays=N_RAND, full_shape=True)))
        return

    import torch
    import torch.distributed as dist
    import nerf_pytorch_amd as npa
    from nerf_pytorch_amd import parallel
    hb = npa.hip_backend

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (the render hot path has no CPU fallback)")
    npa.set_precision(args.precision)
    rank, world, dev = parallel.init_distributed(backend=args.backend, force_group=True if args.force_group else None)
    grouped = world > 1 or (parallel.FORCE_GROUP and dist.is_initialized())      # collectives run (possibly over one rank)
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: launch through torch.distributed.run with "
                         f"--nproc-per-node {args.gpus} (or run `python bench.py --gpus {args.gpus}` without a torchrun environment)")
    if torch.cuda.device_count() < (world if "LOCAL_RANK" in os.environ else 1) and os.environ.get("NERF_ALLOW_SHARED_GPU") != "1":
        raise SystemExit(f"bench.py: {world} ranks but only {torch.cuda.device_count()} GPU(s) visible")
    if args.mode == "render_only" and args.config != "lego":
        raise SystemExit("bench.py: --mode render_only is BASELINE configs[4] (lego spiral); use --config lego")

    def barrier():
        if grouped:
            dist.barrier()
        torch.cuda.synchronize()

    def measure(precision, steps, warmup, fn, with_kernels):
        """(seconds for `steps` steps with the per-kernel timer OFF, max over ranks; per-kernel summary of a SEPARATE pass)"""
        npa.set_precision(precision)
        hb.TIMER = None
        for i in range(warmup):
            fn(i)
        barrier()
        t0 = time.perf_counter()
        for i in range(steps):
            fn(i)
        barrier()
        el = time.perf_counter() - t0
        LAST_LOCAL_SECONDS[0] = el          # this rank's own clock (the N > 1 line reports the spread over ranks)
        if grouped:
            t = torch.tensor([el], dtype=torch.float64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            el = float(t.item())
        kern = {}
        if with_kernels:
            timer = hb.KernelTimer()
            hb.TIMER = timer
            for i in range(max(2, min(steps, 6))):
                fn(i)
            torch.cuda.synchronize()
            hb.TIMER = None
            kern = timer.summary()
        return el, kern

    ses = Session(args.config, args, rank, world, dev, args.rays, args.strong)
    n, n_global = ses.n, ses.n_global
    step = ses.step_fn(args.mode)
    rays_per_step = ses.rays_per_step(args.mode)
    elapsed, kern = measure(args.precision, args.steps, args.warmup, step, with_kernels=True)
    by_rank = per_rank_ms(LAST_LOCAL_SECONDS[0], args.steps, world) if grouped else None

    # ---- N > 1: what the exchange costs alone, whether it was overlapped, and whether the ranks still agree
    multi = None
    if grouped and args.mode == "train":
        bufs = [torch.zeros(hb.N_PARAMS, device=dev) for _ in range(2)]
        for b in bufs:
            dist.all_reduce(b)
        barrier()
        t0 = time.perf_counter()
        reps = 10
        for _ in range(reps):
            for b in bufs:
                dist.all_reduce(b)
        barrier()
        multi = {"allreduce_ms": 1e3 * (time.perf_counter() - t0) / reps,
                 "allreduce_what": "2 x 2.38 MB fp32 all-reduce (both networks' buckets), back to back, nothing else on the GPU",
                 "overlap_started": ses.sync.started,
                 "overlap_started_what": f"exchanges started under the backward over {args.warmup + args.steps + max(2, min(args.steps, 6))} "
                                         "steps (two per step: the coarse network's bucket while the fine network's backward still runs, the fine network's "
                                         "the moment its backward ends; finish() waits and scales)",
                 "ranks_identical": parallel.ranks_identical([ses.net_c.flat_params(), ses.net_f.flat_params()]),
                 "rccl_ranks_seen": parallel.ranks_seen(), "backend": dist.get_backend(),
                 "rccl_version": rccl_version() if dist.get_backend() == "nccl" else None}
    elif grouped:
        multi = {"rccl_ranks_seen": parallel.ranks_seen()}
    if multi is not None:
        multi["ms_per_step_by_rank"] = by_rank
        multi["fabric_topology"] = parallel.fabric_topology() if rank == 0 else None
        multi["dist_env"] = {k: os.environ.get(k) for k in ("HSA_ENABLE_IPC_MODE_LEGACY", "NERF_DIST_NO_DEVICE_ID", "NERF_DIST_TIMEOUT_S",
                                                            "NCCL_DEBUG", "RCCL_MSCCL_ENABLE", "MASTER_ADDR")}

    # ---- secondary numbers of the same run (never the headline)
    other_infer = infer_chain = second = None
    if args.mode == "train":
        k_inf = max(5, args.steps // 2)
        el_i, _ = measure(args.precision, k_inf, 2, ses.infer_step, with_kernels=False)
        infer_chain = None
        if hb.INFER_ONE_LAUNCH and hb.render_infer_supported(N_SAMPLES, N_IMPORTANCE, args.precision):
            # the same batches through the chain of six launches, measured alternately with the one-launch kernel (the clock the
            # chip grants drifts over a run: whichever is measured first after the training steps looks ~1 % slower)
            el_c = None
            for _ in range(1):
                hb.INFER_ONE_LAUNCH = False
                try:
                    e, _ = measure(args.precision, k_inf, 2, ses.infer_step, with_kernels=False)
                finally:
                    hb.INFER_ONE_LAUNCH = True
                el_c = e if el_c is None else min(el_c, e)
                e, _ = measure(args.precision, k_inf, 2, ses.infer_step, with_kernels=False)
                el_i = min(el_i, e)
            infer_chain = n * world * k_inf / el_c
        other_infer = n * world * k_inf / el_i
    if not args.single_datapath and args.mode != "render_only":
        p2 = "bf16x3" if args.precision == "fp32" else "fp32"
        k2 = max(4, args.steps // 4)
        el2, kern2 = measure(p2, k2, 2, step, with_kernels=True)
        tab2 = kernel_table(kern2)
        second = {"dtype": DTYPE_NAME[p2].split(" ")[0], "value": n * world * k2 / el2, "unit": "rays/s",
                  "steps": k2, "ms_per_step": 1e3 * el2 / k2, "roofline": roofline_of(tab2), "kernels": _brief(tab2)}
    errors = {}
    two_word = None
    if not args.single_datapath and args.mode == "train" and args.precision == "fp16x3" and world == 1:
        # the price of fp32-class GRADIENTS on this datapath: fp16x3w = the same forward, two-word operands in the weight-gradient GEMM
        # (the instrument behind "the 11-bit operand storage costs training nothing detectable")
        def _tw():
            kt = max(5, args.steps // 2)
            elt, kernt = measure("fp16x3w", kt, 2, step, with_kernels=True)
            return {"dtype": DTYPE_NAME["fp16x3w"].split(" ")[0], "value": n * world * kt / elt, "unit": "rays/s", "steps": kt, "ms_per_step": 1e3 * elt / kt,
                    "kernels": _brief(kernel_table(kernt)), "gradient_vs_fp64": gradient_vs_fp64(dev, "fp16x3w"),
                    "headline_gradient_vs_fp64": gradient_vs_fp64(dev, "fp16x3")}
        two_word = _guarded(errors, "two_word_datapath", _tw)
    npa.set_precision(args.precision)
    default_run = (world == 1 and args.mode == "train" and args.config == "lego" and not args.strong and not args.no_configs
                   and args.rays == N_RAND)

    # ---- the headline leg again, for seconds instead of 0.15 s, with board power and clock sampled meanwhile: the loop `value`
    # stands for runs 200 k iterations (run_nerf.py:711), and the roof of the MFMA-bound kernels is the power cap
    def sustained_leg(fn, seconds, rays_per_call, chunk=50):
        hb.TIMER = None
        for i in range(5):
            fn(i)
        torch.cuda.synchronize()
        marks = []
        with PowerSampler() as ps:
            time.sleep(0.3)                                      # a few idle samples first
            idle = ps.summary()
            t0 = time.perf_counter()
            i = 0
            while time.perf_counter() - t0 < seconds:
                for _ in range(chunk):
                    fn(i)
                    i += 1
                torch.cuda.synchronize()
                marks.append((time.perf_counter(), i))
            t1 = time.perf_counter()
            busy = ps.summary(t0 + min(1.0, 0.25 * seconds), t1)
        per_s, last_t, last_i = [], t0, 0
        for t, k in marks:                                       # rate per ~1 s window: the DVFS settling, if any, shows here
            if t - last_t >= 1.0 or (t, k) == marks[-1]:
                per_s.append(round(rays_per_call * (k - last_i) / (t - last_t)))
                last_t, last_i = t, k
        return {"seconds": t1 - t0, "steps": i, "rays_per_s": rays_per_call * i / (t1 - t0), "ms_per_step": 1e3 * (t1 - t0) / i,
                "rays_per_s_by_second": per_s, "power": busy, "idle_power": idle}
    sustained = None
    if default_run and args.sustained_s > 0 and rank == 0:
        def _sus():
            out = {"train": sustained_leg(step, args.sustained_s, n), "infer": sustained_leg(ses.infer_step, max(2.0, 0.3 * args.sustained_s), n)}
            out["what"] = ("training steps (the headline's step function) / no_grad render() calls issued back to back for the stated seconds, "
                           "synchronised every 50 calls; power = board power and shader clock of GPU 0 sampled every 50 ms after the first second; "
                           "cap_w = the board's power limit")
            return out
        sustained = _guarded(errors, "sustained", _sus)

    # ---- the reduced INFERENCE class (fp16 main term + fp8 correction terms, csrc/field_ring8.h): never the headline; measured
    # alternately with the headline's inference, with its own north-star gate
    reduced_infer = None
    if not args.single_datapath and args.mode == "train" and args.precision == "fp16x3" and rank == 0:
        def _red():
            k_inf = max(5, args.steps // 2)
            best = {"fp16x3": None, "fp16_fp8c": None}
            for _ in range(1):
                for prec in ("fp16_fp8c", "fp16x3"):
                    e, _k = measure(prec, k_inf, 2, ses.infer_step, with_kernels=False)
                    best[prec] = e if best[prec] is None else min(best[prec], e)
            _e, kern_r = measure("fp16_fp8c", 4, 1, ses.infer_step, with_kernels=True)
            out = {"dtype": DTYPE_NAME["fp16_fp8c"], "rays_per_s": n * world * k_inf / best["fp16_fp8c"],
                   "headline_datapath_rays_per_s_same_interleaving": n * world * k_inf / best["fp16x3"],
                   "kernels": {k: round(v["avg_ms"], 4) for k, v in kernel_table(kern_r).items()},
                   "what": "no_grad render() of the same 4096-ray batches under set_precision('fp16_fp8c'): every product of the 256-wide layers = "
                           "W_hi16 x_hi16 (fp16 MFMA) + W_hi8 x_lo8 + W_lo8 x_hi8 (fp8 e4m3 MFMAs, K = 128), ~2^-15 per product, 2 instead of 3 "
                           "MFMA-equivalents -- in the REFINING pass (three quarters of the points); the coarse pass runs on the three-term fp16 "
                           "products (sample_pdf amplifies 2^-15 errors of the coarse weights: all-reduced images sit at 62-79 dB of the reference's "
                           "on the 4096- / 32,768-ray fixtures, this form at 89-100 dB, tests/test_gpu_golden_cfg.py); every ray's last fine sample "
                           "evaluated with the three-term products (guard launch); chain of 7 launches"}
            if default_run:     # BASELINE configs[4] on this class: 800x800 frames of the lego spiral
                ef, _k = measure("fp16_fp8c", 2, 1, ses.frame_step, with_kernels=False)
                out["render_only"] = {"rays_per_s": args.frame * args.frame * 2 / ef, "s_per_frame": ef / 2, "frame": args.frame, "chunk": args.chunk}
            if not args.no_gate:
                g = ses.gate("fp16_fp8c", with_operands=False)
                out["precision_gate"] = None if g is None else {k: g[k] for k in ("psnr_delta_db", "psnr_vs_ref_db", "target_psnr_db", "passed")}
            return out
        reduced_infer = _guarded(errors, "reduced_inference", _red)
        npa.set_precision(args.precision)

    bf16x3_leg = None
    if args.bf16x3_leg and not args.single_datapath and args.mode == "train" and args.precision == "fp16x3":
        # rounds 1-3's headline datapath in the same run (bf16 parts: 2^-17 products, 8-bit weight-gradient operands)
        def _b3():
            kb = max(5, args.steps // 2)
            elb, kernb = measure("bf16x3", kb, 2, step, with_kernels=True)
            out = {"dtype": DTYPE_NAME["bf16x3"], "value": n * world * kb / elb, "unit": "rays/s", "steps": kb, "ms_per_step": 1e3 * elb / kb,
                   "kernels": _brief(kernel_table(kernb))}
            if not args.no_gate and rank == 0:
                g = ses.gate("bf16x3", with_operands=False)
                out["precision_gate"] = None if g is None else {k: g[k] for k in ("psnr_delta_db", "psnr_vs_ref_db", "target_psnr_db", "passed")}
            return out
        bf16x3_leg = _guarded(errors, "bf16x3_datapath", _b3)
        npa.set_precision(args.precision)

    gate = None
    if not args.no_gate and rank == 0:
        gate = ses.gate(args.precision)
        if args.long and gate is not None:
            gate["training"] = convergence_table(dev, args.long_steps, checkpoints=(args.long_steps // 4, args.long_steps // 2), twins=args.long_twins, twins16=args.long_twins)
        elif default_run and gate is not None and not args.no_training_gate:
            # the converging pair of --long (teacher 5 / student 6), 300 steps, fp32 / fp32 one ulp away / headline
            tr = _guarded(errors, "precision_gate.training", lambda: convergence_table(dev, 300, seeds=(0,), which=("fp32", "fp32_twin", args.precision)))
            if tr is not None:
                gate["training"] = tr

    # ---- short legs of the other single-GPU configurations (BASELINE configs[2], [4] and the 32,768-ray batch of [3])
    legs = None
    if default_run:
        legs = {}
        lego_gate = None if gate is None else {k: gate[k] for k in ("psnr_delta_db", "psnr_vs_ref_db", "target_psnr_db", "passed") if k in gate}

        def leg_fern():
            fern = Session("fern", args, rank, world, dev, N_RAND, False)
            try:
                el, _ = measure(args.precision, 10, 3, fern.train_step, with_kernels=False)
                g = None if args.no_gate else fern.gate(args.precision, with_operands=False)
            finally:
                fern.close()
            return {"workload": "BASELINE configs[2]: fern-like 504x378, NDC rays near=0 far=1, raw_noise_std=1, N_rand=4096 x (64+128), training step",
                    "value": N_RAND * 10 / el, "unit": "rays/s", "steps": 10, "ms_per_step": 1e3 * el / 10,
                    "precision_gate": None if g is None else {k: g[k] for k in ("psnr_delta_db", "psnr_vs_ref_db", "target_psnr_db", "passed")}}

        def leg_render():
            el, _ = measure(args.precision, 2, 1, ses.frame_step, with_kernels=False)
            return {"workload": f"BASELINE configs[4] on one GPU: {args.frame}x{args.frame} frames of the lego spiral, no_grad render(c2w=...), chunks of {args.chunk}",
                    "value": args.frame * args.frame * 2 / el, "unit": "rays/s", "steps": 2, "s_per_frame": el / 2, "precision_gate": lego_gate}

        def leg_big():
            big = Session("lego", args, rank, world, dev, N_RAND, True)
            try:
                el, _ = measure(args.precision, 2, 1, big.train_step, with_kernels=False)
                plan = sys.modules[npa.parallel.__name__.rsplit(".", 1)[0] + ".render"].LAST_BACKWARD_PLAN      # (npa.render is the function)
                resident_gb = hb.saved_bytes(32768, N_SAMPLES, N_IMPORTANCE, args.precision) / 1e9
            finally:
                big.close()
                hb.WORKSPACE.clear()
                torch.cuda.empty_cache()
            return {"workload": f"the 32,768-ray global batch of BASELINE configs[3] on ONE GPU (lego 64+128, training step; backward plan '{plan[0]}': "
                                f"{-(-plan[1] // plan[2])} sub-chunk(s) of {plan[2]} rays that all keep their saved activations, {resident_gb:.1f} of the 288 GB)",
                    "backward_plan": list(plan), "saved_activations_gb": resident_gb,
                    "value": 32768 * 2 / el, "unit": "rays/s", "steps": 2, "ms_per_step": 1e3 * el / 2, "precision_gate": lego_gate}
        def leg_coarse():
            # BASELINE configs[0] is the reference's CPU-plumbing configuration; its SHAPE on the GPU (no fine network: one pass, one
            # compositing, no sample_pdf; the golden fixture lego_coarse_only.npz pins the same shape against the reference)
            n0 = 1024
            co = Session("lego", args, rank, world, dev, n0, False)
            try:
                kw0 = dict(co.kwargs_train, N_importance=0, network_fine=None)
                opt0 = npa.FlatAdam(list(co.net_c.parameters()), lr=5e-4, betas=(0.9, 0.999))

                def step0(i):
                    rgb, _, _, _ = npa.render(co.H, co.W, co.K, chunk=args.chunk, rays=co.batches[i % co.pool], verbose=False, retraw=True, **kw0)
                    opt0.zero_grad()
                    npa.img2mse(rgb, co.targets[i % co.pool]).backward()
                    opt0.step()
                el, _ = measure(args.precision, 20, 5, step0, with_kernels=False)
            finally:
                co.close()
            return {"workload": "the shape of BASELINE configs[0] on the GPU: lego 400x400, N_rand=1024 x 64 samples, no fine network, training step",
                    "value": n0 * 20 / el, "unit": "rays/s", "steps": 20, "ms_per_step": 1e3 * el / 20}
        def leg_train_loop():
            # What a user of the drop-in RUNS: the body of the reference's train() loop (run_nerf.py:711-784) line for line on this package's
            # names -- config_parser() on the text of the reference's configs/lego.txt (+ N_rand=4096, BASELINE configs[1]), create_nerf(args)
            # with ITS defaults (datapath, optimizer), a ray batch drawn EVERY step (sample_ray_batch: image choice on the host like the
            # reference's np.random.choice, pixels + rays in one launch), render(**render_kwargs_train), img2mse x 2, mse2psnr x 2,
            # backward, optimizer.step(), the learning-rate decay lines.  Nothing is pre-staged except the images / poses in HBM (the
            # reference keeps them in host memory and copies one image per step; here they are resident).
            import tempfile
            import numpy as np
            import workloads as wl
            with tempfile.TemporaryDirectory() as tmp:
                cfg_path = os.path.join(tmp, "lego.txt")
                with open(cfg_path, "w") as f:
                    f.write(LEGO_TXT)
                a = npa.config_parser().parse_args(["--config", cfg_path, "--N_rand", str(N_RAND), "--basedir", tmp, "--no_reload"])
            prev = npa.get_precision()
            npa.set_precision(npa.DEFAULT_PRECISION)          # what `import nerf_pytorch_amd` gives
            try:
                import contextlib
                import io
                with contextlib.redirect_stdout(io.StringIO()):        # create_nerf prints like the reference's
                    render_kwargs_train, _te, start, _gv, optimizer = npa.create_nerf(a, device=dev)
                render_kwargs_train["network_fn"].load_state_dict(ses.Pc)
                render_kwargs_train["network_fine"].load_state_dict(ses.Pf)
                render_kwargs_train.update(near=2.0, far=6.0)           # run_nerf.py:618-622
                H, W, K = ses.H, ses.W, ses.K
                g = torch.Generator().manual_seed(5)
                n_img = 100                                             # lego's 100 training views at half_res: 400 x 400
                images = torch.rand(n_img, H, W, 3, generator=g).to(dev)
                poses = torch.stack([torch.as_tensor(wl.pose_spherical(float(th), -30.0, 4.0), dtype=torch.float32)
                                     for th in np.linspace(-180, 180, n_img + 1)[:-1]]).to(dev)
                i_train = np.arange(n_img)
                rng = np.random.RandomState(0)
                state = {"global_step": start}

                def loop_body(i, what="all"):
                    img_i = rng.choice(i_train)
                    target = images[img_i]
                    pose = poses[img_i, :3, :4]
                    precrop = a.precrop_frac if state["global_step"] < a.precrop_iters else None
                    batch_rays, target_s = npa.sample_ray_batch(H, W, K, pose, target, a.N_rand, precrop_frac=precrop)
                    if what == "sampling":
                        return
                    rgb, disp, acc, extras = npa.render(H, W, K, chunk=a.chunk, rays=batch_rays, verbose=False, retraw=True, **render_kwargs_train)
                    optimizer.zero_grad()
                    img_loss = npa.img2mse(rgb, target_s)
                    trans = extras["raw"][..., -1]                      # noqa: F841 (the reference computes it too)
                    loss = img_loss
                    psnr = npa.mse2psnr(img_loss)                       # noqa: F841
                    if "rgb0" in extras:
                        img_loss0 = npa.img2mse(extras["rgb0"], target_s)
                        loss = loss + img_loss0
                        psnr0 = npa.mse2psnr(img_loss0)                 # noqa: F841
                    loss.backward()
                    optimizer.step()
                    decay_rate = 0.1
                    decay_steps = a.lrate_decay * 1000
                    new_lrate = a.lrate * (decay_rate ** (state["global_step"] / decay_steps))
                    for param_group in optimizer.param_groups:
                        param_group["lr"] = new_lrate
                    state["global_step"] += 1
                k = max(args.steps, 20)
                # measured alternately with the headline's step function (pre-staged batches), best of two each: the shader clock the
                # chip grants drifts by ~1 % over a run
                best = {"loop": None, "headline": None}
                for _ in range(2):
                    e, _k = measure(npa.get_precision(), k, 5, loop_body, with_kernels=False)
                    best["loop"] = e if best["loop"] is None else min(best["loop"], e)
                    e, _k = measure(npa.get_precision(), k, 5, step, with_kernels=False)
                    best["headline"] = e if best["headline"] is None else min(best["headline"], e)
                e_s, _k = measure(npa.get_precision(), 200, 5, lambda i: loop_body(i, "sampling"), with_kernels=False)
                # the host's share: the same loop body issued without waiting for the GPU (how far ahead of the GPU the Python side runs)
                t0 = time.perf_counter()
                for i in range(k):
                    loop_body(i)
                host_issue = (time.perf_counter() - t0) / k
                torch.cuda.synchronize()
            finally:
                npa.set_precision(prev)
            rate, head = N_RAND * k / best["loop"], N_RAND * k / best["headline"]
            return {"workload": "run_nerf.py:711-784 as written: config_parser(configs/lego.txt text, N_rand=4096) -> create_nerf defaults -> per step "
                                "sample_ray_batch (random image, precrop for the first 500 steps) -> render(**render_kwargs_train) -> img2mse x2 + mse2psnr x2 "
                                "-> backward -> optimizer.step() -> lr decay over param_groups; 100 synthetic 400x400 views resident in HBM",
                    "rays_per_s": rate, "unit": "rays/s", "steps": k, "ms_per_step": 1e3 * best["loop"] / k,
                    "headline_step_same_interleaving_rays_per_s": head, "ratio_to_headline_step": rate / head,
                    "datapath": DTYPE_NAME[npa.DEFAULT_PRECISION].split(" ")[0], "optimizer": type(optimizer).__name__,
                    "sample_ray_batch_ms": 1e3 * e_s / 200, "host_issue_ms_per_step": 1e3 * host_issue,
                    "host_issue_what": "wall time per step of the Python loop body alone (launch issue, no synchronisation inside the window): below "
                                       "ms_per_step means the host runs ahead of the GPU"}
        for name, fn in (("train_loop", leg_train_loop), ("fern_train", leg_fern), ("render_only", leg_render), ("batch_32768", leg_big), ("coarse_only_1024", leg_coarse)):
            res = _guarded(errors, "configs." + name, fn)
            if res is not None:
                legs[name] = res

    if rank == 0:
        cfg, H, W = ses.cfg, ses.H, ses.W
        fr = args.frame
        total_rays = rays_per_step * world * args.steps
        value = total_rays / elapsed
        kernels = kernel_table(kern)
        roofline = roofline_of(kernels)
        if roofline is not None:
            tr, src = pmc_traffic(roofline["kernel"], args.precision)
            if tr is not None:
                k = kernels[roofline["kernel"]]
                roofline["traffic"] = tr
                roofline["traffic_note"] = (f"bytes per launch (mean over coarse+fine launches) from {src} (rocprofv3 --pmc passes of this "
                                            f"command): 2*FETCH_SIZE + WRITE_SIZE; algorithmic bytes per launch here: "
                                            f"{k['algorithmic_GBps'] * 1e9 * k['avg_ms'] * 1e-3:.4g}")
            peak = PEAK_FP32_MFMA_TFLOPS if args.precision == "fp32" else PEAK_BF16_MFMA_TFLOPS
            alg_per_ray = FLOP_TRAIN_PER_RAY if args.mode == "train" else FLOP_FWD_PER_RAY
            roofline["step_algorithmic_frac"] = value * alg_per_ray / world / 1e12 / peak
            roofline["step_algorithmic_tflops"] = value * alg_per_ray / world / 1e12
            timed_ms = sum(v["total_ms"] for v in kernels.values())
            if timed_ms > 0:        # pipe occupancy of the whole step: sum over the timed kernels of busy_frac x their time / step time
                passes = max(2, min(args.steps, 6))
                roofline["step_mfma_busy_frac"] = sum(v["mfma_busy_frac"] * v["total_ms"] for v in kernels.values()) / passes / (1e3 * elapsed / args.steps)
        workload = {
            "train": f"{args.config}-like {W}x{H}, N_rand={n} rays/GPU x (64 coarse + 128 fine) samples, two 8x256 networks, perturb=1, "
                     + ("white_bkgd" if cfg["white_bkgd"] else f"NDC rays near=0 far=1, raw_noise_std={cfg['raw_noise_std']}")
                     + "; step = render() + MSE(rgb)+MSE(rgb0) + backward" + (" + RCCL grad all-reduce" if world > 1 else "") + " + fused Adam",
            "infer": f"{args.config}-like, {n} rays/GPU x (64+128) samples, no_grad render()",
            "render_only": f"lego render_only: {fr}x{fr} frames of a pose_spherical spiral, {fr * fr} rays/frame in chunks of {args.chunk}, "
                           f"no_grad render(c2w=...), one frame per GPU per step, frames dealt round-robin, no collective",
        }[args.mode]
        line = {
            "metric": {"train": "rays/sec (coarse+fine, 64+128 samples), training step",
                       "infer": "rays/sec (coarse+fine, 64+128 samples), inference",
                       "render_only": "rays/sec (coarse+fine, 64+128 samples), render_only frames"}[args.mode],
            "value": value, "unit": "rays/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True,
            "scaling": "strong" if args.strong else "weak", "vs_baseline": None,
            "dtype": DTYPE_NAME[args.precision], "data": "synthetic",
            "config": {"workload": workload, "global_batch_rays": rays_per_step * world if args.mode == "render_only" else n_global,
                       "parallelism": f"ray-shard dp{world}" if args.mode != "render_only" else f"frame-parallel x{world}",
                       "boundary": "nerf_pytorch_amd.render(H, W, K, chunk, rays=batch_rays, **render_kwargs) as run_nerf.py:760"},
            "world_size": dist.get_world_size() if grouped else 1,
            "collective": ((f"RCCL {rccl_version()} all-reduce (torch.distributed backend nccl)" if dist.get_backend() == "nccl"
                            else f"all-reduce over torch.distributed backend {dist.get_backend()}") + ", 2 x 2.38 MB fp32 per step, the coarse network's started under the fine network's backward"
                           if grouped and args.mode == "train" else None),
            "precision_gate": gate, "roofline": roofline, "kernels": kernels,
        }
        if multi is not None:
            line["multi_gpu"] = multi
        if legs is not None:
            line["configs"] = legs
        if other_infer is not None:
            line["inference_rays_per_s"] = other_infer
            if infer_chain is not None:
                line["inference"] = {"one_launch_rays_per_s": other_infer, "chain_of_launches_rays_per_s": infer_chain,
                                     "what": "no_grad render() of the same batches: render_infer_kernel (one launch per ray chunk, the default) "
                                             "vs sample_coarse -> field forward -> composite -> sample_fine -> field forward -> composite"}
        if second is not None:
            line["other_datapath"] = second
        if two_word is not None:
            line["two_word_datapath"] = two_word
        if bf16x3_leg is not None:
            line["bf16x3_datapath"] = bf16x3_leg
        if reduced_infer is not None:
            line["reduced_inference"] = reduced_infer
        if sustained is not None:
            line["sustained"] = sustained
            line["sustained_rays_per_s"] = sustained["train"]["rays_per_s"]
            line["power"] = sustained["train"]["power"]
        eb = None
        if world == 1 and not args.no_eager_baseline and args.mode != "render_only":
            eb = _guarded(errors, "rocm_eager_baseline", lambda: rocm_eager_baseline(
                args.config, dev, n, frame=args.frame if default_run else 0, chunk=args.chunk, pose=ses.spiral[0]))
        if eb is not None:
            line["rocm_eager_baseline"] = eb
            ref = eb["train_rays_per_s"] if args.mode == "train" else eb["infer_rays_per_s"]
            line["speedup_vs_rocm_eager"] = {"headline": value / ref}
            if second is not None:
                line["speedup_vs_rocm_eager"][second["dtype"]] = second["value"] / ref
            if bf16x3_leg is not None:
                line["speedup_vs_rocm_eager"]["bf16x3"] = bf16x3_leg["value"] / ref
            if sustained is not None:
                line["speedup_vs_rocm_eager"]["sustained"] = sustained["train"]["rays_per_s"] / ref
            if reduced_infer is not None:
                line["speedup_vs_rocm_eager"]["reduced_inference"] = reduced_infer["rays_per_s"] / eb["infer_rays_per_s"]
            if other_infer is not None:
                line["speedup_vs_rocm_eager"]["inference"] = other_infer / eb["infer_rays_per_s"]
            if eb.get("render_only") and legs and legs.get("render_only"):      # configs[4]: frames against frames
                line["speedup_vs_rocm_eager"]["render_only"] = legs["render_only"]["value"] / eb["render_only"]["rays_per_s"]
                if reduced_infer is not None and reduced_infer.get("render_only"):
                    line["speedup_vs_rocm_eager"]["reduced_render_only"] = reduced_infer["render_only"]["rays_per_s"] / eb["render_only"]["rays_per_s"]
        if world == 1 and not args.no_cpu_baseline:
            cb = _guarded(errors, "cpu_baseline", lambda: cpu_baseline(args.config))
            if cb is not None:
                line["cpu_baseline"] = cb
        line["leg_seconds"] = dict(LEG_SECONDS, total_since_start=round(time.perf_counter() - T_START, 2))
        if errors:
            line["errors"] = errors
        print(json.dumps(line))
    ses.close()
    if grouped:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
