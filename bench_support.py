"""Secondary measurements of bench.py, kept out of the contract file: the two reference baselines timed beside the product (the
oracle on the host cores, the reference's algorithm as eager PyTorch-ROCm ops on the same GPU), the board-power sampler of the
sustained leg, the training-equivalence table, and the text of the reference's lego config for the `train_loop` leg.  Nothing here is
on the product path; `oracle/` is imported by the two baseline functions (as the thing measured BESIDE the product) and by
`gradient_vs_fp64` (as the CHECKER of the gate's gradient: its fp64 network with the kernel's own ReLU pattern forced) -- never by
anything that produces `value`."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for _p in (ROOT, os.path.join(ROOT, "oracle")):
    if _p not in sys.path:
        sys.path.insert(0, _p)
N_SAMPLES, N_IMPORTANCE = 64, 128       # BASELINE.json: 64 coarse + 128 fine samples


# the text of the reference's configs/lego.txt (a config FILE of the reference's CLI is data the drop-in must accept unchanged; the
# reference tree is not on the GPU box, so the 13 settings are restated here for the train_loop leg)
LEGO_TXT = """expname = blender_paper_lego
basedir = ./logs
datadir = ./data/nerf_synthetic/lego
dataset_type = blender

no_batching = True

use_viewdirs = True
white_bkgd = True
lrate_decay = 500

N_samples = 64
N_importance = 128
N_rand = 1024

precrop_iters = 500
precrop_frac = 0.5

half_res = True
"""


# --------------------------------------------------------------------------------------------- baselines (reported beside)
CPU_FULL_SHAPE_FILE = os.path.join(ROOT, "profiles", "r06_cpu_full_shape.json")


def cpu_baseline(cfg_name, n_rays=256, full_shape=False):
    """The oracle (bit-identical restatement of the reference, CPU, fp32) timed on this box's host cores on a bounded
    sample of the same workload: training steps of n_rays rays x (64+128) samples.  full_shape (bench.py --cpu-full): ONE
    warm-up and ONE timed step of the metric's own 4096-ray batch on all host threads (~2 minutes); its record is committed under
    profiles/ and the default line carries it beside the 256-ray sample as `full_shape_rays_per_s`."""
    import torch
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import nerf_oracle as orc
    import workloads as wl
    cfg = wl.LEGO if cfg_name == "lego" else wl.FERN
    Pc, Pf = wl.scene_params()
    Pc = {k: v.requires_grad_(True) for k, v in Pc.items()}
    Pf = {k: v.requires_grad_(True) for k, v in Pf.items()}
    opt = torch.optim.Adam(list(Pc.values()) + list(Pf.values()), lr=5e-4, betas=(0.9, 0.999))
    batch = wl.lego_batch(n_rays, seed=1) if cfg_name == "lego" else wl.fern_batch(n_rays, seed=1)
    rays = orc.assemble_render_rays(cfg["H"], cfg["W"], wl.intrinsics(cfg), batch[0], batch[1], cfg["ndc"], cfg["near"], cfg["far"])
    target = torch.rand(n_rays, 3)
    std = cfg["raw_noise_std"]

    def step():
        opt.zero_grad()
        rnd = dict(t_rand=torch.rand(n_rays, N_SAMPLES), u=torch.rand(n_rays, N_IMPORTANCE))
        if std > 0:
            rnd.update(noise_c=torch.randn(n_rays, N_SAMPLES), noise_f=torch.randn(n_rays, N_SAMPLES + N_IMPORTANCE))
        out = orc.trace_rays(rays, Pc, Pf, N_SAMPLES, N_IMPORTANCE, perturb=1.0, white_bkgd=cfg["white_bkgd"],
                             raw_noise_std=std, **rnd)
        loss = orc.mse(out["rgb_map"], target) + orc.mse(out["rgb0"], target)
        loss.backward()
        opt.step()
    step()
    t0 = time.perf_counter()
    reps = 0
    while reps < (1 if full_shape else 2) or (not full_shape and time.perf_counter() - t0 < 8.0 and reps < 20):
        step()
        reps += 1
    dt = (time.perf_counter() - t0) / reps
    out = {"value": n_rays / dt, "unit": "rays/s", "cores": torch.get_num_threads(), "kind": "port",
           "sample": f"{reps} training steps of {n_rays} rays x (64+128) samples, {cfg_name} workload (oracle = bit-identical "
                     f"restatement of the reference, torch CPU fp32, {torch.get_num_threads()} threads of "
                     f"{os.cpu_count()} host CPUs)"}
    if full_shape:
        model = next((l.split(":", 1)[1].strip() for l in open("/proc/cpuinfo") if l.startswith("model name")), "unknown") if os.path.exists("/proc/cpuinfo") else "unknown"
        out.update(seconds_per_step=dt, host_cpus=os.cpu_count(), cpu_model=model, torch=torch.__version__, rays=n_rays)
    elif os.path.exists(CPU_FULL_SHAPE_FILE):
        import json
        rec = json.load(open(CPU_FULL_SHAPE_FILE))
        out["full_shape_rays_per_s"] = rec.get("value")
        out["full_shape"] = {k: rec.get(k) for k in ("rays", "seconds_per_step", "cores", "host_cpus", "cpu_model")}
        out["full_shape"]["source"] = os.path.relpath(CPU_FULL_SHAPE_FILE, ROOT) + " (bench.py --cpu-full on an MI355X box's host; the default run keeps the 256-ray sample)"
    return out


def rocm_eager_baseline(cfg_name, dev, n_rays, steps=5, frame=0, chunk=32768, pose=None):
    """Baseline leg: the reference's own algorithm as eager PyTorch-ROCm ops on this GPU (the oracle's torch ops with CUDA
    tensors = what run_nerf.py executes after set_default_tensor_type('torch.cuda.FloatTensor'), run_nerf.py:876), same
    workload shape, training step and no_grad render.  It is the denominator of the north-star '>= 10x' target; like
    cpu_baseline it is timed beside the product, never part of `value`.  Every step is timed on its own (synchronised):
    the rate is that of the MEDIAN step, and the spread is reported."""
    import torch
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import nerf_oracle as orc
    import workloads as wl
    cfg = wl.LEGO if cfg_name == "lego" else wl.FERN
    Pc, Pf = wl.scene_params()
    Pc = {k: v.to(dev).requires_grad_(True) for k, v in Pc.items()}
    Pf = {k: v.to(dev).requires_grad_(True) for k, v in Pf.items()}
    opt = torch.optim.Adam(list(Pc.values()) + list(Pf.values()), lr=5e-4, betas=(0.9, 0.999))
    batch = (wl.lego_batch(n_rays, seed=1) if cfg_name == "lego" else wl.fern_batch(n_rays, seed=1)).to(dev)
    K = wl.intrinsics(cfg)
    target = torch.rand(n_rays, 3, device=dev)
    std = cfg["raw_noise_std"]

    def train():
        rays = orc.assemble_render_rays(cfg["H"], cfg["W"], K, batch[0], batch[1], cfg["ndc"], cfg["near"], cfg["far"])
        rnd = dict(t_rand=torch.rand(n_rays, N_SAMPLES, device=dev), u=torch.rand(n_rays, N_IMPORTANCE, device=dev))
        if std > 0:
            rnd.update(noise_c=torch.randn(n_rays, N_SAMPLES, device=dev), noise_f=torch.randn(n_rays, N_SAMPLES + N_IMPORTANCE, device=dev))
        out = orc.trace_rays(rays, Pc, Pf, N_SAMPLES, N_IMPORTANCE, perturb=1.0, white_bkgd=cfg["white_bkgd"], raw_noise_std=std,
                             retraw=True, **rnd)
        opt.zero_grad()
        loss = orc.mse(out["rgb_map"], target) + orc.mse(out["rgb0"], target)
        loss.backward()
        opt.step()

    def infer():
        with torch.no_grad():
            rays = orc.assemble_render_rays(cfg["H"], cfg["W"], K, batch[0], batch[1], cfg["ndc"], cfg["near"], cfg["far"])
            orc.trace_rays(rays, Pc, Pf, N_SAMPLES, N_IMPORTANCE, perturb=0.0, white_bkgd=cfg["white_bkgd"], retraw=True)
    res, spread = {}, {}
    for name, fn in (("train", train), ("infer", infer)):
        fn()
        torch.cuda.synchronize()
        times = []
        for _ in range(steps):
            t0 = time.perf_counter()
            fn()
            torch.cuda.synchronize()
            times.append(time.perf_counter() - t0)
        times.sort()
        res[name] = n_rays / times[len(times) // 2]
        spread[name] = {"fastest_step_rays_per_s": n_rays / times[0], "slowest_step_rays_per_s": n_rays / times[-1]}
    frame_leg = None
    if frame > 0 and pose is not None and cfg_name == "lego":
        # BASELINE configs[4]: one frame of the spiral as render_path runs it (run_nerf.py:137-175): get_rays for the full image,
        # batchify_rays in chunks of `chunk`, the network in netchunk slices
        focal = cfg["focal"] * frame / cfg["W"]
        Kf = [[focal, 0, 0.5 * frame], [0, focal, 0.5 * frame], [0, 0, 1]]

        ro, rd = orc.pinhole_rays(frame, frame, Kf, torch.as_tensor(pose[:3, :4], dtype=torch.float32))
        ro, rd = ro.to(dev), rd.to(dev)         # (ray generation itself stays outside the timed frame: it favours the baseline)

        def one_frame():
            with torch.no_grad():
                rays = orc.assemble_rays(ro, rd, cfg["near"], cfg["far"])
                orc.trace_in_chunks(rays, chunk, P_coarse=Pc, P_fine=Pf, n_coarse=N_SAMPLES, n_fine=N_IMPORTANCE, perturb=0.0,
                                    white_bkgd=cfg["white_bkgd"])
        with torch.no_grad():           # warm-up on the first chunk of the frame (allocator, kernel selection)
            orc.trace_in_chunks(orc.assemble_rays(ro.reshape(-1, 3)[:chunk], rd.reshape(-1, 3)[:chunk], cfg["near"], cfg["far"]), chunk, P_coarse=Pc,
                                P_fine=Pf, n_coarse=N_SAMPLES, n_fine=N_IMPORTANCE, perturb=0.0, white_bkgd=cfg["white_bkgd"])
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        one_frame()
        torch.cuda.synchronize()
        tf = time.perf_counter() - t0
        frame_leg = {"rays_per_s": frame * frame / tf, "s_per_frame": tf, "frame": frame, "chunk": chunk}
    del Pc, Pf, opt
    torch.cuda.empty_cache()
    return {"train_rays_per_s": res["train"], "infer_rays_per_s": res["infer"], "render_only": frame_leg, "unit": "rays/s", "steps": steps,
            "rate_of": "median step", "spread": spread,
            "what": f"reference algorithm as eager PyTorch-ROCm ops on this GPU (oracle ops on cuda tensors), {n_rays} rays x (64+128), "
                    f"{cfg_name} workload, torch {torch.__version__}, timed after the product legs (warm GPU)"}


def gradient_vs_fp64(dev, precision, n_rays=256, n_samples=192):
    """What a datapath's BACKWARD arithmetic costs, isolated from the sampler (the checker's role, like the PSNR gate's fixture): one
    field evaluation of the fine network over n_rays x n_samples points, upstream gradient = the adjoint of raw2outputs for an MSE
    loss (what a training step produces), parameter gradient through the C ABI on `precision` against fp64 autograd of the oracle's
    network evaluated ON THIS GPU with the kernel's own ReLU pattern forced (a unit within rounding of zero legitimately takes either
    side of its kink; a cross-forward comparison would measure sample_pdf's sensitivity instead -- tools/EXPERIMENTS.md, round 4).
    The reference's own fp32-vs-fp64 gradient noise is 3.5e-5 of max|g|."""
    import torch
    import nerf_oracle as orc
    import nerf_pytorch_amd as npa
    import workloads as wl
    hb = npa.hip_backend
    _Pc, Pf = wl.scene_params()
    net = npa.NeRF(D=8, W=256, input_ch=63, output_ch=5, skips=[4], input_ch_views=27, use_viewdirs=True).to(dev)
    net.load_state_dict(Pf)
    g = torch.Generator().manual_seed(11)
    rays = wl.synthetic_rays(n_rays, seed=29).to(dev)
    z = torch.sort(torch.rand(n_rays, n_samples, generator=g) * 4.0 + 2.0, -1)[0].to(dev)
    target = torch.rand(n_rays, 3, generator=g).to(dev)
    packed = net.packed_params(precision)
    raw, act = hb.field_fwd(packed, rays, z, save_act=True, precision=precision)
    rgb, _, _, _, _ = hb.raw2outputs(raw, z, rays, 11, None, 0.0, True, rays_d_offset=3)
    d_rgb = (2.0 / (3 * n_rays)) * (rgb - target)
    d_raw = hb.raw2outputs_bwd(raw, z, rays, 11, None, 0.0, True, d_rgb.contiguous(), None, None, rays_d_offset=3)
    masks = hb.relu_patterns(act, n_rays, n_samples, precision)
    grad = torch.empty(hb.N_PARAMS, dtype=torch.float32, device=dev)
    hb.field_bwd(packed, act, d_raw, grad, accumulate=False, precision=precision, params=net.flat_params())
    hb.WORKSPACE.give(act)
    P64 = {k: v.to(dev).double().requires_grad_(True) for k, v in Pf.items()}
    d64 = d_raw.double().reshape(-1, 4)
    chunk = 64
    for lo in range(0, n_rays, chunk):
        r = rays[lo:lo + chunk].double()
        pts = (r[:, None, 0:3] + r[:, None, 3:6] * z[lo:lo + chunk, :, None].double()).reshape(-1, 3)
        dirs = r[:, None, 8:11].expand(r.shape[0], n_samples, 3).reshape(-1, 3)
        feats = torch.cat([orc.posenc(pts, 10), orc.posenc(dirs, 4)], -1)
        out, _ = orc.field_mlp_forced_relu(P64, feats, [m[lo * n_samples:(lo + chunk) * n_samples] for m in masks])
        (out * d64[lo * n_samples:(lo + chunk) * n_samples]).sum().backward()
    ref = torch.cat([P64[nm].grad.reshape(-1) for nm, _, _ in hb.param_table()])
    err = grad.double() - ref
    return {"rel_l2_vs_fp64": float(err.norm() / ref.norm()), "max_err_over_max_grad": float(err.abs().max() / ref.abs().max()),
            "points": n_rays * n_samples,
            "vs_fp64_what": "parameter gradient of ONE field evaluation (fine network, training-loss upstream gradient) through the C ABI vs fp64 autograd "
                    "of the oracle's network with the kernel's own ReLU pattern forced, fp64 on this GPU: the backward's arithmetic alone "
                    "(reference fp32 vs fp64: 3.5e-5 of max|g|)"}


# --------------------------------------------------------------------------------------------- board power and clocks
class PowerSampler:
    """Board power and shader clock of GPU 0 sampled from a second thread while a leg runs ("put the roof in the
    record").  Source: the amdgpu hwmon files when they exist (power1_average / power1_input in microwatts, freq1_input in Hz:
    ~0.1 ms per sample), else `rocm-smi --showpower --showclocks --csv` (~40 ms per sample).  Never raises: a box without
    either reports `source: none`."""

    def __init__(self, period_s=0.05):
        import glob
        import threading
        self.period, self.samples, self._stop = period_s, [], threading.Event()
        self.cap_w, self.source = None, "none"
        self._power = self._freq = None
        # a node exposes the hwmon directories of ALL its GPUs, also those of other tenants: take the one whose PCI address is
        # the address of the device this process computes on (torch device 0)
        want = None
        try:
            import torch
            pr = torch.cuda.get_device_properties(0)
            want = f"{pr.pci_domain_id:04x}:{pr.pci_bus_id:02x}:{pr.pci_device_id:02x}.0"
        except Exception:               # noqa: BLE001
            pass
        self.pci = want
        for hw in sorted(glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*")):
            if want is None or os.path.basename(os.path.realpath(os.path.dirname(os.path.dirname(hw)))).lower() != want:
                continue
            pw = [f for f in ("power1_average", "power1_input") if os.path.exists(os.path.join(hw, f))]
            if pw:
                self._power = os.path.join(hw, pw[0])
                fq = os.path.join(hw, "freq1_input")
                self._freq = fq if os.path.exists(fq) else None
                cap = os.path.join(hw, "power1_cap")
                try:
                    self.cap_w = float(open(cap).read()) / 1e6
                except Exception:       # noqa: BLE001
                    pass
                self.source = "hwmon"
                break
        if self._power is None:
            import shutil
            if shutil.which("rocm-smi"):
                self.source = "rocm-smi"
        self._thread = threading.Thread(target=self._run, daemon=True)

    def _read(self):
        if self.source == "hwmon":
            try:
                w = float(open(self._power).read()) / 1e6
                mhz = float(open(self._freq).read()) / 1e6 if self._freq else None
                return w, mhz
            except Exception:           # noqa: BLE001
                return None
        if self.source == "rocm-smi":
            import subprocess
            try:
                out = subprocess.run(["rocm-smi", "--showpower", "--showclocks", "--showmaxpower", "--csv"], capture_output=True, text=True, timeout=5).stdout
                rows = [r for r in out.strip().splitlines() if r.startswith("card")]
                head = [r for r in out.strip().splitlines() if r.startswith("device")]
                if not rows or not head:
                    return None
                cols, vals = head[0].split(","), rows[0].split(",")
                rec = dict(zip(cols, vals))
                w = next((float(v) for k, v in rec.items() if "Current Socket Graphics Package Power" in k or "Average Graphics Package Power" in k), None)
                cap = next((float(v) for k, v in rec.items() if "Max Graphics Package Power" in k), None)
                if cap:
                    self.cap_w = cap
                sclk = next((v for k, v in rec.items() if k.startswith("sclk clock speed")), None)
                mhz = float(sclk.strip("()Mhz")) if sclk else None
                return (w, mhz) if w is not None else None
            except Exception:           # noqa: BLE001
                return None
        return None

    def _run(self):
        while not self._stop.is_set():
            r = self._read()
            if r is not None:
                self.samples.append((time.perf_counter(),) + r)
            self._stop.wait(self.period)

    def __enter__(self):
        self._thread.start()
        return self

    def __exit__(self, *exc):
        self._stop.set()
        self._thread.join(timeout=10)

    def summary(self, t0=None, t1=None):
        rows = [r for r in self.samples if (t0 is None or r[0] >= t0) and (t1 is None or r[0] <= t1)]
        if not rows:
            return {"source": self.source, "samples": 0, "pci": getattr(self, "pci", None)}
        w = sorted(r[1] for r in rows)
        f = [r[2] for r in rows if r[2] is not None]
        return {"source": self.source, "pci": getattr(self, "pci", None), "samples": len(rows), "cap_w": self.cap_w, "mean_w": sum(w) / len(w), "p95_w": w[min(len(w) - 1, int(0.95 * len(w)))],
                "max_w": w[-1], "sclk_mhz_mean": (sum(f) / len(f)) if f else None, "sclk_mhz_min": min(f) if f else None,
                "frac_of_cap": (sum(w) / len(w) / self.cap_w) if self.cap_w else None}


# (teacher, student) pairs of the training-equivalence table: ("scene", t, s) = scene_params(t) as the teacher, scene_params(s) as the student's
# initialisation; ("near", t, eps) = the student starts at the teacher's weights perturbed by a relative eps (a partly converged model).
# Chosen by tools/exp_pairs.py (round 5, gpurun_out/exp_pairs.log -> profiles/r05_pair_search.txt) among 22 candidates as pairs that
# CONVERGE: held-out PSNR 43.4 / 37.6 / 46.4 dB after 500 steps on fp32 and fp16x3 alike, so the fp32-vs-twin distance -- the yardstick --
# is hundredths of a dB (rounds 3-4 had two pairs stuck at 17-19 dB with a twin distance of 1.1 dB).
CONVERGING_PAIRS = (("scene", 5, 6), ("scene", 4, 6), ("scene", 2, 3))


def convergence_table(dev, steps, seeds=(0, 1, 2), n_batch=1024, which=None, pairs=CONVERGING_PAIRS, checkpoints=None, twins=1, twins16=0):
    """Training equivalence of the datapaths, measured instead of argued: the same student is fitted to a teacher scene's images
    with the fused Adam for `steps` steps of `n_batch` rays, once per datapath and seed, with identical initialisation, batch order
    and random draws; held-out PSNR (2048 rays, evaluated on the exact fp32 datapath) at every checkpoint and after the last step.
    Per datapath: mean and spread (max - min) over the seeds, and the largest per-seed difference to the fp32 datapath (over all
    checkpoints: `max_abs_diff_to_fp32_db_any_checkpoint`).  `fp32_twin`, `fp32_twin2` ... (`twins` of them) are the fp32 datapath
    itself started one ulp away (each with its own perturbation): training is chaotic in the rounding, so a datapath is equivalent
    when it stays inside the spread of the fp32 family.  `twins16` adds as many perturbed starts of the fp16x3 datapath, so that the
    two FAMILIES can be compared (`families`: per seed the [min, max] of each family's final PSNR)."""
    import math
    import torch
    import nerf_pytorch_amd as npa
    import workloads as wl
    kw = dict(D=8, W=256, input_ch=63, output_ch=5, skips=[4], input_ch_views=27, use_viewdirs=True)
    checkpoints = sorted(set(c for c in (checkpoints or ()) if 0 < c < steps)) + [steps]

    def net(P):
        m = npa.NeRF(**kw).to(dev)
        m.load_state_dict(P)
        return m
    rk = dict(N_samples=64, N_importance=128, white_bkgd=True, raw_noise_std=0.)
    prev_prec = npa.get_precision()
    results = {}
    try:
        for seed in seeds:
            kind, a, b = pairs[seed % len(pairs)]
            Tc, Tf = wl.scene_params(seed=a)
            Sc, Sf = wl.scene_params(seed=b) if kind == "scene" else wl.teacher_params(seed=a, eps=b)
            tc, tf = net(Tc), net(Tf)
            pool = wl.synthetic_rays(n_batch * 16, seed=770 + seed).to(dev)
            held = wl.synthetic_rays(2048, seed=780 + seed).to(dev)
            npa.set_precision("fp32")
            with torch.no_grad():
                tgt_pool = torch.cat([npa.render_rays(pool[i:i + 4096], tc, None, network_fine=tf, perturb=0., **rk)["rgb_map"]
                                      for i in range(0, pool.shape[0], 4096)])
                tgt_held = npa.render_rays(held, tc, None, network_fine=tf, perturb=0., **rk)["rgb_map"]

            def psnr(nc, nf):
                npa.set_precision("fp32")
                with torch.no_grad():
                    out = npa.render_rays(held, nc, None, network_fine=nf, perturb=0., **rk)["rgb_map"]
                mse = float(((out - tgt_held) ** 2).mean())
                if not (mse > 0.0 and math.isfinite(mse)):
                    raise RuntimeError(f"convergence_table: held-out mse {mse!r}; out [{float(out.min())}, {float(out.max())}] "
                                       f"nan {int(torch.isnan(out).sum())}, target [{float(tgt_held.min())}, {float(tgt_held.max())}], "
                                       f"same storage {out.data_ptr() == tgt_held.data_ptr()}")
                return -10 * math.log10(mse)
            runs = [("fp32", "fp32", 0)] + [("fp32_twin" + (str(t) if t > 1 else ""), "fp32", t) for t in range(1, twins + 1)]
            runs += [("fp16x3", "fp16x3", 0)] + [(f"fp16x3_twin{t}", "fp16x3", t) for t in range(1, twins16 + 1)] + [("bf16x3", "bf16x3", 0)]
            for name, prec, twin in runs:
                if which is not None and name not in which and not (twin > 1 and name.rstrip("0123456789") in which):
                    continue
                torch.manual_seed(seed)
                nc, nf = net(Sc), net(Sf)
                if twin:
                    # the yardstick: the SAME datapath started 1e-7 (relative, ~1 ulp) away -- how far two runs of one
                    # datapath drift apart in this many steps
                    gt = torch.Generator(device="cpu").manual_seed(5000 * twin + seed)
                    with torch.no_grad():
                        for p in list(nc.parameters()) + list(nf.parameters()):
                            p.mul_((1.0 + 1e-7 * torch.randn(p.shape, generator=gt)).to(dev))
                opt = npa.FlatAdam(list(nc.parameters()) + list(nf.parameters()), lr=5e-4, betas=(0.9, 0.999))
                g = torch.Generator(device="cpu").manual_seed(1000 + seed)
                at = []
                for it in range(1, steps + 1):
                    idx = torch.randint(0, pool.shape[0], (n_batch,), generator=g).to(dev)
                    npa.set_precision(prec)
                    opt.zero_grad()
                    out = npa.render_rays(pool[idx], nc, None, network_fine=nf, perturb=1.0, **rk)
                    (npa.img2mse(out["rgb_map"], tgt_pool[idx]) + npa.img2mse(out["rgb0"], tgt_pool[idx])).backward()
                    opt.step()
                    if it in checkpoints:
                        at.append(psnr(nc, nf))
                results.setdefault(name, []).append(at)
    finally:
        npa.set_precision(prev_prec)
    table = {}
    for name, vals in results.items():
        last = [v[-1] for v in vals]
        table[name] = {"psnr_db_per_seed": [round(v, 3) for v in last], "mean_db": sum(last) / len(last), "spread_db": max(last) - min(last),
                       "max_abs_diff_to_fp32_db": max(abs(a[-1] - b[-1]) for a, b in zip(vals, results["fp32"])),
                       "max_abs_diff_to_fp32_db_any_checkpoint": max(abs(x - y) for a, b in zip(vals, results["fp32"]) for x, y in zip(a, b))}
        if len(checkpoints) > 1:
            table[name]["psnr_db_per_seed_at_checkpoints"] = [[round(x, 3) for x in v] for v in vals]
    families = None
    if twins > 1 or twins16 > 0:
        fam = lambda prefix: [[results[k][i][-1] for k in results if k == prefix or k.startswith(prefix + "_twin")] for i in range(len(seeds))]
        f32, f16 = fam("fp32"), fam("fp16x3") if "fp16x3" in results else None
        families = {"fp32_final_db_min_max_per_seed": [[round(min(v), 3), round(max(v), 3)] for v in f32],
                    "fp32_family_spread_db_per_seed": [round(max(v) - min(v), 3) for v in f32], "runs_per_seed": {"fp32": len(f32[0])}}
        if f16:
            families.update({"fp16x3_final_db_min_max_per_seed": [[round(min(v), 3), round(max(v), 3)] for v in f16],
                             "fp16x3_family_mean_minus_fp32_family_mean_db_per_seed": [round(sum(a) / len(a) - sum(b) / len(b), 3) for a, b in zip(f16, f32)]})
            families["runs_per_seed"]["fp16x3"] = len(f16[0])
    return {"steps": steps, "checkpoints": checkpoints, "rays_per_step": n_batch, "seeds": list(seeds), "pairs": [list(pairs[s % len(pairs)]) for s in seeds],
            "what": "student fitted to a teacher scene ('scene': another scene's weights; 'near': the teacher's weights perturbed by a relative eps), fused "
                    "Adam lr 5e-4, same init / batches / draws per datapath; held-out PSNR (2048 rays, evaluated on the fp32 datapath) after the last step "
                    "and at the checkpoints; fp32_twin = the fp32 datapath with the initial parameters perturbed by 1e-7 relative",
            "datapaths": table, **({"families": families} if families else {})}


