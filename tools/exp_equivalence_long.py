"""Round 6: the training-equivalence families at the reference's horizon class -- >= 10,000 optimizer steps
(run_nerf.py:701 trains 200,000; rounds 3-5 stopped at 2,000) -- for THREE datapaths: the exact-fp32 anchor, the default fp16x3
(11-bit operands in the weight-gradient GEMM) and fp16x3w (two-word operands: the same forward, gradients of the forward's product
class).  Per converging (teacher, student) pair and datapath a FAMILY of 1 + N runs started one ulp apart (1e-7 relative), same
batches and draws, fused Adam lr 5e-4, 1024 rays per step; held-out PSNR (2048 rays, evaluated on the fp32 datapath) at every
checkpoint.  Output: one JSON (per run, per checkpoint) + the decision rule evaluated on it:

    fp16x3 stays the default iff, at the last checkpoint of every pair,
      (a) |mean(fp16x3 family) - mean(fp32 family)| <= 2 x sd(fp32 family)   [no systematic deficit], and
      (b) sd(fp16x3 family) / sd(fp32 family) does not keep growing: its value at the last checkpoint is <= 1.5 x its median over the
          checkpoints from 2,000 steps on                                     [the extra gradient noise is not amplified without bound];
    otherwise fp16x3w becomes the default and the headline is re-quoted on it.

usage: python tools/exp_equivalence_long.py [--steps 10000] [--twins 4] [--pairs 0,1] [--out gpurun_out/r06_equivalence.json]
       [--twin-range a:b]  (more runs of the same families; tools/merge_equivalence.py pools the files)"""
import argparse
import json
import math
import os
import statistics
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import nerf_pytorch_amd as npa  # noqa: E402
import workloads as wl  # noqa: E402
from bench_support import CONVERGING_PAIRS  # noqa: E402


def run_family(dev, pair_index, steps, checkpoints, n_batch, precisions, twins, log, twin_range=None):
    kind, a, b = CONVERGING_PAIRS[pair_index]
    kw = dict(D=8, W=256, input_ch=63, output_ch=5, skips=[4], input_ch_views=27, use_viewdirs=True)

    def net(P):
        m = npa.NeRF(**kw).to(dev)
        m.load_state_dict(P)
        return m
    rk = dict(N_samples=64, N_importance=128, white_bkgd=True, raw_noise_std=0.)
    Tc, Tf = wl.scene_params(seed=a)
    Sc, Sf = wl.scene_params(seed=b) if kind == "scene" else wl.teacher_params(seed=a, eps=b)
    tc, tf = net(Tc), net(Tf)
    seed = pair_index
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
        return -10 * math.log10(mse) if (mse > 0 and math.isfinite(mse)) else float("nan")
    results = {}
    for prec in precisions:
        for twin in (range(0, twins + 1) if twin_range is None else range(*twin_range)):
            name = prec if twin == 0 else f"{prec}_twin{twin}"
            torch.manual_seed(seed)
            nc, nf = net(Sc), net(Sf)
            if twin:    # the yardstick: the SAME datapath started 1e-7 (relative, ~1 ulp) away
                gt = torch.Generator(device="cpu").manual_seed(5000 * twin + seed)
                with torch.no_grad():
                    for p in list(nc.parameters()) + list(nf.parameters()):
                        p.mul_((1.0 + 1e-7 * torch.randn(p.shape, generator=gt)).to(dev))
            opt = npa.FlatAdam(list(nc.parameters()) + list(nf.parameters()), lr=5e-4, betas=(0.9, 0.999))
            g = torch.Generator(device="cpu").manual_seed(1000 + seed)
            at = []
            t0 = time.perf_counter()
            for it in range(1, steps + 1):
                idx = torch.randint(0, pool.shape[0], (n_batch,), generator=g).to(dev)
                npa.set_precision(prec)
                opt.zero_grad()
                out = npa.render_rays(pool[idx], nc, None, network_fine=nf, perturb=1.0, **rk)
                (npa.img2mse(out["rgb_map"], tgt_pool[idx]) + npa.img2mse(out["rgb0"], tgt_pool[idx])).backward()
                opt.step()
                if it in checkpoints:
                    at.append(round(psnr(nc, nf), 4))
            torch.cuda.synchronize()
            results[name] = at
            log(f"pair {pair_index} {list(CONVERGING_PAIRS[pair_index])} {name}: {at}  ({time.perf_counter() - t0:.0f} s)")
    return results


def family_stats(results, prec, k):
    vals = [v[k] for n, v in results.items() if n == prec or n.startswith(prec + "_twin")]
    return statistics.mean(vals), (statistics.stdev(vals) if len(vals) > 1 else 0.0), vals


def decide(per_pair, checkpoints, precisions):
    if "fp32" not in precisions:        # more runs of one family only (--twin-range): pooled and judged by tools/merge_equivalence.py
        return [], None
    table, keep = [], True
    for pi, results in per_pair.items():
        ratios = []
        for k, c in enumerate(checkpoints):
            row = {"pair": pi, "steps": c}
            for prec in precisions:
                m, sd, vals = family_stats(results, prec, k)
                row[prec] = {"mean_db": round(m, 4), "sd_db": round(sd, 4), "min_db": round(min(vals), 4), "max_db": round(max(vals), 4)}
            sd32 = max(row["fp32"]["sd_db"], 1e-4)
            for prec in precisions[1:]:
                row[prec]["mean_minus_fp32_db"] = round(row[prec]["mean_db"] - row["fp32"]["mean_db"], 4)
                row[prec]["sd_over_fp32_sd"] = round(row[prec]["sd_db"] / sd32, 3)
            table.append(row)
            if "fp16x3" in row:
                ratios.append((c, row["fp16x3"]["sd_over_fp32_sd"]))
        last = table[-1]
        if "fp16x3" in last:
            a_ok = abs(last["fp16x3"]["mean_minus_fp32_db"]) <= 2 * max(last["fp32"]["sd_db"], 1e-4)
            late = [r for c, r in ratios if c >= 2000] or [r for _c, r in ratios]
            b_ok = ratios[-1][1] <= 1.5 * statistics.median(late)
            last["rule"] = {"a_no_systematic_deficit": a_ok, "b_sd_ratio_not_growing": b_ok, "sd_ratio_by_checkpoint": ratios}
            keep = keep and a_ok and b_ok
    return table, keep


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=10000)
    ap.add_argument("--twins", type=int, default=4)
    ap.add_argument("--pairs", default="0")
    ap.add_argument("--rays", type=int, default=1024)
    ap.add_argument("--precisions", default="fp32,fp16x3,fp16x3w")
    ap.add_argument("--twin-range", default="", help="a:b = only the runs with perturbation numbers a .. b-1 (0 = the unperturbed run): more runs of the "
                                                     "same families from another process / call; merge with tools/merge_equivalence.py")
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "r06_equivalence.json"))
    args = ap.parse_args()
    dev = torch.device("cuda", 0)
    cps = [c for c in (250, 500, 1000, 2000, 3000, 5000, 7500, 10000, 15000, 20000, 30000, 50000) if c < args.steps] + [args.steps]
    precisions = args.precisions.split(",")
    os.makedirs(os.path.dirname(args.out), exist_ok=True)
    logf = open(args.out.replace(".json", ".log"), "a")

    def log(msg):
        print(msg, flush=True)
        logf.write(msg + "\n")
        logf.flush()
    per_pair = {}
    for pi in [int(x) for x in args.pairs.split(",")]:
        tr = tuple(int(x) for x in args.twin_range.split(":")) if args.twin_range else None
        per_pair[pi] = run_family(dev, pi, args.steps, cps, args.rays, precisions, args.twins, log, twin_range=tr)
        table, keep = decide(per_pair, cps, precisions)
        with open(args.out, "w") as f:          # (rewritten after every pair: a call cut short keeps what finished)
            json.dump({"steps": args.steps, "checkpoints": cps, "rays_per_step": args.rays, "runs_per_family": 1 + args.twins,
                       "pairs": {str(k): list(CONVERGING_PAIRS[k]) for k in per_pair}, "precisions": precisions,
                       "held_out_psnr_db": {str(k): v for k, v in per_pair.items()}, "family_table": table,
                       "decision": {"fp16x3_stays_default": keep, "rule": __doc__.split("decision rule evaluated on it:")[1].split("usage:")[0].strip()},
                       "what": __doc__.split("Output:")[0].strip()}, f, indent=1)
    for row in table:
        log(json.dumps(row))
    log(f"decision: fp16x3 stays the default = {keep}")


if __name__ == "__main__":
    main()
