"""Pool the runs of several tools/exp_equivalence_long.py outputs (same pair, same step count, disjoint perturbation numbers) into one
family table: per checkpoint and datapath n, mean, sd, and for the non-fp32 families the difference of the means to fp32 with its
standard error (Welch) -- the well-powered form of the decision rule's clause (a).

usage: python tools/merge_equivalence.py --pair 0 --out profiles/r06_training_equivalence_10k_pooled.json file1.json file2.json ..."""
import argparse
import json
import math
import statistics


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--pair", default="0")
    ap.add_argument("--out", default="")
    ap.add_argument("files", nargs="+")
    args = ap.parse_args()
    runs, cps, steps = {}, None, None
    for f in args.files:
        d = json.load(open(f))
        if args.pair not in d["held_out_psnr_db"]:
            continue
        if cps is None:
            cps, steps = d["checkpoints"], d["steps"]
        assert d["checkpoints"] == cps and d["steps"] == steps, f"{f}: other checkpoints / step count"
        for name, v in d["held_out_psnr_db"][args.pair].items():
            if name in runs:
                assert runs[name] == v, f"{f}: run {name} differs from an earlier file (runs are bit-reproducible unless a kernel changed)"
            runs[name] = v
    fam = lambda prec: {n: v for n, v in runs.items() if n == prec or n.startswith(prec + "_twin")}
    precisions = [p for p in ("fp32", "fp16x3", "fp16x3w") if fam(p)]
    table = []
    for k, c in enumerate(cps):
        row = {"steps": c}
        for p in precisions:
            vals = [v[k] for v in fam(p).values()]
            row[p] = {"n": len(vals), "mean_db": round(statistics.mean(vals), 4), "sd_db": round(statistics.stdev(vals), 4),
                      "min_db": min(vals), "max_db": max(vals)}
        for p in precisions[1:]:
            a, b = row[p], row["fp32"]
            se = math.sqrt(a["sd_db"] ** 2 / a["n"] + b["sd_db"] ** 2 / b["n"])
            a["mean_minus_fp32_db"] = round(a["mean_db"] - b["mean_db"], 4)
            a["standard_error_db"] = round(se, 4)
            a["z"] = round((a["mean_db"] - b["mean_db"]) / se, 2) if se > 0 else None
            a["sd_over_fp32_sd"] = round(a["sd_db"] / max(b["sd_db"], 1e-4), 3)
        table.append(row)
        print(json.dumps(row))
    rule = None
    if "fp16x3" in precisions:
        last = table[-1]
        ratios = [(r["steps"], r["fp16x3"]["sd_over_fp32_sd"]) for r in table]
        late = [x for c, x in ratios if c >= 2000] or [x for _c, x in ratios]
        rule = {"a_no_systematic_deficit": abs(last["fp16x3"]["mean_minus_fp32_db"]) <= 2 * last["fp32"]["sd_db"],
                "b_sd_ratio_not_growing": ratios[-1][1] <= 1.5 * statistics.median(late),
                "sd_ratio_by_checkpoint": ratios, "median_sd_ratio_from_2000_steps": statistics.median(late),
                "what": "the decision rule of tools/exp_equivalence_long.py evaluated on the POOLED families"}
        rule["fp16x3_stays_default"] = rule["a_no_systematic_deficit"] and rule["b_sd_ratio_not_growing"]
        print(json.dumps(rule))
    out = {"pair": args.pair, "decision": rule, "steps": steps, "checkpoints": cps, "files": args.files, "runs": {p: sorted(fam(p)) for p in precisions},
           "family_table": table, "held_out_psnr_db": runs,
           "what": "pooled families of tools/exp_equivalence_long.py runs (1024 rays per step, fused Adam lr 5e-4, held-out PSNR of 2048 rays on the "
                   "fp32 datapath); mean_minus_fp32_db with its Welch standard error: |z| < 2 = no detectable difference of the family means"}
    if args.out:
        json.dump(out, open(args.out, "w"), indent=1)


if __name__ == "__main__":
    main()
