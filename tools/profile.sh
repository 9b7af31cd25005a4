#!/bin/bash
# rocprofv3 evidence for bench.py (one datapath per call): kernel stats (timing) + separate PMC passes (HBM bytes, MFMA
# busy, LDS, waits).  usage: tools/profile.sh <fp16x3|bf16x3|fp32> [extra bench flags]
# Writes gpurun_out/prof_<prec>/ and the two summaries profiles/ expects:
#   gpurun_out/prof_<prec>/kernel_stats.csv, gpurun_out/prof_<prec>/pmc_summary.csv
PREC=${1:-fp16x3}; shift
TAG=${TAG:-$PREC}           # directory tag: TAG=bf16x3_render_only tools/profile.sh bf16x3 --mode render_only --steps 2 --warmup 1
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CMD="python $R/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-eager-baseline --no-gate --single-datapath --no-configs --precision $PREC $@"
echo "# $CMD" > $OUT/command.txt
timeout 600 rocprofv3 --kernel-trace --stats -f csv -d $OUT/stats -o bench -- $CMD > $OUT/stats.log 2>&1; echo "stats rc=$?"
for pass in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_F32" "TCC_HIT_sum TCC_MISS_sum TCC_EA0_WRREQ_sum"; do
  tag=$(echo $pass | tr ' ' '_' | cut -c1-40)
  timeout 600 rocprofv3 --kernel-trace --pmc $pass -f csv -d $OUT/pmc_$tag -o bench -- $CMD > $OUT/pmc_$tag.log 2>&1; echo "pmc $tag rc=$?"
done
cd $R
python - "$TAG" "$CMD" <<'PY'
import csv, glob, collections, sys, os
prec, cmd = sys.argv[1], sys.argv[2]
out = f"gpurun_out/prof_{prec}"
for f in glob.glob(f"{out}/stats/**/*kernel_stats.csv", recursive=True):
    rows = open(f).read().splitlines()
    keep = [rows[0]] + [r for r in rows[1:] if "nerf::" in r][:24]
    open(f"{out}/kernel_stats.csv", "w").write("\n".join(keep) + "\n")
    print("\n".join(keep[:12]))
# per launch size: the coarse (262,144 points) and the fine (786,432 points) launch of a kernel are different rows
by = collections.defaultdict(list)
for f in glob.glob(f"{out}/stats/**/*kernel_trace.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        if "nerf::" in row["Kernel_Name"]:
            k = (row["Kernel_Name"].replace("void ", "").split("(")[0], int(row["Grid_Size_X"]) // max(1, int(row["Workgroup_Size_X"])),
                 row["Workgroup_Size_X"], row["VGPR_Count"], row["Accum_VGPR_Count"], row["LDS_Block_Size"])
            by[k].append(int(row["End_Timestamp"]) - int(row["Start_Timestamp"]))
with open(f"{out}/kernel_by_launch.csv", "w") as fo:
    fo.write(f"# rocprofv3 --kernel-trace -- {cmd.replace(os.getcwd() + '/', '')}: every nerf:: kernel by launch size (workgroups)\n")
    fo.write("kernel,workgroups,workgroup_size,vgprs,agprs,lds_bytes,dispatches,mean_ns,min_ns,max_ns\n")
    for k, v in sorted(by.items(), key=lambda kv: -sum(kv[1])):
        fo.write(",".join(str(x) for x in k) + f",{len(v)},{sum(v) / len(v):.0f},{min(v)},{max(v)}\n")
agg = collections.defaultdict(lambda: [0, 0.0])
for f in sorted(glob.glob(f"{out}/pmc_*/**/*counter_collection.csv", recursive=True)):
    for row in csv.DictReader(open(f)):
        kn = row["Kernel_Name"]
        if "nerf::" not in kn:
            continue
        k = (kn.replace("void ", "").split("(")[0], row["Counter_Name"])
        agg[k][0] += 1; agg[k][1] += float(row["Counter_Value"])
with open(f"{out}/pmc_summary.csv", "w") as fo:
    fo.write(f"# rocprofv3 --kernel-trace --pmc <one group per pass> -- {cmd.replace(os.getcwd() + '/', '')}\n")
    fo.write("# mean per dispatch over the coarse (262,144 pts) and fine (786,432 pts) launches of every step; FETCH_SIZE / WRITE_SIZE in KiB;\n")
    fo.write("# gfx950: FETCH_SIZE under-reports wide coalesced reads by 2x (MI355X_MICROARCH.md, HBM section); TCC_EA0_WRREQ = 64-byte write requests\n")
    fo.write("kernel,counter,dispatches,mean_per_dispatch\n")
    for (kn, cn), (n, v) in sorted(agg.items()):
        fo.write(f"{kn},{cn},{n},{v / n:.6g}\n")
print(open(f"{out}/pmc_summary.csv").read()[:200])
PY
