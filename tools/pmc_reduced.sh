#!/bin/bash
# PMC counters of the inference forward, three-term fp16 vs reduced class (tools/exp_reduced.py as the rocprofv3 target):
# MFMA busy, LDS activity / bank conflicts, wait cycles.  Writes gpurun_out/pmc_reduced/summary.csv
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/pmc_reduced
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CMD="python $R/tools/exp_reduced.py"
for pass in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS" "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_SALU"; do
  tag=$(echo $pass | tr ' ' '_' | cut -c1-40)
  timeout 300 rocprofv3 --kernel-trace --pmc $pass -f csv -d $OUT/pmc_$tag -o r -- $CMD > $OUT/pmc_$tag.log 2>&1; echo "pmc $tag rc=$?"
done
cd $R
python - <<'PY'
import csv, glob, collections
out = "gpurun_out/pmc_reduced"
agg = collections.defaultdict(lambda: [0, 0.0])
for f in sorted(glob.glob(f"{out}/pmc_*/**/*counter_collection.csv", recursive=True)):
    for row in csv.DictReader(open(f)):
        kn = row["Kernel_Name"]
        if "field_fwd16r_kernel" not in kn:
            continue
        g = "x".join(row[k] for k in ("Grid_Size",) if k in row)
        k = (kn.replace("void ", "").split("(")[0], g, row["Counter_Name"])
        agg[k][0] += 1; agg[k][1] += float(row["Counter_Value"])
with open(f"{out}/summary.csv", "w") as fo:
    fo.write("kernel,grid,counter,dispatches,mean_per_dispatch\n")
    for (kn, g, cn), (n, v) in sorted(agg.items()):
        fo.write(f"{kn},{g},{cn},{n},{v / n:.6g}\n")
print(open(f"{out}/summary.csv").read())
PY
