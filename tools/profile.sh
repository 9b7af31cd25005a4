#!/bin/bash
# rocprofv3 evidence for bench.py: kernel stats (timing) + separate PMC passes (HBM bytes, MFMA busy, LDS).
# Writes under gpurun_out/prof_*; summaries are copied into profiles/ by hand afterwards.
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
CMD="python $R/bench.py --steps 4 --warmup 2 --no-cpu-baseline --single-datapath"
{
echo "== kernel stats"
timeout 600 rocprofv3 --kernel-trace --stats -f csv -d $R/gpurun_out/prof_stats -o bench -- $CMD 2>&1 | grep -E '^\{|rror' | cut -c1-600
for pass in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_F32" "TCC_HIT_sum TCC_MISS_sum"; do
  tag=$(echo $pass | tr ' ' '_' | cut -c1-40)
  echo "== pmc $pass"
  timeout 600 rocprofv3 --kernel-trace --pmc $pass -f csv -d $R/gpurun_out/prof_pmc_$tag -o bench -- $CMD 2>&1 | grep -E 'rror|nvalid' | head -5
done
cd $R
python - <<'PY'
import csv, glob, collections, os
for f in sorted(glob.glob("gpurun_out/prof_stats/**/*kernel_stats.csv", recursive=True)):
    print("##", f); print(open(f).read()[:3000])
for f in sorted(glob.glob("gpurun_out/prof_pmc_*/**/*counter_collection.csv", recursive=True)):
    agg = collections.defaultdict(lambda: [0, 0.0])
    for row in csv.DictReader(open(f)):
        k = (row["Kernel_Name"].split("(")[0][:60], row["Counter_Name"])
        agg[k][0] += 1; agg[k][1] += float(row["Counter_Value"])
    print("##", f)
    for (kn, cn), (n, v) in sorted(agg.items()):
        if any(s in kn for s in ("field_", "wgrad", "expand")):
            print(f"{kn:62s} {cn:28s} dispatches={n:4d} mean={v/n:.6g}")
PY
} > $R/gpurun_out/profile.log 2>&1
tail -150 $R/gpurun_out/profile.log
