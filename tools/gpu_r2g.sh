#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/gpurun_out/r2g
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L 2>/dev/null | grep -oE "^\s*(SQ_|TCC_|TCP_|TA_|TD_|GRBM_)[A-Za-z0-9_]+" | sort -u | tr -d ' ' > $R/gpurun_out/r2g/counters.txt
wc -l $R/gpurun_out/r2g/counters.txt
CMD="python $R/tools/exp_fwd16_only.py"
for pass in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM" "SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SALU SQ_VALU_MFMA_BUSY_CYCLES" "TCP_PENDING_STALL_CYCLES_sum TCP_TCC_WRITE_REQ_sum TCP_TCC_READ_REQ_sum TCP_GATE_EN1_sum" "TCC_EA0_WRREQ_STALL_sum TCC_EA0_WRREQ_sum TCC_TAG_STALL_sum TCC_BUSY_sum"; do
  tag=$(echo $pass | tr ' ' '_' | cut -c1-30)
  timeout 300 rocprofv3 --kernel-trace --pmc $pass -f csv -d $R/gpurun_out/r2g/pmc_$tag -o x -- $CMD > $R/gpurun_out/r2g/log_$tag.txt 2>&1
  echo "pass $tag rc=$?"
done
timeout 300 rocprofv3 --kernel-trace --stats -f csv -d $R/gpurun_out/r2g/stats -o x -- $CMD > /dev/null 2>&1
cd $R
python - <<'PY'
import csv, glob, collections
for f in sorted(glob.glob("gpurun_out/r2g/pmc_*/**/*counter_collection.csv", recursive=True)):
    agg = collections.defaultdict(lambda: [0, 0.0])
    for row in csv.DictReader(open(f)):
        k = (row["Kernel_Name"].split("(")[0][-40:], row["Counter_Name"])
        agg[k][0] += 1; agg[k][1] += float(row["Counter_Value"])
    for (kn, cn), (n, v) in sorted(agg.items()):
        if "field_" in kn: print(f"{kn:42s} {cn:32s} n={n:3d} mean={v/n:.5g}")
for f in glob.glob("gpurun_out/r2g/stats/**/*kernel_stats.csv", recursive=True):
    print(open(f).read()[:1500])
PY
