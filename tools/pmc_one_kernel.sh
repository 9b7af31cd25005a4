#!/bin/bash
# PMC counters of one field kernel of one library: tools/pmc_one_kernel.sh <lib.so|-> <fwd|fwdsave|dgrad> <tag>
R=${GRAFT_REPO_ROOT:-$(pwd)}; LIB=$1; WHAT=$2; TAG=$3
OUT=$R/gpurun_out/pmc_$TAG; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for pass in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS SQ_LDS_IDX_ACTIVE"; do
  tag=$(echo $pass | tr ' ' '_' | cut -c1-30)
  timeout 300 rocprofv3 --kernel-trace --pmc $pass -f csv -d $OUT/$tag -o k -- python $R/tools/exp_one_kernel.py $LIB $WHAT 6 > $OUT/$tag.log 2>&1
done
cd $R
python - $OUT <<'PY'
import csv, glob, collections, sys
agg = collections.defaultdict(lambda: [0, 0.0])
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        if "field_" not in row["Kernel_Name"]: continue
        k = (row["Kernel_Name"].replace("void ", "").split("(")[0], row["Counter_Name"]); agg[k][0] += 1; agg[k][1] += float(row["Counter_Value"])
for (kn, cn), (n, v) in sorted(agg.items()): print(f"{kn},{cn},{n},{v / n:.6g}")
PY
