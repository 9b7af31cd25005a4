#!/bin/bash
mkdir -p gpurun_out; R=$(pwd)
python tools/probe/coal_probe.py 2>&1 | grep -v amdgpu
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --pmc TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum WRITE_SIZE -f csv -d $R/gpurun_out/coal -o coal -- python $R/tools/probe/coal_probe.py > $R/gpurun_out/coal.log 2>&1
cd $R
python - <<'PY'
import csv, glob, collections
agg = collections.defaultdict(lambda: [0, 0.0])
for f in glob.glob("gpurun_out/coal/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        if "coal_k" in row["Kernel_Name"]:
            k = (row["Kernel_Name"][:40], row["Counter_Name"]); agg[k][0] += 1; agg[k][1] += float(row["Counter_Value"])
for k, (n, v) in sorted(agg.items()):
    print(k, n, v / n)
PY
