#!/bin/bash
mkdir -p gpurun_out; R=$(pwd)
cd /tmp && export TMPDIR=/tmp
for v in - libexp_f16_norows.so; do
  tag=$(echo $v | tr -d '.-'); tag=${tag:-base}
  timeout 300 rocprofv3 --kernel-trace --pmc TCC_EA0_WRREQ_sum WRITE_SIZE -f csv -d $R/gpurun_out/wr_$tag -o wr -- python $R/tools/exp_fwd3.py $v > $R/gpurun_out/wr_$tag.log 2>&1
done
cd $R
python - <<'PY'
import csv, glob, collections
for d in sorted(glob.glob("gpurun_out/wr_*/")):
    agg = collections.defaultdict(lambda: [0, 0.0])
    for f in glob.glob(d + "**/*counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(f)):
            if "fwd16" in row["Kernel_Name"] and row["Counter_Name"] == "WRITE_SIZE":
                k = row["Kernel_Name"][:48]; agg[k][0] += 1; agg[k][1] += float(row["Counter_Value"])
    for k, (n, v) in sorted(agg.items()):
        print(d, k, n, "KiB/launch %.0f  B/pt %.0f" % (v / n, v / n * 1024 / 786432))
PY
