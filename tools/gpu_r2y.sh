#!/bin/bash
mkdir -p gpurun_out; R=$(pwd)
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -p no:cacheprovider -k "bf16 or mixed or operand or golden" > gpurun_out/r2y_tests.log 2>&1; echo "pytest rc=$?"
grep -v "amdgpu.ids" gpurun_out/r2y_tests.log | tail -3
python tools/exp_fwd3.py - --bwd 2>&1 | grep -v amdgpu
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --pmc TCC_EA0_WRREQ_sum WRITE_SIZE -f csv -d $R/gpurun_out/wr2 -o wr -- python $R/tools/exp_fwd3.py - > $R/gpurun_out/wr2.log 2>&1
cd $R
python - <<'PY'
import csv, glob, collections
agg = collections.defaultdict(lambda: [0, 0.0])
for f in glob.glob("gpurun_out/wr2/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        if "fwd16" in row["Kernel_Name"] and row["Counter_Name"] == "WRITE_SIZE":
            k = row["Kernel_Name"][:48]; agg[k][0] += 1; agg[k][1] += float(row["Counter_Value"])
for k, (n, v) in sorted(agg.items()):
    print(k, n, "KiB/launch %.0f  B/pt %.0f" % (v / n, v / n * 1024 / 786432))
PY
python bench.py --steps 20 --warmup 5 --single-datapath --no-cpu-baseline --no-eager-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print(d['value'], d['ms_per_step'])
for k,v in d['kernels'].items(): print('   ',k,round(v['avg_ms'],3))"
