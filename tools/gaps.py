"""Idle time between kernels of a training step from a rocprofv3 --kernel-trace CSV: per pair (previous kernel -> next kernel) the
mean gap, and the step's totals (busy, idle).  usage: python tools/gaps.py <..._kernel_trace.csv> [first_kernel_substring]"""
import collections
import csv
import sys

rows = []
for r in csv.DictReader(open(sys.argv[1])):
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].replace("void ", "").split("(")[0].replace("nerf::", "")))
rows.sort()
marker = sys.argv[2] if len(sys.argv) > 2 else "adam_kernel"
# steps end with the optimizer's last launch: split after every SECOND adam launch
ends = [i for i, r in enumerate(rows) if marker in r[2]]
steps = []
for a, b in zip(ends[1::2], ends[3::2]):
    steps.append(rows[a + 1:b + 1])
steps = [s for s in steps if any("field_dgrad3r" in k[2] for k in s)]
print(f"{len(steps)} steps")
gaps = collections.defaultdict(list)
busy, idle, wall = [], [], []
for s in steps[2:]:
    b = sum(e - st for st, e, _ in s)
    w = s[-1][1] - s[0][0]
    busy.append(b); wall.append(w); idle.append(w - b)
    for (s0, e0, k0), (s1, e1, k1) in zip(s, s[1:]):
        gaps[(k0[:44], k1[:44])].append(s1 - e0)
n = max(1, len(busy))
print(f"per step: wall {sum(wall) / n / 1e3:.1f} us, kernels {sum(busy) / n / 1e3:.1f} us, idle between kernels {sum(idle) / n / 1e3:.1f} us, launches {len(steps[-1]) if steps else 0}")
small = collections.defaultdict(list)
for s in steps[2:]:
    per = collections.defaultdict(int)
    for st, e, k in s:
        per[k] += e - st
    for k, v in per.items():
        small[k].append(v)
print("kernel time per step (us):")
for k, v in sorted(small.items(), key=lambda kv: -sum(kv[1])):
    print(f"  {sum(v) / len(v) / 1e3:9.1f}  {k[:100]}")
print("gaps (us) by (previous -> next):")
for k, v in sorted(gaps.items(), key=lambda kv: -sum(kv[1]) / len(kv[1])):
    print(f"  {sum(v) / len(v) / 1e3:7.2f}  {k[0]} -> {k[1]}")
