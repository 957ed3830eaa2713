#!/usr/bin/env python3
"""Mean duration per kernel name from a rocprofv3 --kernel-trace CSV directory (last N launches of each kernel)."""
import csv, glob, os, sys, collections
csv.field_size_limit(1 << 30)
d = sys.argv[1]
last = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
f = sorted(glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True))[0]
per = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    per[r["Kernel_Name"]].append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]) - int(r["Start_Timestamp"])))
rows = []
for k, v in per.items():
    v.sort()
    v = v[-last:]
    rows.append((sum(x[1] for x in v) / len(v) / 1e3, len(v), k))
rows.sort(reverse=True)
for us, n, k in rows[:12]:
    print(f"{us:9.2f} us x {n:5d}  {k[:110]}")
