#!/usr/bin/env python3
"""Condenses rocprofv3 output directories into the small files kept under profiles/:
    summarize_prof.py stats <dir> <out.md> "<title>"        top kernels of a --kernel-trace --stats run
    summarize_prof.py pmc <dir> <kernel substring> <out.json>  per-counter medians for one kernel"""
import csv, glob, json, os, statistics, sys

csv.field_size_limit(1 << 30)


def stats(d, out, title, top=int(os.environ.get("HX_TOP", "16"))):
    f = sorted(glob.glob(os.path.join(d, "**", "*kernel_stats.csv"), recursive=True))[0]
    rows = list(csv.DictReader(open(f)))
    with open(out, "w") as o:
        o.write(f"# {title}\n\n| kernel | calls | avg us | total ms | % |\n|---|---|---|---|---|\n")
        for r in rows[:top]:
            o.write(f"| `{r['Name'][:110]}` | {r['Calls']} | {float(r['AverageNs']) / 1e3:.2f} | "
                    f"{float(r['TotalDurationNs']) / 1e6:.2f} | {r['Percentage']} |\n")
    print(open(out).read())


def pmc(d, needle, out):
    vals, durs = {}, []
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if needle in r["Kernel_Name"]:
                vals.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
                durs.append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    res = {"kernel_contains": needle, "launches": max((len(v) for v in vals.values()), default=0),
           "median_duration_ns_under_pmc": statistics.median(durs) if durs else None,
           "counters_median": {k: statistics.median(v) for k, v in vals.items()}}
    json.dump(res, open(out, "w"), indent=1)
    print(json.dumps(res))


if __name__ == "__main__":
    if sys.argv[1] == "stats":
        stats(sys.argv[2], sys.argv[3], sys.argv[4])
    else:
        pmc(sys.argv[2], sys.argv[3], sys.argv[4])
