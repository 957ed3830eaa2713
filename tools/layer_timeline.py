#!/usr/bin/env python3
"""Per-layer timeline of the decode graph from a rocprofv3 --kernel-trace CSV:
    layer_timeline.py <dir> [out.md]
Finds the steady-state decode steps (runs of kernels between two argmax/lm_head launches),
and prints for the median step: every kernel of one middle layer with its duration and the
gap to its predecessor (end -> start), plus per-kernel-name totals over the whole step
(busy time, gap time).  This is what shows where a step's time goes BETWEEN kernels."""
import csv, glob, os, statistics, sys

csv.field_size_limit(1 << 30)


def short(name):
    n = name.replace("void (anonymous namespace)::", "").replace("(anonymous namespace)::", "")
    n = n.replace("hx::", "")
    return n[:70]


def main():
    d = sys.argv[1]
    f = sorted(glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True))[0]
    rows = []
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
    rows.sort()
    # a decode step starts at its step-head launch (decode_advance_kernel before round 4)
    starts = [i for i, r in enumerate(rows) if "decode_advance" in r[2] or "decode_step_head" in r[2]]
    steps = []
    for a, b in zip(starts, starts[1:]):
        seg = rows[a:b]
        if len(seg) < 50 or len(seg) > 2000:
            continue
        if any("paged_read_kernel" in r[2] or "read_stream_kernel" in r[2] for r in seg):
            continue      # a null step of bench.py's whole_step.null_step (math-free stand-in launches): not a decode step
        must = os.environ.get("TIMELINE_MUST_CONTAIN")      # e.g. gemm_xreg_wide: the 64-row steps of bench.py's whole_step_64
        if must and not any(must in r[2] for r in seg):
            continue
        if not must and any("gemm_xreg_wide" in r[2] for r in seg):
            continue      # (a 64-row step has had the 32-row step's launch count since the wide layer went to 5 launches)
        span = seg[-1][1] - seg[0][0]
        steps.append((span, a, b))
    if not steps:
        print("no decode steps found"); return
    lens = int(os.environ.get("TIMELINE_KERNELS") or statistics.mode([b - a for _, a, b in steps]))   # (196: bench.py's bare 64-row step)
    steps = [s for s in steps if s[2] - s[1] == lens]
    steps.sort()
    span, a, b = steps[len(steps) // 2]
    seg = rows[a:b]
    out = []
    out.append(f"decode steps found: {len(steps)} with {lens} kernels; median span {span / 1e3:.1f} us "
               f"(min {steps[0][0] / 1e3:.1f}, max {steps[-1][0] / 1e3:.1f})\n")
    busy, gaps, calls = {}, {}, {}
    prev_end = None
    for s, e, n in seg:
        k = short(n)
        busy[k] = busy.get(k, 0) + (e - s)
        calls[k] = calls.get(k, 0) + 1
        if prev_end is not None:
            gaps[k] = gaps.get(k, 0) + (s - prev_end)
        prev_end = e
    out.append("| kernel | calls | busy us (avg) | total busy us | gap before us (avg) |\n|---|---|---|---|---|")
    tot_b = tot_g = 0
    for k in sorted(busy, key=lambda k: -busy[k]):
        out.append(f"| `{k}` | {calls[k]} | {busy[k] / calls[k] / 1e3:.2f} | {busy[k] / 1e3:.1f} | "
                   f"{gaps.get(k, 0) / calls[k] / 1e3:.2f} |")
        tot_b += busy[k]; tot_g += gaps.get(k, 0)
    out.append(f"\nstep: busy {tot_b / 1e3:.1f} us + gaps {tot_g / 1e3:.1f} us = {(tot_b + tot_g) / 1e3:.1f} us\n")
    big = [(seg[i][0] - seg[i - 1][1], i, short(seg[i][2])) for i in range(1, len(seg)) if seg[i][0] - seg[i - 1][1] > 500]
    if big:
        out.append("gaps over 0.5 us in this step (us, kernel index in the step, kernel that waited): "
                   + "; ".join(f"{g / 1e3:.1f} @{i} {n[:28]}" for g, i, n in big[:60]) + "\n")
    # one middle layer: kernels between the 16th and 17th attention launch
    att = [i for i, r in enumerate(seg) if "attn_decode" in r[2] or "layer_chain" in r[2]]
    if len(att) > 17:
        i0, i1 = att[15], att[16]
        out.append("one middle layer (start offsets relative to its attention launch):\n")
        out.append("| kernel | start us | dur us | gap before us |\n|---|---|---|---|")
        t0 = seg[i0][0]
        for i in range(i0, i1):
            s, e, n = seg[i]
            out.append(f"| `{short(n)}` | {(s - t0) / 1e3:.2f} | {(e - s) / 1e3:.2f} | {(s - seg[i - 1][1]) / 1e3:.2f} |")
        out.append(f"\nlayer period: {(seg[i1][0] - t0) / 1e3:.2f} us")
    txt = "\n".join(out)
    print(txt)
    if len(sys.argv) > 2:
        open(sys.argv[2], "w").write(txt + "\n")
    if len(sys.argv) > 3:      # machine-readable: what bench.py's roofline.in_step_us quotes (profiles/r*_bench7b_in_step.json)
        import json
        att_k = [k for k in busy if "attn_decode_kernel" in k]
        if att_k:
            k = max(att_k, key=lambda k: busy[k])
            json.dump({"attention_kernel": k, "attention_us": round(busy[k] / calls[k] / 1e3, 2), "calls_per_step": calls[k],
                       "steps": len(steps), "step_span_us": round(span / 1e3, 1),
                       "source": "rocprofv3 --kernel-trace of bench.py --steps 64 (tools/prof_step.sh), median decode step, mean over its attention launches"},
                      open(sys.argv[3], "w"), indent=1)


if __name__ == "__main__":
    main()
