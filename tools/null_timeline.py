#!/usr/bin/env python3
"""Per-launch times of the null step next to the built step from ONE rocprofv3 --kernel-trace run of tools/null_layer.py
(rocpd sqlite output):  null_timeline.py <results.db> [out.md]
Steps are the kernel runs between two decode_step_head launches; a step with paged_read launches is a null step."""
import collections, sqlite3, statistics, sys


def short(n):
    for k in ("paged_read", "read_stream", "attn_decode", "gemm_packed", "decode_step_head", "argmax", "add_rms_norm_slab", "Cijk"):
        if k in n:
            return k
    if "gemm_xreg" in n:
        i = n.index("gemm_xreg")
        return n[i:i + 48].split(">")[0] + ">"
    return n[:40]


def main():
    c = sqlite3.connect(sys.argv[1])
    rows = list(c.execute("select name, start, end, grid_x, grid_y from kernels order by start"))
    names = [short(r[0]) for r in rows]
    idx = [i for i, n in enumerate(names) if n == "decode_step_head"]
    segs = [(a, b) for a, b in zip(idx, idx[1:]) if 100 < b - a < 400]
    out = []
    for label, pick in (("null step", lambda a, b: "paged_read" in names[a:b]), ("built step", lambda a, b: "paged_read" not in names[a:b])):
        ss = [s for s in segs if pick(*s)]
        if not ss:
            continue
        n = statistics.mode([b - a for a, b in ss])
        ss = [s for s in ss if s[1] - s[0] == n]
        a0 = ss[0][0]
        per = collections.OrderedDict()
        for j in range(n):
            k = f"{names[a0 + j]} grid {rows[a0 + j][3] // 256}x{rows[a0 + j][4]}"
            d = statistics.median((rows[a + j][2] - rows[a + j][1]) / 1e3 for a, _ in ss)
            g = statistics.median((rows[a + j][1] - rows[a + j - 1][2]) / 1e3 for a, _ in ss) if j else 0.0
            per.setdefault(k, []).append((d, g))
        span = statistics.median((rows[b - 1][2] - rows[a][1]) / 1e3 for a, b in ss)
        out.append(f"## {label}: {len(ss)} steps of {n} launches, median span {span:.1f} us (first launch start to last launch end)\n")
        out.append("| launch | per step | mean us | mean gap before us | total us |\n|---|---|---|---|---|")
        for k, v in per.items():
            out.append(f"| `{k}` | {len(v)} | {statistics.mean(x[0] for x in v):.2f} | {statistics.mean(x[1] for x in v):.2f} | {sum(x[0] + x[1] for x in v):.1f} |")
        # the five launches of a middle layer, by position (null steps: read_stream launches differ only in bytes)
        if label == "null step":
            first = [j for j in range(n) if names[a0 + j] == "paged_read"][0]
            out.append("\nby position in a layer (layers 1..30): " + "; ".join(
                f"{['paged read', 'o', 'gate|up', 'down', 'qkv'][k]} {statistics.median((rows[a + first + 5 * l + k][2] - rows[a + first + 5 * l + k][1]) / 1e3 for a, _ in ss for l in range(1, 31)):.2f}"
                for k in range(5)) + " us")
        out.append("")
    text = "\n".join(out)
    print(text)
    if len(sys.argv) > 2:
        open(sys.argv[2], "w").write(text + "\n")


main()
