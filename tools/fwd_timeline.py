#!/usr/bin/env python3
"""Where the persistent prefill attention kernel spends its time: in-kernel stamps (make EXPERIMENTS=1,
hx_debug_fwd_stamps) of wave 0 of every workgroup, summarised per event pair.

events: 1 kernel entry, 2 first requests issued + next item decoded, 3 first tile landed, 10 tile step begins,
11 its arithmetic is issued, 12 next tile landed (then the barrier), 20 last tile: next Q landed, 21 seam barrier passed,
22 O rows stored, 23 next-next item decoded, 24 first tile of the next item landed (then the barrier)."""
import math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from hydrainfer_amd import _lib
from hydrainfer_amd._C.kernel.flash_attn import mha_varlen_fwd

B, n, kv = (int(x) for x in (sys.argv[1:4] if len(sys.argv) > 3 else (4, 704, 704)))
dev, dt = torch.device("cuda:0"), torch.bfloat16
g = torch.Generator(device=dev).manual_seed(0)
rnd = lambda *s: torch.randn(s, generator=g, device=dev, dtype=torch.float32).to(dt)
H, D, bs = 32, 128, 16
nb = (kv + bs - 1) // bs
kc, vc, q = rnd(B * nb, bs, H, D), rnd(B * nb, bs, H, D), rnd(B * n, H, D)
out = torch.empty_like(q)
perm = torch.randperm(B * nb, generator=g, device=dev).to(torch.int32)
cu_b = torch.arange(0, (B + 1) * nb, nb, dtype=torch.int32, device=dev)
cu_q = torch.arange(0, (B + 1) * n, n, dtype=torch.int32, device=dev)
cu_k = torch.arange(0, (B + 1) * kv, kv, dtype=torch.int32, device=dev)
fn = lambda: mha_varlen_fwd(out, q, kc, vc, cu_q, cu_k, perm, cu_b, None, n, kv, 1 / math.sqrt(D), 0, -1, 0, 0)
l = _lib.lib()
assert _lib.has_experiments(), "build with make -C hydrainfer_amd/csrc EXPERIMENTS=1"
for _ in range(5):
    fn()
buf = torch.zeros(512 * 512, dtype=torch.int64, device=dev)
l.hx_debug_fwd_stamps(buf.data_ptr())
fn(); torch.cuda.synchronize()
buf.zero_()
fn(); torch.cuda.synchronize()
l.hx_debug_fwd_stamps(None)
a = buf.cpu().numpy().reshape(512, 512)
t0 = min(int(r[0]) >> 8 for r in a if r[0])
ends, pairs, per_wg = [], {}, []
clocks, place = [], {}
for wg, r in enumerate(a):
    raw = [(int(x) & 255, (int(x) >> 8) & ((1 << 55) - 1)) for x in r if x]
    cyc = dict((e, x) for e, x in raw if e in (30, 31))
    for e, x in raw:
        if e == 32:      # (XCC, SE, SH, CU) of the workgroup
            hw = x & 0xffffffff
            place[wg] = ((x >> 32) & 15, (hw >> 13) & 7, (hw >> 12) & 1, (hw >> 8) & 15)
    ev = [(e, (x - t0) / 100.0) for e, x in raw if e not in (30, 31, 32)]
    if not ev:
        continue
    if 30 in cyc and 31 in cyc:      # shader-clock cycles per 100 MHz tick between kernel entry and the last store
        clocks.append((cyc[31] - cyc[30]) / max(1e-9, (ev[-1][1] - ev[0][1])) / 1e3)
    ends.append(ev[-1][1])
    for (e0, x0), (e1, x1) in zip(ev, ev[1:]):
        pairs.setdefault((e0, e1), []).append(x1 - x0)
    per_wg.append((wg, ev))
print(f"{B} x {n} of {kv}: {len(per_wg)} workgroups, last stamp at {max(ends):.2f} us, median {np.median(ends):.2f}, first {min(ends):.2f}")
if place:
    by_cu = {}
    for wg, pl in sorted(place.items()):
        by_cu.setdefault(pl, []).append(wg)
    print(f"{len(by_cu)} CUs host the {len(place)} workgroups; workgroups per CU: "
          f"{sorted(set(len(v) for v in by_cu.values()))}; first CUs: " + "; ".join(f"{k}: {v}" for k, v in list(sorted(by_cu.items()))[:12]))
    d = sorted(set(v[1] - v[0] for v in by_cu.values() if len(v) == 2))
    print("workgroup id distance between the two workgroups of a CU:", d[:20])
if clocks:
    print(f"shader clock over the workgroups' lifetimes: median {np.median(clocks):.2f} GHz (min {min(clocks):.2f}, max {max(clocks):.2f})")
tot = sum(sum(v) for v in pairs.values())
for k, v in sorted(pairs.items(), key=lambda kv: -sum(kv[1])):
    print(f"  {k[0]:3d} -> {k[1]:3d}: n {len(v):6d}  mean {np.mean(v):6.2f} us  p90 {np.percentile(v, 90):6.2f}  share {100 * sum(v) / tot:5.1f} %")
for wg, ev in per_wg[:: max(1, len(per_wg) // 4)][:4]:
    print(f"wg {wg}: " + " ".join(f"{e}@{x:.2f}" for e, x in ev[:60]))
# end time and step count of every workgroup, sorted by end time (who is the tail?)
rows = []
for wg, ev in per_wg:
    steps = sum(1 for e, _ in ev if e == 10)
    solo = [x1 - x0 for (e0, x0), (e1, x1) in zip(ev, ev[1:]) if e0 == 10 and e1 == 11]
    rows.append((ev[-1][1], wg, steps, np.mean(solo[: len(solo) // 2]) if solo else 0, np.mean(solo[len(solo) // 2:]) if solo else 0))
rows.sort()
print("end us, wg, steps, mean 10->11 first half, second half:")
for r in rows[:: max(1, len(rows) // 16)] + rows[-3:]:
    print(f"  {r[0]:7.2f}  wg {r[1]:4d}  steps {r[2]:3d}  {r[3]:.2f}  {r[4]:.2f}")
