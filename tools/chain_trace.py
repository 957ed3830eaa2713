#!/usr/bin/env python3
"""Per-item timeline of ONE decode-chain launch (debug option chain_trace): for every phase the
time its first item took a ticket, the time its dependency was seen, when its last item ended, and
how long items took — shows where a chain launch waits.
    python tools/chain_trace.py [7b|13b] [M]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from hydrainfer_amd import _lib
from hydrainfer_amd._C.kernel import gemm

dev, dt = torch.device("cuda:0"), torch.bfloat16
model = sys.argv[1] if len(sys.argv) > 1 else "7b"
M = int(sys.argv[2]) if len(sys.argv) > 2 else 32
hid, inter = (4096, 11008) if model == "7b" else (5120, 13824)
q_size, qkv_n = hid, 3 * hid
g = torch.Generator(device=dev).manual_seed(0)
rnd = lambda *s, sc=1.0: (torch.randn(s, device=dev, generator=g) * sc).to(dt)
W = [dict(o=rnd(hid, q_size, sc=.02), gu=rnd(2 * inter, hid, sc=.02), dn=rnd(hid, inter, sc=.02),
          qkv=rnd(qkv_n, hid, sc=.02), n1=rnd(hid), n2=rnd(hid)) for _ in range(3)]
for w in W:
    w.update(po=gemm.pack_weight(w["o"]), pgu=gemm.pack_weight(w["gu"]), pdn=gemm.pack_weight(w["dn"]), pqkv=gemm.pack_weight(w["qkv"]))
attn, h0 = rnd(M, q_size), rnd(M, hid)
need = gemm.chain_workspace_floats(M, hid, inter, q_size)
MAX_ITEMS = 8192
cws = torch.zeros(need + MAX_ITEMS * 8, dtype=torch.float32, device=dev)
qkvp = torch.empty(gemm.workspace_floats(M, qkv_n, hid), dtype=torch.float32, device=dev)
hm, ho, xp, xn = (torch.empty_like(h0) for _ in range(4))
actb = torch.empty((M, inter), dtype=dt, device=dev)
sync = torch.zeros(gemm.SYNC_WORDS, dtype=torch.int32, device=dev)
assert _lib.lib().hx_debug_set_option(b"chain_trace", 1) == 0
for it in range(3):   # last one is the one read back (weights cold: other sets in between)
    w = W[it]
    sync.zero_()
    gemm.decode_chain(attn, h0, w["po"], w["pgu"], w["pdn"], w["pqkv"], inter, w["n1"], w["n2"], 1e-5, hm, ho, xp, actb, xn,
                      qkvp, cws, sync)
torch.cuda.synchronize()
tr = cws[need:].view(torch.int64).cpu().numpy().reshape(-1, 4)
n = int((tr[:, 0] != 0).sum())
tr = tr[:n]
t0 = tr[:, 0].min()
tick = (tr[:, 0] - t0) / 100.0
ready = np.where(tr[:, 1] > 0, (tr[:, 1] - t0) / 100.0, tick)
end = (tr[:, 2] - t0) / 100.0
ph = tr[:, 3] & 0xF
ph = np.where((ph == 1) & (((tr[:, 3] >> 4) & 15) == 1), 6, ph)   # second norm
names = ["o", "norm1", "gu", "silu", "down", "qkv", "norm2"]
print(f"{model} M={M}: {n} items, launch span {end.max():.1f} us, R={os.environ.get('HX_CHAIN_R', 'default')}")
print("phase  items  first_ticket  last_ticket  first_ready  last_ready  first_end  last_end  med_run(ready->end)  med_wait(ticket->ready)")
for p in (0, 1, 2, 3, 4, 6, 5):
    m = ph == p
    if not m.any():
        continue
    print(f"{names[p]:6s} {int(m.sum()):5d}  {tick[m].min():11.2f}  {tick[m].max():11.2f}  {ready[m].min():11.2f}  "
          f"{ready[m].max():10.2f}  {end[m].min():9.2f}  {end[m].max():8.2f}  {np.median(end[m] - ready[m]):10.2f}"
          f"  {np.median(ready[m] - tick[m]):10.2f}")
# ticket hand-out rate
order = np.sort(tick)
print("tickets handed out by time (us):", " ".join(f"{int((order <= t).sum())}@{t}" for t in (1, 2, 4, 8, 16, 32, 64, 96, 128)))
print("xcc histogram:", np.bincount(((tr[:, 3] >> 8) & 15).astype(int), minlength=8))
if os.environ.get("HX_TRACE_DUMP"):
    k = int(os.environ["HX_TRACE_DUMP"])
    print("ticket kind start ready end  (every %d-th item)" % k)
    for t in range(0, n, k):
        print(f"{t:5d} {names[int(ph[t])]:6s} {tick[t]:8.2f} {ready[t]:8.2f} {end[t]:8.2f}   wait {ready[t]-tick[t]:6.2f} run {end[t]-ready[t]:6.2f}")
    # streaming concurrency: number of GEMM items between ready and end, per 4 us bin
    bins = np.arange(0, end.max() + 4, 4)
    gem = np.isin(ph, (0, 2, 4, 5))
    act = [(int(((ready[gem] <= b + 2) & (end[gem] > b + 2)).sum())) for b in bins]
    print("running GEMM items per 4-us bin:", " ".join(f"{int(b)}:{a}" for b, a in zip(bins, act)))
