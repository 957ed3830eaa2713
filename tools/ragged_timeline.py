#!/usr/bin/env python3
"""Where a RAGGED decode-attention launch spends its time, per workgroup and per CU: in-kernel time stamps (100 MHz) of
attn_decode_kernel in an EXPERIMENTS build, static grid against the RANKED form.
    make -C hydrainfer_amd/csrc EXPERIMENTS=1 OUTDIR=../../build/lib_exp
    HX_LIB_PATH=$PWD/build/lib_exp/libhydra_hip.so python tools/ragged_timeline.py [uniform|bimodal|<ctx>]"""
import math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from hydrainfer_amd import _lib
from hydrainfer_amd._C.kernel.flash_attn import decode_attention_fused, decode_rank
from hydrainfer_amd.model.runner import ragged_contexts

kind = sys.argv[1] if len(sys.argv) > 1 else "uniform"
dev, dt = torch.device("cuda:0"), torch.bfloat16
B, H, D, bs, L = 32, 32, 128, 16, 4
lens = ragged_contexts(kind, B) if kind in ("uniform", "bimodal") else [int(kind)] * B
g = torch.Generator(device=dev).manual_seed(0)
nb = [(l + bs - 1) // bs for l in lens]
n_blocks = sum(nb) + 8
pool = torch.randn((L, 2, n_blocks, bs, H, D), generator=g, device=dev, dtype=torch.float32).to(dt)
perm = torch.randperm(n_blocks, generator=g, device=dev).to(torch.int32)[: sum(nb)].contiguous()
cu_b = torch.tensor([0] + list(np.cumsum(nb)), dtype=torch.int32, device=dev)
cu_k = torch.tensor([0] + list(np.cumsum(lens)), dtype=torch.int32, device=dev)
cu_q = torch.arange(0, B + 1, dtype=torch.int32, device=dev)
q = torch.randn((B, H, D), generator=g, device=dev, dtype=torch.float32).to(dt)
out = torch.empty_like(q)
junk = torch.empty(1 << 28, dtype=torch.uint8, device=dev)
assert _lib.has_experiments(), "EXPERIMENTS build needed (see the docstring)"
l = _lib.lib()


inv = 1.0 / torch.pow(10000.0, torch.arange(0, D, 2, dtype=torch.float) / D)
fr = torch.einsum("i,j->ij", torch.arange(4096, dtype=torch.float), inv)
cos_sin = torch.cat([fr.cos()[:, None, :], fr.sin()[:, None, :]], dim=1).to(dt).to(dev)
k_new = torch.randn((B, H, D), generator=g, device=dev, dtype=torch.float32).to(dt)
v_new = torch.randn((B, H, D), generator=g, device=dev, dtype=torch.float32).to(dt)
pos = torch.tensor([l_ - 1 for l_ in lens], dtype=torch.int32, device=dev)
slots = torch.stack([perm[int(cu_b[i]) + (l_ - 1) // bs] * bs + (l_ - 1) % bs for i, l_ in enumerate(lens)]).to(torch.int32)
rank = decode_rank(cu_k)
use_rank = [None]


def launch(layer):
    decode_attention_fused(out, q, k_new, v_new, pool[layer, 0], pool[layer, 1], pos, cos_sin, slots, cu_q, cu_k, perm, cu_b, 960,
                           1 / math.sqrt(D), 0, rank_desc=use_rank[0])


def run(dealt):
    use_rank[0] = rank if dealt else None
    for i in range(4):
        launch(i % L)
    buf = torch.zeros(3 * 1024 * 16, dtype=torch.int64, device=dev)
    l.hx_debug_fwd_stamps(buf.data_ptr())
    rows = []
    for rep in range(5):
        junk.zero_(); buf.zero_()
        launch(rep % L)
        torch.cuda.synchronize()
        rows.append(buf.cpu().numpy().reshape(-1, 16).astype(np.int64).copy())
    l.hx_debug_fwd_stamps(None)
    a = rows[-1]
    live = a[:, 0] > 0
    t0 = a[live, 0].min()
    ev = (a[:, :8] - t0) / 100.0
    print(f"--- {'ranked' if dealt else 'static'}: {int(live.sum())} workgroups, launch ends at {ev[live, 7].max():.1f} us")
    n_wg = 1024
    wg = np.arange(a.shape[0]) % n_wg
    cu_end = np.zeros(256)
    for c in range(256):
        m = live & (wg % 256 == c)
        cu_end[c] = ev[m, 7].max()
    print(f"    CU end times: min {cu_end.min():.1f}  median {np.median(cu_end):.1f}  max {cu_end.max():.1f}")
    dur = ev[:, 7] - ev[:, 0]
    if dealt:
        ident = a[:, 8]
        seq, rng = (ident >> 16) & 0xff, np.zeros(a.shape[0], dtype=np.int64)
    else:
        seq, rng = (np.arange(a.shape[0]) // H) % B, np.zeros(a.shape[0], dtype=np.int64)
    worst = np.argsort(-ev[:, 7] * live)[:8]
    for i in worst:
        print(f"    item row {i}: wg {wg[i]} (CU {wg[i] % 256}, slot {wg[i] // 256}, iter {i // n_wg}) seq {seq[i]} len {lens[seq[i]]} range {rng[i]}: "
              + " ".join(f"{ev[i, k]:.1f}" for k in range(8)))
    # the items of the CU that ended last
    c = int(np.argmax(cu_end))
    print(f"    items of CU {c}:")
    for i in np.nonzero(live & (wg % 256 == c))[0]:
        print(f"      row {i} slot {wg[i] // 256} iter {i // n_wg} seq {seq[i]} len {lens[seq[i]]} range {rng[i]}: start {ev[i, 0]:.1f} tiles done {ev[i, 5]:.1f} end {ev[i, 7]:.1f}")
    # duration against length
    by = {}
    for i in np.nonzero(live)[0]:
        by.setdefault((lens[seq[i]], int(rng[i])), []).append(dur[i])
    print("    item duration by (sequence length, range): " + "; ".join(f"{k}: {np.median(v):.1f}" for k, v in sorted(by.items())[-10:]))


print(f"{kind}: lens {lens}")
run(0)
run(1)
