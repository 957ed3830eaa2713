#!/usr/bin/env python3
"""Where the fused decode attention launch spends its time: in-kernel time stamps (100 MHz s_memrealtime) written by wave 0
of every workgroup of attn_decode_kernel in an EXPERIMENTS build (hx_debug_fwd_stamps sets the buffer).
    make -C hydrainfer_amd/csrc EXPERIMENTS=1 OUTDIR=../../build/lib_exp
    HX_LIB_PATH=$PWD/build/lib_exp/libhydra_hip.so python tools/decode_timeline.py [ctx]
Decision it serves: which part of the launch's 3 us over its math-free stand-in (tools/null_layer.py) is head, tail or
steady state.  Events: 0 entry, 1 scalar metadata in, 2 page ids in + first tile requested, 3 fused prologue done,
4 first tile computed, 5 last tile computed, 6 merge barrier passed, 7 end."""
import math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from hydrainfer_amd import _lib
from hydrainfer_amd._C.kernel import gemm
from hydrainfer_amd._C.kernel.flash_attn import decode_attention_fused

ctx = int(sys.argv[1]) if len(sys.argv) > 1 else 832
dev, dt = torch.device("cuda:0"), torch.bfloat16
B, H, D, bs, hid, L = 32, 32, 128, 16, 4096, 6
nb_seq = (ctx + bs - 1) // bs
g = torch.Generator(device=dev).manual_seed(0)
pool = torch.randn((L, 2, B * nb_seq, bs, H, D), generator=g, device=dev, dtype=torch.float32).to(dt)
perm = torch.randperm(B * nb_seq, generator=g, device=dev).to(torch.int32)
cu_b = torch.arange(0, (B + 1) * nb_seq, nb_seq, dtype=torch.int32, device=dev)
cu_q = torch.arange(0, B + 1, dtype=torch.int32, device=dev)
cu_k = torch.arange(0, (B + 1) * ctx, ctx, dtype=torch.int32, device=dev)
pos = torch.full((B,), ctx - 1, dtype=torch.int32, device=dev)
slots = torch.stack([perm[b * nb_seq + (ctx - 1) // bs] * bs + (ctx - 1) % bs for b in range(B)]).to(torch.int32)
out = torch.empty((B, H, D), dtype=dt, device=dev)
x = torch.randn((B, hid), generator=g, device=dev, dtype=torch.float32).to(dt)
wqkv = (torch.randn((3 * H * D, hid), generator=g, device=dev, dtype=torch.float32) * 0.02).to(dt)
slabs = torch.empty(gemm.xreg_workspace_floats(B, 3 * H * D, hid), dtype=torch.float32, device=dev)   # ONE slab, as in the step
inv = 1.0 / torch.pow(10000.0, torch.arange(0, D, 2, dtype=torch.float) / D)
fr = torch.einsum("i,j->ij", torch.arange(4096, dtype=torch.float), inv)
cos_sin = torch.cat([fr.cos()[:, None, :], fr.sin()[:, None, :]], dim=1).to(dt).to(dev)
n_slabs = gemm.linear_decode_partial_xreg(x, gemm.pack_weight_xreg(wqkv), 3 * H * D, slabs)
junk = torch.empty(1 << 28, dtype=torch.uint8, device=dev)


def launch(layer):
    decode_attention_fused(out, out, out, out, pool[layer, 0], pool[layer, 1], pos, cos_sin, slots, cu_q, cu_k, perm, cu_b, ctx,
                           1 / math.sqrt(D), 0, slabs, n_slabs)


assert _lib.has_experiments(), "build with make -C hydrainfer_amd/csrc EXPERIMENTS=1 OUTDIR=../../build/lib_exp and set HX_LIB_PATH"
l = _lib.lib()
for i in range(6):
    launch(i % L)
buf = torch.zeros(B * H * 16, dtype=torch.int64, device=dev)
l.hx_debug_fwd_stamps(buf.data_ptr())
rows = []
for rep in range(5):
    junk.zero_()                      # a writing kernel in front, as a GEMM would be
    buf.zero_()
    launch(rep % L)
    torch.cuda.synchronize()
    a = buf.cpu().numpy().reshape(B * H, 16)[:, :8].astype(np.int64)
    t0 = a[:, 0].min()
    rows.append((a - t0) / 100.0)     # us since the first workgroup's entry
l.hx_debug_fwd_stamps(None)
a = np.median(np.stack(rows[1:]), axis=0)       # [workgroup, event]
names = ["entry", "scalar metadata in", "page ids in, first tile requested", "fused prologue done", "first tile computed",
         "last tile computed", "merge barrier passed", "end"]
print(f"fused decode attention, B={B} H={H} D={D} ctx={ctx}: {B * H} workgroups, us since the first workgroup entered (median of 4 launches)")
print("| event | earliest | median | latest |\n|---|---|---|---|")
for k, n in enumerate(names):
    print(f"| {k} {n} | {a[:, k].min():.2f} | {np.median(a[:, k]):.2f} | {a[:, k].max():.2f} |")
d = np.diff(a, axis=1)
print("\nper workgroup, between consecutive events (median over workgroups): " +
      "; ".join(f"{names[k]} -> {names[k + 1]} {np.median(d[:, k]):.2f}" for k in range(7)))
