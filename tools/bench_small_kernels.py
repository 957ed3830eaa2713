#!/usr/bin/env python3
"""Per-launch cost of the small decode kernels inside a hipGraph (64 back-to-back launches)."""
import os, sys, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from hydrainfer_amd._C.kernel import norm, activation, position_embedding as pe
from hydrainfer_amd.model.llama import LLAVA_1_5_7B, build_cos_sin

dev = torch.device("cuda:0"); dt = torch.bfloat16
B, H, D, hid, inter = 32, 32, 128, 4096, 11008
x = torch.randn((B, hid), device=dev).to(dt); res = torch.randn((B, hid), device=dev).to(dt)
w = torch.ones(hid, device=dev, dtype=dt); out = torch.empty_like(x)
gu = torch.randn((B, 2 * inter), device=dev).to(dt)
qkv = torch.randn((B, 3 * hid), device=dev).to(dt)
pos = torch.arange(700, 700 + B, dtype=torch.int32, device=dev)
cs = build_cos_sin(LLAVA_1_5_7B, dt, dev)
kc = torch.zeros((64, 16, H, D), dtype=dt, device=dev); vc = torch.zeros_like(kc)
slots = torch.arange(B, dtype=torch.int32, device=dev) * 16
e = torch.empty(1, device=dev)

def timeit(fn, n=64, reps=5):
    s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        fn()
    torch.cuda.current_stream().wait_stream(s)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(n):
            fn()
    ts = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / n * 1e3)
    return statistics.median(ts)

q = qkv[:, :hid].view(B, H, D); k = qkv[:, hid:2 * hid].view(B, H, D); v = qkv[:, 2 * hid:].view(B, H, D)
print("torch fill_(1 elem)      %.2f us" % timeit(lambda: e.fill_(1.0)))
print("rms_norm 32x4096         %.2f us" % timeit(lambda: norm.rms_norm(out, x, w, 1e-5)))
print("add_rms_norm 32x4096     %.2f us" % timeit(lambda: norm.add_rms_norm(out, res, x, w, 1e-5)))
print("silu_and_mul 32x11008    %.2f us" % timeit(lambda: activation.silu_and_mul(gu[:, :inter], gu[:, inter:])))
print("rope_set_kv_cache        %.2f us" % timeit(lambda: pe.rope_set_kv_cache(q, k, v, pos, cs, D, slots, kc, vc)))
