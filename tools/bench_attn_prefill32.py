#!/usr/bin/env python3
"""Prefill / dense attention: 16x16x32 kernel (fwd_mfma32=0) vs the 32x32x16 kernel (fwd_mfma32=1)
on the LLaVA-1.5 shapes; checks that both give the same output within the attention tolerance."""
import math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from hydrainfer_amd import _lib
from hydrainfer_amd._C.kernel.flash_attn import mha_varlen_fwd

dev, dt = torch.device("cuda:0"), torch.bfloat16
g = torch.Generator(device=dev).manual_seed(0)
rnd = lambda *s: torch.randn(s, generator=g, device=dev, dtype=torch.float32).to(dt)


def timeit(fn, reps=30):
    for _ in range(5):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


def paged(B, n, kv, H=32, D=128, bs=16):
    nb = (kv + bs - 1) // bs
    kc, vc = rnd(B * nb, bs, H, D), rnd(B * nb, bs, H, D)
    q = rnd(B * n, H, D)
    out = torch.empty_like(q)
    perm = torch.randperm(B * nb, generator=g, device=dev).to(torch.int32)
    cu_b = torch.arange(0, (B + 1) * nb, nb, dtype=torch.int32, device=dev)
    cu_q = torch.arange(0, (B + 1) * n, n, dtype=torch.int32, device=dev)
    cu_k = torch.arange(0, (B + 1) * kv, kv, dtype=torch.int32, device=dev)
    flops = 4 * H * D * B * sum(kv - n + i + 1 for i in range(n))
    return (lambda: mha_varlen_fwd(out, q, kc, vc, cu_q, cu_k, perm, cu_b, None, n, kv, 1 / math.sqrt(D), 0, -1, 0, 0)), flops, out


def dense(n_img, n=577, H=16, D=64):
    q, k, v = rnd(n_img * n, H, D), rnd(n_img * n, H, D), rnd(n_img * n, H, D)
    out = torch.empty_like(q)
    cu = torch.arange(0, (n_img + 1) * n, n, dtype=torch.int32, device=dev)
    return (lambda: mha_varlen_fwd(out, q, k, v, cu, cu, None, None, None, n, n, 1 / math.sqrt(D), 0, -1, -1, 0)), 4 * H * D * n_img * n * n, out


cases = {"prefill 4x704": paged(4, 704, 704), "prefill 1x704": paged(1, 704, 704), "prefill 32x704": paged(32, 704, 704),
         "chunk 1x2048 of 2048": paged(1, 2048, 2048), "chunk 3x683 of 704": paged(3, 683, 704),
         "chunk 2048 of 4096": paged(1, 2048, 4096), "clip 1x577 d64": dense(1), "clip 8x577 d64": dense(8)}
l = _lib.lib()
# (label, options): the 16x16x32 kernel, the 32x32x16 kernel with one workgroup per item, its persistent form
variants = [("16x16x32", {"fwd_mfma32": 0}), ("32x32x16 per item", {"fwd_persistent": 0}),
            ("persistent", {}), ("persistent prio 0", {"fwd_priority": 0}), ("persistent prio 1", {"fwd_priority": 1}),
            ("single tiles", {"fwd_units": 0}), ("units", {"fwd_units": 1}), ("units short first", {"fwd_units": 2}),
            ("groups of 4", {"fwd_seq_group": 4}), ("groups of 2", {"fwd_seq_group": 2}), ("groups of 1", {"fwd_seq_group": 1})]
if len(sys.argv) > 1:
    variants = [v for v in variants if v[0] in sys.argv[1].split(",")] or variants
defaults = {"fwd_mfma32": 1, "fwd_persistent": 1, "fwd_priority": -1, "fwd_seq_group": 0, "fwd_units": -1}
for name, (fn, flops, out) in cases.items():
    res, outs = [], []
    for label, opts in variants:
        for k, v in {**defaults, **opts}.items():
            l.hx_debug_set_option(k.encode(), v)
        us = timeit(fn)
        fn(); torch.cuda.synchronize()
        outs.append(out.float().clone())
        res.append(f"{label} {us:6.1f} us {flops / us / 1e6:4.0f} TF")
    err = max((o - outs[0]).abs().max().item() for o in outs)
    print(f"{name:22s} " + " | ".join(res) + f" | max |diff| {err:.4f}", flush=True)
for k, v in defaults.items():
    l.hx_debug_set_option(k.encode(), v)
