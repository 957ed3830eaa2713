#!/usr/bin/env python3
"""The elementwise / scatter ops of the hot path at PREFILL sizes (2816 = 4 x 704 tokens and
22528 = 32 x 704 tokens of a 7B model): time per launch and HBM rate against the algorithmic bytes."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from hydrainfer_amd._C.kernel import activation, cache_kernels, kv_cache_kernels, norm, position_embedding as pe

dev, dt = torch.device("cuda:0"), torch.bfloat16
H, D, hidden, inter, bs = 32, 128, 4096, 11008, 16


def timeit(fn, reps=20):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


for n in (2816, 22528):
    g = torch.Generator(device=dev).manual_seed(0)
    rnd = lambda *s: torch.randn(s, device=dev, generator=g).to(dt)
    x, res, w = rnd(n, hidden), rnd(n, hidden), rnd(hidden)
    out = torch.empty_like(x)
    gate_up = rnd(n, 2 * inter)
    q, k, v = rnd(n, H, D), rnd(n, H, D), rnd(n, H, D)
    pos = torch.arange(n, dtype=torch.int32, device=dev) % 704
    cos_sin = rnd(4096, 2, D // 2)
    n_blocks = (n + bs - 1) // bs + 4
    kc, vc = torch.zeros((n_blocks, bs, H, D), dtype=dt, device=dev), torch.zeros((n_blocks, bs, H, D), dtype=dt, device=dev)
    slots = torch.randperm(n_blocks * bs, device=dev, generator=g)[:n].to(torch.int32)
    img = rnd(n, H, D)
    img_cache = torch.zeros((n_blocks, bs, H, D), dtype=dt, device=dev)
    e = 2
    rows = [
        ("rms_norm", lambda: norm.rms_norm(out, x, w, 1e-5), 2 * n * hidden * e),
        ("add_rms_norm (residual updated in place)", lambda: norm.add_rms_norm(out, res, x, w, 1e-5), 4 * n * hidden * e),
        ("silu_and_mul", lambda: activation.silu_and_mul(gate_up[:, :inter], gate_up[:, inter:]), 3 * n * inter * e),
        ("apply_rotary_pos_emb (q, k in place)", lambda: pe.apply_rotary_pos_emb(q, k, pos, cos_sin, D, False), 4 * n * H * D * e),
        ("set_kv_cache", lambda: kv_cache_kernels.set_kv_cache(slots, k, v, kc, vc), 4 * n * H * D * e),
        ("rope_set_kv_cache (rope + append, one launch)", lambda: pe.rope_set_kv_cache(q, k, v, pos, cos_sin, D, slots, kc, vc), 7 * n * H * D * e),
        ("set_image_cache", lambda: cache_kernels.set_image_cache(slots, img, img_cache), 2 * n * hidden * e),
    ]
    print(f"{n} tokens")
    for what, fn, b in rows:
        us = timeit(fn)
        print(f"  {what:48s} {us:8.1f} us  {b / us / 1e3:8.1f} GB/s  ({b / 1e6:.1f} MB)")
