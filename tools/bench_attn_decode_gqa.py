#!/usr/bin/env python3
"""Grouped-query decode attention: GB/s of unique KV bytes for the grouped kernel
(attn_decode_gqa.hip) at several key-split counts vs the per-query-head kernel."""
import math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from hydrainfer_amd import _lib
from hydrainfer_amd._C.kernel.flash_attn import mha_varlen_fwd

dev, dt = torch.device("cuda:0"), torch.bfloat16
g = torch.Generator(device=dev).manual_seed(0)
lib = _lib.lib()


def case(B, H, HK, ctx, D=128, bs=16, n_layers=4):
    nb = (ctx + bs - 1) // bs
    pool = torch.randn((n_layers, 2, B * nb, bs, HK, D), generator=g, device=dev, dtype=torch.float32).to(dt)
    perm = torch.randperm(B * nb, generator=g, device=dev).to(torch.int32)
    cu_b = torch.arange(0, (B + 1) * nb, nb, dtype=torch.int32, device=dev)
    cu_q = torch.arange(0, B + 1, dtype=torch.int32, device=dev)
    cu_k = torch.arange(0, (B + 1) * ctx, ctx, dtype=torch.int32, device=dev)
    q = torch.randn((B, H, D), generator=g, device=dev, dtype=torch.float32).to(dt)
    out = torch.empty_like(q)
    nbytes = 2 * (2 * HK * D * ctx * B + 2 * B * H * D) + 4 * B * nb

    def t(splits, iters=20):
        run = lambda i: mha_varlen_fwd(out, q, pool[i % n_layers, 0], pool[i % n_layers, 1], cu_q, cu_k, perm, cu_b,
                                       None, 1, ctx, 1 / math.sqrt(D), 0.0, -1, 0, splits)
        for i in range(3):
            run(i)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(iters):
            run(i)
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / iters * 1e3
    row = [f"B={B} H={H} HK={HK} ctx={ctx}: {nbytes / 1e6:7.1f} MB |"]
    for s in (0, 1, 2, 4, 8):
        us = t(s)
        row.append(f"s={s}: {us:6.1f}us {nbytes / us / 1e3:6.0f}GB/s")
    lib.hx_debug_set_option(b"decode_gqa", 0)
    us = t(0)
    lib.hx_debug_set_option(b"decode_gqa", 1)
    row.append(f"| per-head kernel: {us:6.1f}us {nbytes / us / 1e3:6.0f}GB/s")
    print(" ".join(row))


for args in ((32, 32, 8, 832), (32, 28, 4, 832), (32, 32, 8, 4096), (8, 32, 8, 832), (1, 32, 8, 8192), (64, 32, 8, 832)):
    case(*args)
