#!/usr/bin/env python3
"""Decode attention (MHA, H=32, D=128, bf16) at small batch: time vs number of key splits."""
import math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from hydrainfer_amd._C.kernel.flash_attn import mha_varlen_fwd
dev, dt = torch.device("cuda:0"), torch.bfloat16
g = torch.Generator(device=dev).manual_seed(0)
H, D, bs, n_layers = 32, 128, 16, 8
for B in (1, 2, 4, 8, 16):
    for ctx in (832, 2048):
        nb = (ctx + bs - 1) // bs
        pool = torch.randn((n_layers, 2, B * nb, bs, H, D), generator=g, device=dev, dtype=torch.float32).to(dt)
        perm = torch.randperm(B * nb, generator=g, device=dev).to(torch.int32)
        cu_b = torch.arange(0, (B + 1) * nb, nb, dtype=torch.int32, device=dev)
        cu_q = torch.arange(0, B + 1, dtype=torch.int32, device=dev)
        cu_k = torch.arange(0, (B + 1) * ctx, ctx, dtype=torch.int32, device=dev)
        q = torch.randn((B, H, D), generator=g, device=dev, dtype=torch.float32).to(dt)
        out = torch.empty_like(q)
        nbytes = 2 * (2 * H * D * ctx * B + 2 * B * H * D)
        row = [f"B={B:2d} ctx={ctx:4d} {nbytes / 1e6:6.1f}MB |"]
        for s in (0, 1, 2, 3, 4, 6, 8, 13, 26):
            run = lambda i: mha_varlen_fwd(out, q, pool[i % n_layers, 0], pool[i % n_layers, 1], cu_q, cu_k, perm,
                                           cu_b, None, 1, ctx, 1 / math.sqrt(D), 0.0, -1, 0, s)
            for i in range(3):
                run(i)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for i in range(24):
                run(i)
            e1.record(); torch.cuda.synchronize()
            row.append(f"s{s}:{e0.elapsed_time(e1) / 24 * 1e3:5.1f}")
        print(" ".join(row))
