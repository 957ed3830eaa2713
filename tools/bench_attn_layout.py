#!/usr/bin/env python3
"""Decode attention over the reference's cache layout [block][token][head][dim] vs a head-major
block [block][head][token][dim] (the same logical tensor, passed as a strided view): does a
contiguous 4 KiB (head, block) tile stream faster than 16 rows 8 KiB apart?  B=32, H=32, D=128."""
import math, os, statistics, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from hydrainfer_amd._C.kernel.flash_attn import mha_varlen_fwd

dev, dt = torch.device("cuda:0"), torch.bfloat16
B, H, D, bs, L = 32, 32, 128, 16, 8
for ctx in (712, 832, 959):
    nb_seq = (ctx + bs - 1) // bs
    n_blocks = B * nb_seq
    g = torch.Generator(device=dev).manual_seed(0)
    std = torch.randn((L, 2, n_blocks, bs, H, D), generator=g, device=dev, dtype=torch.float32).to(dt)
    hm = std.permute(0, 1, 2, 4, 3, 5).contiguous()                 # [L, 2, nb, H, bs, D]
    hm_view = hm.permute(0, 1, 2, 4, 3, 5)                          # logical [L, 2, nb, bs, H, D]
    perm = torch.randperm(n_blocks, generator=g, device=dev).to(torch.int32)
    cu_b = torch.arange(0, (B + 1) * nb_seq, nb_seq, dtype=torch.int32, device=dev)
    cu_q = torch.arange(0, B + 1, dtype=torch.int32, device=dev)
    cu_k = torch.arange(0, (B + 1) * ctx, ctx, dtype=torch.int32, device=dev)
    q = torch.randn((B, H, D), generator=g, device=dev, dtype=torch.float32).to(dt)
    o1, o2 = torch.empty_like(q), torch.empty_like(q)
    scale = 1 / math.sqrt(D)
    nbytes = 2 * (2 * H * D * ctx * B + 2 * B * H * D) + 4 * B * nb_seq

    def timeit(pool, out):
        def body():
            for l in range(2 * L):
                mha_varlen_fwd(out, q, pool[l % L, 0], pool[l % L, 1], cu_q, cu_k, perm, cu_b, None, 1, ctx, scale,
                               0.0, -1, 0, 1)
        body(); torch.cuda.synchronize()
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr):
            body()
        ts = []
        for _ in range(7):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); gr.replay(); e1.record(); torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) / (2 * L) * 1e3)
        return statistics.median(ts)
    t1, t2 = timeit(std, o1), timeit(hm_view, o2)
    print(f"ctx {ctx}: token-major {t1:.1f} us {nbytes / t1 / 1e6:.2f} TB/s | head-major {t2:.1f} us "
          f"{nbytes / t2 / 1e6:.2f} TB/s | same output {torch.equal(o1, o2)}", flush=True)
    del std, hm
