#!/usr/bin/env python3
"""Decode attention at head_dim 256 on the reference grid's head shapes (tests/layer/test_attention.py:42-49: heads (8,8)
and (8,1), head_size 256, kv 100 / 1024, fp16 + bf16) — the decision it serves: round-4 review item 6, the 16-key-tile
instantiations spilled 28..310 registers; round 5 runs D = 256 on 8-key tiles without scratch.  A/B against a library of
another revision with HX_LIB_PATH.  32 sequences, random pages, a hipGraph of 20 launches over 4 layers' caches."""
import math, os, statistics, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from hydrainfer_amd import _lib
from hydrainfer_amd._C.kernel.flash_attn import mha_varlen_fwd

dev = torch.device("cuda:0")
B, D, bs, L = 32, 256, 16, 4
print(f"library: {_lib.LIB_PATH}")
for dt in (torch.bfloat16, torch.float16):
    for (H, HK) in ((8, 8), (8, 1)):
        for ctx in (100, 1024, 4096):
            nb = (ctx + bs - 1) // bs
            g = torch.Generator(device=dev).manual_seed(0)
            pool = torch.randn((L, 2, B * nb, bs, HK, D), generator=g, device=dev, dtype=torch.float32).to(dt)
            perm = torch.randperm(B * nb, generator=g, device=dev).to(torch.int32)
            cu_b = torch.arange(0, (B + 1) * nb, nb, dtype=torch.int32, device=dev)
            cu_q = torch.arange(0, B + 1, dtype=torch.int32, device=dev)
            cu_k = torch.arange(0, (B + 1) * ctx, ctx, dtype=torch.int32, device=dev)
            q = torch.randn((B, H, D), generator=g, device=dev, dtype=torch.float32).to(dt)
            out = torch.empty_like(q)

            def run(i):
                mha_varlen_fwd(out, q, pool[i % L, 0], pool[i % L, 1], cu_q, cu_k, perm, cu_b, None, 1, ctx, 1 / math.sqrt(D), 0.0, -1, 0, 0)
            run(0); torch.cuda.synchronize()
            gr = torch.cuda.CUDAGraph()
            with torch.cuda.graph(gr):
                for i in range(20):
                    run(i)
            gr.replay(); torch.cuda.synchronize()
            ts = []
            for _ in range(5):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(); gr.replay(); e1.record(); e1.synchronize()
                ts.append(e0.elapsed_time(e1) / 20 * 1e3)
            us = statistics.median(ts)
            nbytes = 2 * (2 * HK * D * ctx * B + 2 * B * H * D)
            print(f"{str(dt).split('.')[-1]:9s} heads ({H},{HK}) ctx {ctx:5d}: {us:8.2f} us per launch  {nbytes / us / 1e3:8.1f} GB/s", flush=True)
