#!/usr/bin/env python3
"""Decode linears at batch sizes 16..64: weight-streaming HIP kernel vs library, cold weights
(a 320 MB fill evicts the Infinity Cache before every timed call)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from hydrainfer_amd._C.kernel.gemm import linear_decode
dev, dt = torch.device("cuda:0"), torch.bfloat16
shapes = {"qkv": (12288, 4096), "o": (4096, 4096), "gate_up": (22016, 4096), "down": (4096, 11008)}
evict = torch.empty(320 << 20, dtype=torch.uint8, device=dev)


def cold(fn, reps=5):
    ts = []
    for i in range(reps + 1):
        evict.fill_(i)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); e1.synchronize()
        if i:
            ts.append(e0.elapsed_time(e1) * 1e3)
    return sorted(ts)[len(ts) // 2]


for M in (16, 32, 48, 64):
    row, tot_h, tot_l = [], 0.0, 0.0
    for name, (N, K) in shapes.items():
        w = (torch.randn(N, K, device=dev, dtype=torch.float32) * 0.02).to(dt)
        x = torch.randn(M, K, device=dev, dtype=torch.float32).to(dt)
        h = cold(lambda: linear_decode(x, w))
        l = cold(lambda: torch.matmul(x, w.t()))
        tot_h += h; tot_l += l
        row.append(f"{name}: hip {h:5.1f} lib {l:5.1f}")
    print(f"M={M:2d} | " + " | ".join(row) + f" | layer: hip {tot_h:6.1f} us, lib {tot_l:6.1f} us")
