#!/usr/bin/env python3
"""Times the four decode GEMM shapes of LLaVA-1.5-7B at M=32 (library GEMMs via torch)."""
import os, sys, statistics
import torch

def main():
    dev = torch.device("cuda:0")
    dt = torch.bfloat16
    M = int(os.environ.get("M", "32"))
    shapes = {"qkv": (12288, 4096), "o": (4096, 4096), "gate_up": (22016, 4096), "down": (4096, 11008),
              "lm_head": (32064, 4096)}
    n_copies = 6   # rotate weights so they never sit in the 256 MiB L3
    for name, (N, K) in shapes.items():
        ws = [torch.randn((N, K), device=dev, dtype=torch.float32).to(dt) * 0.02 for _ in range(n_copies)]
        x = torch.randn((M, K), device=dev, dtype=torch.float32).to(dt)
        for i in range(3):
            torch.matmul(x, ws[i % n_copies].t())
        torch.cuda.synchronize()
        ts = []
        for r in range(5):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for i in range(24):
                torch.matmul(x, ws[i % n_copies].t())
            e1.record(); torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) / 24 * 1e3)
        us = statistics.median(ts)
        print(f"{name:8s} N={N:6d} K={K:6d}: {us:7.2f} us  {N*K*2/us/1e3:7.1f} GB/s")

main()
