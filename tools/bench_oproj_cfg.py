#!/usr/bin/env python3
"""LDS-slice kernel configurations for the o projection (its input is row-major): HX_GEMM_CFG="N:K:R:NW" per process.
    HX_GEMM_CFG=5120:5120:2:4 python tools/bench_oproj_cfg.py"""
import os, sys, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from hydrainfer_amd._C.kernel import gemm

dev, dt = torch.device("cuda:0"), torch.bfloat16
M = int(os.environ.get("M", "32"))
for (N, K) in ((4096, 4096), (5120, 5120)):
    nc = 8
    pk = [gemm.pack_weight((torch.randn((N, K), device=dev) * 0.02).to(dt)) for _ in range(nc)]
    x = torch.randn((M, K), device=dev).to(dt)
    b = torch.empty(gemm.workspace_floats(M, N, K), dtype=torch.float32, device=dev)
    fn = lambda: [gemm.linear_decode_partial_packed(x, pk[i % nc], N, b) for i in range(16)]
    s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        fn()
    torch.cuda.current_stream().wait_stream(s)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        fn()
    ts = []
    for _ in range(9):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / 16 * 1e3)
    t = statistics.median(ts)
    print(f"cfg {os.environ.get('HX_GEMM_CFG', 'default'):18s} N={N} K={K} M={M}: {t:6.2f} us {N * K * 2 / t / 1e6:5.2f} TB/s", flush=True)
