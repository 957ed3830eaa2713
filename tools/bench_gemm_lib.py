#!/usr/bin/env python3
"""Library-GEMM alternatives for the M=32 decode shapes: weight layout [N,K] (x @ W^T) vs
pre-transposed [K,N] (x @ Wt), hipBLASLt vs rocBLAS backend.  In-graph timing, weights rotated."""
import os, sys, statistics
import torch

def graph_time(fn, n_inner, reps=5):
    s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        fn()
    torch.cuda.current_stream().wait_stream(s)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        fn()
    ts = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / n_inner * 1e3)
    return statistics.median(ts)

dev = torch.device("cuda:0"); dt = torch.bfloat16
shapes = {"qkv": (12288, 4096), "o": (4096, 4096), "gate_up": (22016, 4096), "down": (4096, 11008)}
for backend in ("hipblaslt", "cublas"):
    try:
        torch.backends.cuda.preferred_blas_library(backend)
    except Exception as e:
        print("backend", backend, "unavailable", e); continue
    for name, (N, K) in shapes.items():
        nc = 8
        ws = [(torch.randn((N, K), device=dev, dtype=torch.float32) * 0.02).to(dt) for _ in range(nc)]
        wts = [w.t().contiguous() for w in ws]
        x = torch.randn((32, K), device=dev, dtype=torch.float32).to(dt)
        outs = [torch.empty((32, N), dtype=dt, device=dev) for _ in range(nc)]
        def f_nk():
            for i in range(16):
                torch.matmul(x, ws[i % nc].t(), out=outs[i % nc])
        def f_kn():
            for i in range(16):
                torch.matmul(x, wts[i % nc], out=outs[i % nc])
        a, b = graph_time(f_nk, 16), graph_time(f_kn, 16)
        print(f"{backend:10s} {name:8s}: W[N,K] {a:7.2f} us {N*K*2/a/1e3:7.1f} GB/s | Wt[K,N] {b:7.2f} us {N*K*2/b/1e3:7.1f} GB/s")
        del ws, wts
