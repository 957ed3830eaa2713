#!/usr/bin/env python3
"""Decode GEMM (M=32, 7B shapes): row-major weights (gemm_skinny_kernel) vs packed weights
(gemm_packed_kernel), slabs only, inside a hipGraph, weights rotated (cold).  Checks bit-identity."""
import os, sys, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from hydrainfer_amd._C.kernel import gemm

dev, dt = torch.device("cuda:0"), torch.bfloat16
M = int(os.environ.get("M", "32"))


def graph_time(fn, n_inner, reps=7):
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


for name, (N, K) in {"qkv": (12288, 4096), "o": (4096, 4096), "gate_up": (22016, 4096), "down": (4096, 11008)}.items():
    nc = 6
    ws = [(torch.randn((N, K), device=dev) * 0.02).to(dt) for _ in range(nc)]
    pk = [gemm.pack_weight(w) for w in ws]
    x = torch.randn((M, K), device=dev).to(dt)
    a = torch.empty(gemm.workspace_floats(M, N, K), dtype=torch.float32, device=dev)
    b = torch.empty_like(a)
    sa = gemm.linear_decode_partial(x, ws[0], a)
    sb = gemm.linear_decode_partial_packed(x, pk[0], N, b)
    same = sa == sb and torch.equal(a[: sa * M * N], b[: sb * M * N])
    t0 = graph_time(lambda: [gemm.linear_decode_partial(x, ws[i % nc], a) for i in range(12)], 12)
    t1 = graph_time(lambda: [gemm.linear_decode_partial_packed(x, pk[i % nc], N, b) for i in range(12)], 12)
    print(f"{name:8s} N={N:6d} K={K:6d}: row-major {t0:6.2f} us {N*K*2/t0/1e6:5.2f} TB/s | packed {t1:6.2f} us "
          f"{N*K*2/t1/1e6:5.2f} TB/s | bit-identical {same}", flush=True)
    del ws, pk
