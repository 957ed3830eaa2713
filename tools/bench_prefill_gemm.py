#!/usr/bin/env python3
"""Library GEMM rate at prefill sizes (M = 704 / 2048 rows) for the 7B projections, weight stored
[N,K] (what the decode kernel streams) vs [K,N], and row / column splits of the gate|up product.
CAUTION: calls are timed back to back, so weights (33-180 MB) sit in the 256 MiB Infinity Cache:
the split variants, which read the weights twice, look up to 25 % faster here than the single
call, but in the pipeline (cold weights) an autotuned row split measured no gain (704-token
prefill 11.88 ms either way; burst TTFT within noise) and was not kept."""
import os, sys
import torch
dev, dt = torch.device("cuda:0"), torch.bfloat16
shapes = {"qkv": (12288, 4096), "o": (4096, 4096), "gate_up": (22016, 4096), "down": (4096, 11008)}


def timeit(fn, reps=20):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


for M in (704, 2048, 4096):
    tot = {"nk": 0.0, "kn": 0.0}
    for name, (N, K) in shapes.items():
        x = torch.randn(M, K, device=dev, dtype=torch.float32).to(dt)
        w_nk = (torch.randn(N, K, device=dev, dtype=torch.float32) * 0.02).to(dt)
        w_kn = w_nk.t().contiguous()
        a = timeit(lambda: torch.matmul(x, w_nk.t()))
        b = timeit(lambda: torch.matmul(x, w_kn))
        tot["nk"] += a; tot["kn"] += b
        fl = 2 * M * N * K
        print(f"M={M:5d} {name:8s} [N,K]: {a:7.1f} us {fl / a / 1e6:6.0f} TF/s | [K,N]: {b:7.1f} us {fl / b / 1e6:6.0f} TF/s")
    print(f"M={M:5d} layer total [N,K] {tot['nk']:.0f} us, [K,N] {tot['kn']:.0f} us")

# the fused gate|up projection at M = 2048 lands on a slow library solution: compare with two
# separate N = 11008 products and with one product over a row-split of x
print("--- gate|up variants")
for M in (704, 1024, 1536, 2048, 3072):
    K, N = 4096, 22016
    x = torch.randn(M, K, device=dev, dtype=torch.float32).to(dt)
    w = (torch.randn(N, K, device=dev, dtype=torch.float32) * 0.02).to(dt)
    out = torch.empty(M, N, device=dev, dtype=dt)
    a = timeit(lambda: torch.matmul(x, w.t(), out=out))
    def two():
        torch.matmul(x, w[:N // 2].t(), out=out[:, :N // 2]) if False else None
        g = torch.matmul(x, w[:N // 2].t()); u = torch.matmul(x, w[N // 2:].t())
    b = timeit(two)
    def halves():
        h = M // 2
        torch.matmul(x[:h], w.t(), out=out[:h]); torch.matmul(x[h:], w.t(), out=out[h:])
    c = timeit(halves)
    fl = 2 * M * N * K
    print(f"M={M:5d} fused {a:7.1f} us {fl / a / 1e6:6.0f} TF/s | gate,up separately {b:7.1f} us {fl / b / 1e6:6.0f} TF/s | two row halves {c:7.1f} us {fl / c / 1e6:6.0f} TF/s")
