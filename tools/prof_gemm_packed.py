#!/usr/bin/env python3
"""Target program for rocprofv3 PMC passes (FETCH_SIZE | WRITE_SIZE, separate passes) over the
packed-weight decode GEMM: the four 7B projection shapes at M = 32, three cold launches each (a
300 MB fill between launches evicts the 256 MiB Infinity Cache)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from hydrainfer_amd._C.kernel import gemm
dev, dt = torch.device("cuda:0"), torch.bfloat16
for name, (N, K) in {"qkv": (12288, 4096), "o": (4096, 4096), "gate_up": (22016, 4096), "down": (4096, 11008)}.items():
    pk = [gemm.pack_weight((torch.randn((N, K), device=dev) * 0.02).to(dt)) for _ in range(3)]
    x = torch.randn((32, K), device=dev).to(dt)
    ws = torch.empty(gemm.workspace_floats(32, N, K), dtype=torch.float32, device=dev)
    big = torch.empty(300 * 1024 * 1024, dtype=torch.uint8, device=dev)
    for i in range(3):
        big.fill_(i)
        gemm.linear_decode_partial_packed(x, pk[i], N, ws)
    torch.cuda.synchronize()
    del pk
