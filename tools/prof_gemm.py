#!/usr/bin/env python3
"""Cold-weight timing of the decode GEMM shapes (library vs HIP kernel variants): a 300 MB
fill between launches evicts the 256 MiB Infinity Cache, as the 13 GB weight stream of a real
decode step does.  Durations come from the rocprofv3 kernel trace of this program."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from hydrainfer_amd import _lib
from hydrainfer_amd._C.kernel.gemm import linear_decode
dev = torch.device("cuda:0"); dt = torch.bfloat16
lib = _lib.lib()
shapes = {"qkv": (12288, 4096), "o": (4096, 4096), "gate_up": (22016, 4096), "down": (4096, 11008)}
variants = [tuple(int(a) for a in v.split("x")) for v in os.environ.get("VARIANTS", "0x0").split(",")]
for name, (N, K) in shapes.items():
    ws = [(torch.randn((N, K), device=dev, dtype=torch.float32) * 0.02).to(dt) for _ in range(3)]
    x = torch.randn((32, K), device=dev, dtype=torch.float32).to(dt)
    big = torch.empty(300 * 1024 * 1024, dtype=torch.uint8, device=dev)
    for (r, nw) in variants:
        lib.hx_debug_set_option(b"gemm_rows_per_wave", r)
        lib.hx_debug_set_option(b"gemm_waves", nw)
        for i in range(3):
            big.fill_(i)
            linear_decode(x, ws[i])
    for i in range(3):
        big.fill_(i + 7)
        torch.matmul(x, ws[i].t())
    torch.cuda.synchronize()
