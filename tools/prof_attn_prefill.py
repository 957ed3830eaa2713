#!/usr/bin/env python3
"""Launches the prefill attention kernel at the BASELINE shape (4 sequences x 704 new tokens,
H=HK=32, D=128, bf16, paged KV, causal) a few times — the target program for rocprofv3
kernel-trace / PMC passes (MFMA busy, LDS bank conflicts)."""
import math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from hydrainfer_amd._C.kernel.flash_attn import mha_varlen_fwd

dev = torch.device("cuda:0")
dt = torch.bfloat16
B, H, D, bs, n = 4, 32, 128, 16, 704
nb_seq = (n + bs - 1) // bs
g = torch.Generator(device=dev).manual_seed(0)
kc = torch.randn((B * nb_seq, bs, H, D), generator=g, device=dev, dtype=torch.float32).to(dt)
vc = torch.randn((B * nb_seq, bs, H, D), generator=g, device=dev, dtype=torch.float32).to(dt)
q = torch.randn((B * n, H, D), generator=g, device=dev, dtype=torch.float32).to(dt)
out = torch.empty_like(q)
perm = torch.randperm(B * nb_seq, generator=g, device=dev).to(torch.int32)
cu_b = torch.arange(0, (B + 1) * nb_seq, nb_seq, dtype=torch.int32, device=dev)
cu = torch.arange(0, (B + 1) * n, n, dtype=torch.int32, device=dev)
for _ in range(12):
    mha_varlen_fwd(out, q, kc, vc, cu, cu, perm, cu_b, None, n, n, 1 / math.sqrt(D), 0, -1, 0, 0)
torch.cuda.synchronize()
flops = 4 * H * D * B * (n * (n + 1) // 2)
print("algorithmic_flops_per_launch", flops)
