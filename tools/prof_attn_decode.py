#!/usr/bin/env python3
"""Launches the decode-attention kernel a few times at the BASELINE shape (B=32, H=32, D=128,
bf16, ctx 832) over distinct layers of a pool larger than the 256 MiB Infinity Cache —
the target program for `rocprofv3 --pmc ...` counter passes."""
import math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from hydrainfer_amd._C.kernel.flash_attn import decode_attention_fused, decode_rank
from hydrainfer_amd.model.llama import LLAVA_1_5_7B, build_cos_sin

dev = torch.device("cuda:0")
dt = torch.bfloat16
B, H, D, bs, ctx, L = 32, 32, 128, 16, 832, 6
nb_seq = (ctx + bs - 1) // bs
n_blocks = B * nb_seq
g = torch.Generator(device=dev).manual_seed(0)
pool = torch.randn((L, 2, n_blocks, bs, H, D), generator=g, device=dev, dtype=torch.float32).to(dt)
perm = torch.randperm(n_blocks, generator=g, device=dev).to(torch.int32)
cu_b = torch.arange(0, (B + 1) * nb_seq, nb_seq, dtype=torch.int32, device=dev)
cu_q = torch.arange(0, B + 1, dtype=torch.int32, device=dev)
cu_k = torch.arange(0, (B + 1) * ctx, ctx, dtype=torch.int32, device=dev)
q = torch.randn((B, H, D), generator=g, device=dev, dtype=torch.float32).to(dt)
out = torch.empty_like(q)
k_new = torch.randn((B, H, D), generator=g, device=dev, dtype=torch.float32).to(dt)
v_new = torch.randn((B, H, D), generator=g, device=dev, dtype=torch.float32).to(dt)
pos = torch.full((B,), ctx - 1, dtype=torch.int32, device=dev)
cs = build_cos_sin(LLAVA_1_5_7B, dt, dev)
slots = (perm[cu_b[:-1].long() + (ctx - 1) // bs] * bs + (ctx - 1) % bs).to(torch.int32)
# the variant the decode graph runs: q / k / v arrive as the split-K slabs of the qkv projection
from hydrainfer_amd._C.kernel import gemm as hip_gemm
x = torch.randn((B, H * D), generator=g, device=dev, dtype=torch.float32).to(dt)
wqkv = (torch.randn((3 * H * D, H * D), generator=g, device=dev, dtype=torch.float32) * 0.02).to(dt)
# (round 2: layers >= 1 get ONE slab from the activations-in-registers GEMM)
slabs = torch.empty(hip_gemm.xreg_workspace_floats(B, 3 * H * D, H * D), dtype=torch.float32, device=dev)
n_slabs = hip_gemm.linear_decode_partial_xreg(x, hip_gemm.pack_weight_xreg(wqkv), 3 * H * D, slabs)
rank = decode_rank(cu_k)      # (round 6: the step hands the launch its rank descriptor — an even batch: flag 0, the RANKED kernel's static numbering)
for i in range(12):
    decode_attention_fused(out, q, k_new, v_new, pool[i % L, 0], pool[i % L, 1], pos, cs, slots, cu_q, cu_k,
                           perm, cu_b, ctx, 1 / math.sqrt(D), 1, slabs, n_slabs, rank)
torch.cuda.synchronize()
nbytes = 2 * (2 * H * D * ctx * B + 2 * B * H * D) + 4 * B * nb_seq
print("algorithmic_bytes_per_launch", nbytes, "qkv_slab_bytes", n_slabs * B * 3 * H * D * 4)
