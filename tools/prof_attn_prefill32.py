#!/usr/bin/env python3
"""Target program for rocprofv3 PMC passes over the prefill attention kernels: HX_FWD32=0/1 picks
the 16x16x32 or the 32x32x16 kernel, HX_PREFILL_B the number of 704-token sequences."""
import math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from hydrainfer_amd import _lib
from hydrainfer_amd._C.kernel.flash_attn import mha_varlen_fwd

dev, dt = torch.device("cuda:0"), torch.bfloat16
B, H, D, bs, n = int(os.environ.get("HX_PREFILL_B", "4")), 32, 128, 16, 704
_lib.lib().hx_debug_set_option(b"fwd_mfma32", int(os.environ.get("HX_FWD32", "1")))
if os.environ.get("HX_ABL"):      # EXPERIMENTS builds: tools/ablate_attn_prefill32.py's variants under the profiler
    assert _lib.lib().hx_debug_set_option(b"fwd_ablate", int(os.environ["HX_ABL"])) == 0
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
print("algorithmic_flops_per_launch", 4 * H * D * B * (n * (n + 1) // 2))
