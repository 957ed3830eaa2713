#!/usr/bin/env python3
"""Decision served: should the KV pool's allocation control where its bytes lie?  bench.py's bare 64-row step was 2 %
faster when the serving leg had run before it (7.74 against 7.90 ms); the kernel trace put all of it in the
decode-attention launches (136.5 against 141.7 us).  Same block tables, same kernel: only where the pool lies differed.
This probe times the fused decode-attention launch (B rows, ctx 832, 7B heads, the runner's block tables) over pools
carved out of one buffer at different byte offsets, and with spare bytes between the K and V pools of a layer.
Result (profiles/r5_kv_pool_placement.md): an odd multiple of 256 bytes between K and V is worth 5-6 % of the launch;
hydrainfer_amd/memory/kv_pool.py allocates pools that way.

    python tools/probes/attn_placement.py [B=64] [L=8]
"""
import math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from hydrainfer_amd._C.kernel.flash_attn import decode_attention_fused
from hydrainfer_amd._C.kernel import gemm as hip_gemm
from hydrainfer_amd.model.llama import LLAVA_1_5_7B, build_cos_sin
from hydrainfer_amd.model.runner import plan_block_tables

dev, dt = torch.device("cuda:0"), torch.bfloat16
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
L = int(sys.argv[2]) if len(sys.argv) > 2 else 8
H, D, bs, ctx = 32, 128, 16, 832
prompt, n_gen = 704, 256
bps = (prompt + n_gen - 1 + bs - 1) // bs
n_blocks = B * bps
tables = plan_block_tables(B, prompt, n_gen, bs, n_blocks)
flat = [b for t in tables for b in (t + [0] * (bps - len(t)))]
i32 = dict(dtype=torch.int32, device=dev)
table = torch.tensor(flat, **i32)
cu_b = torch.arange(0, (B + 1) * bps, bps, **i32)
cu_q = torch.arange(0, B + 1, **i32)
cu_k = torch.arange(0, (B + 1) * ctx, ctx, **i32)
g = torch.Generator(device=dev).manual_seed(0)
rnd = lambda *s: torch.randn(s, generator=g, device=dev, dtype=torch.float32).to(dt)
q, k_new, v_new = rnd(B, H, D), rnd(B, H, D), rnd(B, H, D)
out = torch.empty_like(q)
pos = torch.full((B,), ctx - 1, **i32)
cs = build_cos_sin(LLAVA_1_5_7B, dt, dev)
slots = torch.tensor([tables[r][(ctx - 1) // bs] * bs + (ctx - 1) % bs for r in range(B)], **i32)
x = rnd(B, H * D)
wqkv = (torch.randn((3 * H * D, H * D), generator=g, device=dev, dtype=torch.float32) * 0.02).to(dt)
slabs = torch.empty(hip_gemm.xreg_workspace_floats(B, 3 * H * D, H * D), dtype=torch.float32, device=dev)
n_slabs = hip_gemm.linear_decode_partial_xreg(x, hip_gemm.pack_weight_xreg(wqkv), 3 * H * D, slabs)
pool_elems = L * 2 * n_blocks * bs * H * D
pool_bytes = pool_elems * 2


def time_pool(pool, reps=5):
    def sweep():
        for l in range(L):
            decode_attention_fused(out, q, k_new, v_new, pool[l, 0], pool[l, 1], pos, cs, slots, cu_q, cu_k,
                                   table, cu_b, ctx, 1 / math.sqrt(D), 1, slabs, n_slabs)
    sweep(); sweep()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); sweep(); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3 / L)
    ts.sort()
    return ts[len(ts) // 2], ts[0], ts[-1]


def report(name, pool):
    med, lo, hi = time_pool(pool)
    gbs = 2 * (2 * H * D * ctx * B) / med / 1e3
    print(f"{name:58s} ptr %% 2MiB = {pool.data_ptr() % (2 << 20):8d}  {med:7.2f} us  ({lo:.2f} .. {hi:.2f})  {gbs:7.1f} GB/s", flush=True)


print(f"B = {B}, L = {L}, pool {pool_bytes / 2**30:.2f} GiB, {n_blocks} blocks of {bs * H * D * 2 // 1024} KiB (K) + the same (V)")
half = n_blocks * bs * H * D * 2          # bytes of one layer's K (or V) pool


class Views:
    """K of layer l at off + l * (2 * half + 2 * gap), V behind it at + half + gap: what pool[l, 0] / pool[l, 1] would be
    with `gap` spare bytes after every K / V pool."""
    def __init__(self, buf, off, gap):
        self.kv = []
        for l in range(L):
            k0 = off + l * 2 * (half + gap)
            k = buf[k0:k0 + half].view(dt).view(n_blocks, bs, H, D)
            v = buf[k0 + half + gap:k0 + 2 * half + gap].view(dt).view(n_blocks, bs, H, D)
            self.kv.append((k, v))
        self.ptr = buf.data_ptr() + off
    def __getitem__(self, idx):
        return self.kv[idx[0]][idx[1]]
    def data_ptr(self):
        return self.ptr


big = torch.empty(pool_bytes + (96 << 20), dtype=torch.uint8, device=dev)
big.view(torch.int16).random_(-16000, 16000)
if os.environ.get("SWEEP", "1") == "1":
    for off in (0, 64, 128, 256, 512, 1024, 2048, 4096, 4096 + 256, 8192, 8192 + 256, 65536 + 256, 0):
        report(f"offset {off}, gap 0", Views(big, off, 0))
    for gap in (256, 1024, 4096, 4096 + 256, 65536, 65536 + 256, (1 << 20) + 256):
        for off in (0, 256):
            report(f"offset {off}, gap {gap} after every K / V pool", Views(big, off, gap))
