#!/usr/bin/env python3
"""What the decode attention kernel could reach: its exact read pattern (32 sequences x 32 heads, paged cache
with block_size 16, random page order, 256 B of every key / value row per head) with the arithmetic removed
(hx_debug_paged_read), next to the real kernel on the same cache, layers rotated (cold), hipGraph."""
import math, os, sys, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from hydrainfer_amd import _lib
from hydrainfer_amd._C.kernel.flash_attn import mha_varlen_fwd

dev, dt = torch.device("cuda:0"), torch.bfloat16
B, H, D, bs, L = 32, 32, 128, 16, 6
ctx = int(os.environ.get("CTX", "720"))
tiles = (ctx + bs - 1) // bs
n_blocks = B * tiles
pool = torch.randn((L, 2, n_blocks, bs, H, D), device=dev, dtype=torch.float32).to(dt)
perm = torch.randperm(n_blocks, device=dev).to(torch.int32)
sink = torch.zeros(4, dtype=torch.float32, device=dev)
l = _lib.lib()
page_bytes, row_bytes = bs * H * D * 2, H * D * 2
nbytes = 2 * B * tiles * bs * H * D * 2


def graph_time(fn, n, reps=7):
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
        ts.append(e0.elapsed_time(e1) / n * 1e3)
    return statistics.median(ts)


ident = torch.arange(n_blocks, dtype=torch.int32, device=dev)


def reads(hpw=1, waves=4, depth=2, splits=1, table=None, row_bytes=row_bytes):
    tb = perm if table is None else table
    for i in range(12):
        _lib.check(l.hx_debug_paged_read(pool[i % L, 0].data_ptr(), pool[i % L, 1].data_ptr(), tb.data_ptr(), B, H, tiles,
                                         page_bytes, row_bytes, hpw, waves, depth, splits, sink.data_ptr(),
                                         _lib.current_stream()), "paged_read")


q = torch.randn((B, H, D), device=dev).to(dt)
out = torch.empty_like(q)
cu_q = torch.arange(0, B + 1, dtype=torch.int32, device=dev)
cu_k = torch.arange(0, (B + 1) * ctx, ctx, dtype=torch.int32, device=dev)
cu_b = torch.arange(0, (B + 1) * tiles, tiles, dtype=torch.int32, device=dev)


def attn():
    for i in range(12):
        mha_varlen_fwd(out, q, pool[i % L, 0], pool[i % L, 1], cu_q, cu_k, perm, cu_b, None, 1, ctx, 1 / math.sqrt(D), 0.0, -1, 0, 1)


t_a = graph_time(attn, 12)
print(f"ctx {ctx}: {nbytes / 1e6:.0f} MB per launch; attn_decode_kernel (plain, no fused prologue) {t_a:.1f} us = {nbytes / t_a / 1e6:.2f} TB/s")
for opt in [o for o in os.environ.get("OPTS", "").split(",") if o]:
    k, v = opt.split("=")
    assert l.hx_debug_set_option(k.encode(), int(v)) == 0, opt
    t_o = graph_time(attn, 12)
    print(f"  with {opt}: {t_o:.1f} us = {nbytes / t_o / 1e6:.2f} TB/s")
    l.hx_debug_set_option(k.encode(), 0)
if os.environ.get("FUSED"):
    # the variant the decode graph runs: q / k / v from ONE fp32 slab of the qkv GEMM + RoPE + cache append + attention
    from hydrainfer_amd._C.kernel.flash_attn import decode_attention_fused
    from hydrainfer_amd.model.llama import LLAVA_1_5_7B, build_cos_sin
    cs = build_cos_sin(LLAVA_1_5_7B, dt, dev)
    slab = torch.randn((B, 3 * H * D), device=dev)
    pos = torch.full((B,), ctx - 1, dtype=torch.int32, device=dev)
    slots = (perm[cu_b[:-1].long() + (ctx - 1) // bs] * bs + (ctx - 1) % bs).to(torch.int32)

    def fused():
        for i in range(12):
            decode_attention_fused(out, q, q, q, pool[i % L, 0], pool[i % L, 1], pos, cs, slots, cu_q, cu_k, perm, cu_b, ctx,
                                   1 / math.sqrt(D), 1, slab, 1)
    for v in (0, 1, 0, 1):
        l.hx_debug_set_option(b"decode_hpw4", v)
        t_f = graph_time(fused, 12)
        print(f"  fused form, decode_hpw4={v}: {t_f:.1f} us = {nbytes / t_f / 1e6:.2f} TB/s")
    l.hx_debug_set_option(b"decode_hpw4", 0)
if os.environ.get("KERNEL_ONLY"):
    sys.exit(0)
print("pattern-only reads (heads per workgroup, waves, tiles in flight per wave, KV splits, page order):")
for hpw, waves, depth, splits in ((1, 4, 2, 1), (1, 4, 3, 1), (1, 4, 4, 1), (1, 8, 2, 1), (2, 4, 2, 1), (2, 4, 2, 2), (2, 4, 3, 2),
                                  (2, 8, 2, 2), (4, 4, 2, 1), (4, 4, 2, 4), (4, 4, 3, 4), (4, 8, 2, 4)):
    for name, tb in (("random", None), ("sequential", ident)):
        t = graph_time(lambda: reads(hpw, waves, depth, splits, tb), 12)
        print(f"  hpw {hpw} waves {waves} depth {depth} splits {splits} {name:10s}: {t:6.1f} us = {nbytes / t / 1e6:.2f} TB/s", flush=True)

print("adjacent heads by different waves of one workgroup (heads per workgroup, tile phases per head; 4 rows x 256 B per instruction):")
for hg, npf in ((1, 4), (2, 2), (2, 4), (4, 1), (4, 2), (4, 4), (8, 2), (2, 8)):
    for name, rb in (("token-major pages", row_bytes), ("head-major pages", 256)):
        t = graph_time(lambda: reads(-hg, hg * npf, 2, 1, None, rb), 12)
        print(f"  heads {hg} x phases {npf} ({hg * npf} waves, {32 // hg * 32} workgroups) {name:18s}: {t:6.1f} us = {nbytes / t / 1e6:.2f} TB/s", flush=True)
