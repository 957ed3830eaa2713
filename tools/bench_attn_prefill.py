#!/usr/bin/env python3
"""Times the prefill / dense attention kernel on the shapes of the LLaVA-1.5 path, for one and two
query row blocks per wave (hx_debug_set_option fwd_row_blocks)."""
import math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from hydrainfer_amd import _lib
from hydrainfer_amd._C.kernel.flash_attn import mha_varlen_fwd

dev = torch.device("cuda:0")
dt = torch.bfloat16
g = torch.Generator(device=dev).manual_seed(0)


def rnd(*shape):
    return torch.randn(shape, generator=g, device=dev, dtype=torch.float32).to(dt)


def timeit(fn, reps=30):
    for _ in range(5):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


def paged(B, n, kv, H=32, D=128, bs=16):
    nb = (kv + bs - 1) // bs
    kc, vc = rnd(B * nb, bs, H, D), rnd(B * nb, bs, H, D)
    q = rnd(B * n, H, D)
    out = torch.empty_like(q)
    perm = torch.randperm(B * nb, generator=g, device=dev).to(torch.int32)
    cu_b = torch.arange(0, (B + 1) * nb, nb, dtype=torch.int32, device=dev)
    cu_q = torch.arange(0, (B + 1) * n, n, dtype=torch.int32, device=dev)
    cu_k = torch.arange(0, (B + 1) * kv, kv, dtype=torch.int32, device=dev)
    flops = 4 * H * D * B * sum(kv - n + i + 1 for i in range(n))
    return (lambda: mha_varlen_fwd(out, q, kc, vc, cu_q, cu_k, perm, cu_b, None, n, kv, 1 / math.sqrt(D), 0, -1, 0, 0)), flops


def dense(n_img, n=577, H=16, D=64):
    q, k, v = rnd(n_img * n, H, D), rnd(n_img * n, H, D), rnd(n_img * n, H, D)
    out = torch.empty_like(q)
    cu = torch.arange(0, (n_img + 1) * n, n, dtype=torch.int32, device=dev)
    flops = 4 * H * D * n_img * n * n
    return (lambda: mha_varlen_fwd(out, q, k, v, cu, cu, None, None, None, n, n, 1 / math.sqrt(D), 0, -1, -1, 0)), flops


cases = {"prefill 4x704": paged(4, 704, 704), "prefill 1x704": paged(1, 704, 704),
         "chunk 1x2048 of 2048": paged(1, 2048, 2048), "chunk 3x683": paged(3, 683, 704),
         "clip 1x577 d64": dense(1), "clip 8x577 d64": dense(8)}
for name, (fn, flops) in cases.items():
    row = []
    for xcd in (0, 1):
        _lib.check(_lib.lib().hx_debug_set_option(b"fwd_xcd_remap", xcd), "opt")
        _lib.lib().hx_debug_set_option(b"fwd_row_blocks", 0); _lib.lib().hx_debug_set_option(b"fwd_key_units", 0)
        us = timeit(fn)
        row.append(f"auto xcd={xcd}: {us:6.1f}us {flops / us / 1e6:5.0f}TF")
    _lib.lib().hx_debug_set_option(b"fwd_xcd_remap", 1)
    for rows in (1, 2):
        for keys in (1, 2):
            _lib.check(_lib.lib().hx_debug_set_option(b"fwd_row_blocks", rows), "opt")
            _lib.check(_lib.lib().hx_debug_set_option(b"fwd_key_units", keys), "opt")
            us = timeit(fn)
            row.append(f"r{rows}k{keys}: {us:6.1f}us {flops / us / 1e6:5.0f}TF")
    print(f"{name:22s}", " | ".join(row))
_lib.lib().hx_debug_set_option(b"fwd_row_blocks", 0)
_lib.lib().hx_debug_set_option(b"fwd_key_units", 0)

# mixed step of a chunked-prefill engine: one long chunk + many decode rows in ONE launch
def mixed(n_dec, chunk, ctx=832, H=32, D=128, bs=16):
    q_lens = [chunk] + [1] * n_dec
    kv_lens = [chunk] + [ctx] * n_dec
    nbs = [(k + bs - 1) // bs for k in kv_lens]
    kc, vc = rnd(sum(nbs), bs, H, D), rnd(sum(nbs), bs, H, D)
    q = rnd(sum(q_lens), H, D)
    out = torch.empty_like(q)
    perm = torch.randperm(sum(nbs), generator=g, device=dev).to(torch.int32)
    i32 = lambda x: torch.tensor(x, dtype=torch.int32, device=dev)
    cu = lambda xs: i32([0] + list(torch.tensor(xs).cumsum(0)))
    cu_q, cu_k, cu_b = cu(q_lens), cu(kv_lens), cu(nbs)
    return lambda: mha_varlen_fwd(out, q, kc, vc, cu_q, cu_k, perm, cu_b, None, chunk, max(kv_lens), 1 / math.sqrt(D), 0, -1, 0, 0)

print("--- mixed launches (one prefill chunk + decode rows)")
for n_dec, chunk in ((0, 2017), (31, 2017), (31, 704), (0, 704), (31, 64)):
    print(f"decode rows {n_dec:2d} + chunk {chunk:4d}: {timeit(mixed(n_dec, chunk)):7.1f} us")

print("--- long contexts (chunk q of kv)")
for q_, kv_, B_ in ((2048, 4096, 1), (512, 8192, 1), (2048, 2048, 4), (128, 4096, 8)):
    fn, fl = paged(B_, q_, kv_)
    us = timeit(fn, reps=10)
    print(f"B={B_} q={q_} kv={kv_}: {us:8.1f} us {fl / us / 1e6:6.0f} TF/s")
