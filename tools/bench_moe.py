#!/usr/bin/env python3
"""SURVEY 8 row a12 (MoE routing / permutation; off the LLaVA path): time per launch and HBM rate of the ops at
DeepSeek-V2-Lite-like and Mixtral-like sizes.  Algorithmic bytes: permute reads every token row once and writes it topk
times; unpermute reads topk rows (+ probs) and writes one; sum_out the same without the map."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from hydrainfer_amd._C.kernel import moe

dev = torch.device("cuda:0")


def timeit(fn, reps=20):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


for name, (n, dim, n_exp, topk, dt) in {"4096 tokens x 7168, 256 experts, top 8, bf16": (4096, 7168, 256, 8, torch.bfloat16),
                                        "2048 tokens x 4096, 8 experts, top 2, fp16": (2048, 4096, 8, 2, torch.float16),
                                        "32 tokens x 7168, 256 experts, top 8, bf16 (decode)": (32, 7168, 256, 8, torch.bfloat16)}.items():
    g = torch.Generator(device=dev).manual_seed(0)
    tokens = torch.randn((n, dim), device=dev, generator=g).to(dt)
    logits = torch.randn((n, n_exp), device=dev, generator=g)
    w = torch.empty((n, topk), device=dev); idx = torch.empty((n, topk), dtype=torch.int32, device=dev)
    es = tokens.element_size()
    rows = []
    rows.append(("topk_softmax", timeit(lambda: moe.topk_softmax(logits, w, idx)), n * n_exp * 4 + n * topk * 8))
    moe.topk_softmax(logits, w, idx)
    permuted, rmap = moe.permute_with_index_map(tokens, idx)
    rows.append(("permute_with_index_map (sort + map + copy)", timeit(lambda: moe.permute_with_index_map(tokens, idx)), n * dim * es * (1 + topk)))
    rows.append(("permute (copy only)", timeit(lambda: moe._permute(tokens, rmap, topk, n * topk)), n * dim * es * (1 + topk)))
    probs = w.to(dt)
    rows.append(("unpermute_with_index_map", timeit(lambda: moe.unpermute_with_index_map(permuted, rmap, probs)), n * dim * es * (1 + topk)))
    out = torch.empty((n, dim), dtype=dt, device=dev)
    inp = permuted.view(n, topk, dim)
    rows.append(("sum_out", timeit(lambda: moe.sum_out(inp, out)), n * dim * es * (1 + topk)))
    print(name)
    for what, us, b in rows:
        print(f"  {what:45s} {us:8.1f} us  {b / us / 1e3:8.1f} GB/s  ({b / 1e6:.1f} MB)")
