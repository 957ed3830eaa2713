#!/usr/bin/env python3
"""Decode attention on ragged batches: the static (head, sequence) grid against the RANKED form (length-ranked snake
order over the CUs, attn_decode.hip), one process, interleaved rounds, every variant a hipGraph of launches walking several layers'
caches (no host gaps, no Infinity-Cache flattery).  GPU only.
    python tools/bench_attn_ragged.py [--heads 32] [--sweep]"""
import argparse
import math
import os
import statistics
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import torch

from hydrainfer_amd import _lib
from hydrainfer_amd._C.kernel.flash_attn import decode_attention_fused, decode_rank
from hydrainfer_amd.model.runner import ragged_contexts


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--heads", type=int, default=32)
    ap.add_argument("--rounds", type=int, default=5)
    ap.add_argument("--launches", type=int, default=16)
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    dt = torch.bfloat16
    B, H, D, bs = args.batch, args.heads, 128, 16
    lib = _lib.lib()
    g = torch.Generator(device=dev).manual_seed(0)
    n_layers = 4
    max_blocks = B * 60 + 8
    pool = torch.randn((n_layers, 2, max_blocks, bs, H, D), generator=g, device=dev, dtype=torch.float32).to(dt)
    q = torch.randn((B, H, D), generator=g, device=dev, dtype=torch.float32).to(dt)
    scale = 1 / math.sqrt(D)

    def even(lens):
        t = sum(lens)
        return [t // B + (1 if i < t % B else 0) for i in range(B)]

    sets = {"all 832": [832] * B, "uniform 64..959": ragged_contexts("uniform", B), "bimodal 130/830": ragged_contexts("bimodal", B)}
    sets["even(uniform)"] = even(sets["uniform 64..959"])
    sets["even(bimodal)"] = even(sets["bimodal 130/830"])
    one_long = [200] * B
    one_long[7] = 959
    sets["one long among short"] = one_long

    def meta(lens):
        nb = [(l + bs - 1) // bs for l in lens]
        perm = torch.randperm(max_blocks, generator=g, device=dev).to(torch.int32)
        cu_b = torch.tensor([0] + list(torch.tensor(nb).cumsum(0)), dtype=torch.int32, device=dev)
        cu_k = torch.tensor([0] + list(torch.tensor(lens).cumsum(0)), dtype=torch.int32, device=dev)
        cu_q = torch.arange(0, B + 1, dtype=torch.int32, device=dev)
        return perm[: int(cu_b[-1])].contiguous(), cu_b, cu_k, cu_q

    inv = 1.0 / torch.pow(10000.0, torch.arange(0, D, 2, dtype=torch.float) / D)
    fr = torch.einsum("i,j->ij", torch.arange(4096, dtype=torch.float), inv)
    cos_sin = torch.cat([fr.cos()[:, None, :], fr.sin()[:, None, :]], dim=1).to(dt).to(dev)
    k_new = torch.randn((B, H, D), generator=g, device=dev, dtype=torch.float32).to(dt)
    v_new = torch.randn((B, H, D), generator=g, device=dev, dtype=torch.float32).to(dt)

    def graph_of(lens, m, ranked):
        """The fused launch (RoPE + cache append + attention) of the decode step, with or without the rank descriptor."""
        perm, cu_b, cu_k, cu_q = m
        out = torch.empty_like(q)
        pos = torch.tensor([l - 1 for l in lens], dtype=torch.int32, device=dev)
        slots = torch.stack([perm[int(cu_b[i]) + (l - 1) // bs] * bs + (l - 1) % bs for i, l in enumerate(lens)]).to(torch.int32)
        rank = decode_rank(cu_k) if ranked else None

        def go(i):
            decode_attention_fused(out, q, k_new, v_new, pool[i % n_layers, 0], pool[i % n_layers, 1], pos, cos_sin, slots, cu_q, cu_k,
                                   perm, cu_b, 960, scale, 0, rank_desc=rank)
        s = torch.cuda.Stream(device=dev)
        s.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(s):
            go(0)
        torch.cuda.current_stream(dev).wait_stream(s)
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr):
            for i in range(args.launches):
                go(i)
        return gr, out

    def opt(name, v):
        assert lib.hx_debug_set_option(name.encode(), v) == 0, name

    variants = [("static", False), ("ranked", True)]
    for name, lens in sets.items():
        m = meta(lens)
        nbytes = 2 * (2 * H * D * sum(lens) + 2 * B * H * D) + 4 * sum((l + bs - 1) // bs for l in lens)
        graphs = {vname: graph_of(lens, m, ranked) for vname, ranked in variants}
        for gr, _ in graphs.values():
            gr.replay()
        torch.cuda.synchronize()
        ref = graphs["static"][1].float()
        res = {v: [] for v, _ in variants}
        for _ in range(args.rounds):
            for vname, _o in variants:
                gr, out = graphs[vname]
                gr.replay(); torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(); gr.replay(); e1.record(); torch.cuda.synchronize()
                res[vname].append(e0.elapsed_time(e1) / args.launches * 1e3)
        print(f"--- {name}: sum ctx {sum(lens)}, {nbytes / 1e6:.1f} MB per launch")
        for vname, _o in variants:
            med = statistics.median(res[vname])
            err = (graphs[vname][1].float() - ref).abs().max().item()
            print(f"  {vname:28s} {med:7.2f} us  {nbytes / med / 1e3:7.1f} GB/s  frac {nbytes / med / 1e3 / 8000:.3f}   max |d| vs static {err:.2e}")


if __name__ == "__main__":
    main()
