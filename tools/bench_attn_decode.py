#!/usr/bin/env python3
"""A/B micro-benchmark of the decode-attention kernel at the BASELINE shape (one process,
interleaved rounds — cdna_hip_programming.md §5.4 rule 24).  GPU only."""
import argparse
import math
import statistics
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import torch

from hydrainfer_amd import _lib
from hydrainfer_amd._C.kernel.flash_attn import mha_varlen_fwd


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--heads", type=int, default=32)
    ap.add_argument("--kv-heads", type=int, default=0, help="0 = same as --heads (MHA)")
    ap.add_argument("--ctx", type=int, default=832)
    ap.add_argument("--dtype", default="bf16")
    ap.add_argument("--rounds", type=int, default=7)
    ap.add_argument("--iters", type=int, default=20)
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    dt = torch.bfloat16 if args.dtype == "bf16" else torch.float16
    B, H, D, bs = args.batch, args.heads, 128, 16
    HK = args.kv_heads or H
    nb_seq = (args.ctx + bs - 1) // bs
    n_blocks = B * nb_seq
    g = torch.Generator(device=dev).manual_seed(0)
    # several layers' worth of cache so successive launches do not hit in the 256 MiB L3
    n_layers = 4
    pool = torch.randn((n_layers, 2, n_blocks, bs, HK, D), generator=g, device=dev, dtype=torch.float32).to(dt)
    perm = torch.randperm(n_blocks, generator=g, device=dev).to(torch.int32)
    cu_b = torch.arange(0, (B + 1) * nb_seq, nb_seq, dtype=torch.int32, device=dev)
    cu_q = torch.arange(0, B + 1, dtype=torch.int32, device=dev)
    cu_k = torch.arange(0, (B + 1) * args.ctx, args.ctx, dtype=torch.int32, device=dev)
    q = torch.randn((B, H, D), generator=g, device=dev, dtype=torch.float32).to(dt)
    out = torch.empty_like(q)
    scale = 1 / math.sqrt(D)
    nbytes = 2 * (2 * HK * D * args.ctx * B + 2 * B * H * D) + 4 * B * nb_seq
    lib = _lib.lib()

    def run(layer, splits):
        mha_varlen_fwd(out, q, pool[layer, 0], pool[layer, 1], cu_q, cu_k, perm, cu_b, None, 1, args.ctx,
                       scale, 0.0, -1, 0, splits)

    variants = [(w, nt, s) for w in (4, 8) for nt in (0, 1) for s in (1, 2)]
    res = {v: [] for v in variants}
    for rnd in range(args.rounds):
        for v in variants:
            lib.hx_debug_set_option(b"decode_waves", v[0])
            lib.hx_debug_set_option(b"decode_nt", v[1])
            run(0, v[2]); torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for i in range(args.iters):
                run(i % n_layers, v[2])
            e1.record(); torch.cuda.synchronize()
            res[v].append(e0.elapsed_time(e1) / args.iters * 1e3)
    print(f"shape B={B} H={H} D={D} ctx={args.ctx} {args.dtype}: {nbytes/1e6:.1f} MB algorithmic per launch")
    for v in variants:
        med, mn = statistics.median(res[v]), min(res[v])
        print(f"waves={v[0]} nt={v[1]} splits={v[2]}: median {med:7.2f} us ({nbytes/med/1e3:7.1f} GB/s)  min {mn:7.2f} us ({nbytes/mn/1e3:7.1f} GB/s)")
    # reference: device-to-device copy of the same bytes (reads + writes => 2x traffic)
    src = pool[0].reshape(-1)[: nbytes // 2]
    dst = torch.empty_like(src)
    ts = []
    for _ in range(args.rounds):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(args.iters):
            dst.copy_(src)
        e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / args.iters * 1e3)
    med = statistics.median(ts)
    print(f"torch copy of the same bytes: {med:.2f} us => {2*nbytes/med/1e3:.1f} GB/s read+write")


if __name__ == "__main__":
    main()
