#!/usr/bin/env python3
"""Decode GEMM shapes of LLaVA-1.5-7B (M=32): library GEMM (torch.matmul -> hipBLASLt) vs the
weight-streaming HIP kernel (linear_decode), timed inside a hipGraph (no host launch cost),
weights rotated so they never sit in the 256 MiB Infinity Cache.  Also checks the results."""
import os, sys, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from hydrainfer_amd import _lib
from hydrainfer_amd._C.kernel.gemm import linear_decode


def graph_time(fn, n_inner, reps=5):
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
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
        ts.append(e0.elapsed_time(e1) / n_inner * 1e3)
    return statistics.median(ts)


def main():
    dev = torch.device("cuda:0")
    dt = torch.bfloat16
    M = int(os.environ.get("M", "32"))
    rows_opts = [int(v) for v in os.environ.get("WGS", "512").split(",")]
    shapes = {"qkv": (12288, 4096), "o": (4096, 4096), "gate_up": (22016, 4096), "down": (4096, 11008),
              "lm_head": (32064, 4096)}
    n_copies = 6
    lib = _lib.lib()
    for name, (N, K) in shapes.items():
        ws = [(torch.randn((N, K), device=dev, dtype=torch.float32) * 0.02).to(dt) for _ in range(n_copies)]
        x = torch.randn((M, K), device=dev, dtype=torch.float32).to(dt)
        ref = torch.matmul(x.float(), ws[0].float().t())
        got = linear_decode(x, ws[0])
        err = (got.float() - ref).abs().max().item()
        lib_err = (torch.matmul(x, ws[0].t()).float() - ref).abs().max().item()
        outs = [torch.empty((M, N), dtype=dt, device=dev) for _ in range(n_copies)]

        def lib_fn():
            for i in range(12):
                torch.matmul(x, ws[i % n_copies].t(), out=outs[i % n_copies])
        t_lib = graph_time(lib_fn, 12)
        line = f"{name:8s} N={N:6d} K={K:6d}: lib {t_lib:7.2f} us {N*K*2/t_lib/1e3:7.1f} GB/s (err {lib_err:.3f}) |"
        for rows in rows_opts:
            lib.hx_debug_set_option(b"gemm_wg_target", rows)

            def hip_fn():
                for i in range(12):
                    linear_decode(x, ws[i % n_copies], out=outs[i % n_copies])
            t = graph_time(hip_fn, 12)
            line += f" hip[wgs={rows}] {t:7.2f} us {N*K*2/t/1e3:7.1f} GB/s |"
        print(line + f" hip err {err:.3f}")
        del ws


main()
