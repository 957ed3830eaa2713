#!/usr/bin/env python3
"""What separates the packed decode GEMM from a pure read of its weights: time the four 7B shapes
(cold weights, hipGraph) with parts of the kernel switched off (option gemm_slab_nt bits:
2 = no slab stores, 4 = no x loads, 8 = no LDS reads / MFMA)."""
import os, sys, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from hydrainfer_amd import _lib
from hydrainfer_amd._C.kernel import gemm

dev, dt = torch.device("cuda:0"), torch.bfloat16
M = int(os.environ.get("M", "32"))


def graph_time(fn, n_inner, reps=7):
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
        ts.append(e0.elapsed_time(e1) / n_inner * 1e3)
    return statistics.median(ts)


# warm the clocks
_w = torch.randn((8192, 8192), device=dev, dtype=dt)
for _ in range(200):
    _w @ _w
torch.cuda.synchronize()
sink = torch.zeros(4, dtype=torch.float32, device=dev)


def stream_time(n_bytes, regions):
    l = _lib.lib()
    def body():
        for r in regions:
            _lib.check(l.hx_debug_stream_read(r.data_ptr(), n_bytes, 0, 0, 8, 1, 1024, sink.data_ptr(), _lib.current_stream()), "stream")
    return graph_time(body, len(regions))


masks = [0, 2, 14]
print("0 = the kernel, 2 = no slab stores, 14 = no slab stores, no x loads, no LDS reads / MFMA (compile-time variants)")
for name, (N, K) in {"qkv": (12288, 4096), "o": (4096, 4096), "gate_up": (22016, 4096), "down": (4096, 11008)}.items():
    nc = 6
    pk = [gemm.pack_weight((torch.randn((N, K), device=dev) * 0.02).to(dt)) for _ in range(nc)]
    x = torch.randn((M, K), device=dev).to(dt)
    b = torch.empty(gemm.workspace_floats(M, N, K), dtype=torch.float32, device=dev)
    row = []
    for m in masks:
        assert _lib.lib().hx_debug_set_option(b"gemm_slab_nt", m) == 0
        t = graph_time(lambda: [gemm.linear_decode_partial_packed(x, pk[i % nc], N, b) for i in range(12)], 12)
        row.append(f"{m}:{t:6.2f}")
    _lib.lib().hx_debug_set_option(b"gemm_slab_nt", 0)
    nb = N * K * 2 // 8192 * 8192
    row.append(f"pure read:{stream_time(nb, [p_.view(torch.uint8) for p_ in pk] * 2):6.2f}")
    print(f"{name:8s} N={N:6d} K={K:6d} ({N*K*2/1e6:6.1f} MB): " + "  ".join(row), flush=True)
    del pk
