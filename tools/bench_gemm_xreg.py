#!/usr/bin/env python3
"""Decode GEMM, activations-in-registers kernel (gemm_xreg_kernel) vs the packed LDS-slice kernel
(gemm_packed_kernel) vs a pure read of the same bytes: correctness against an fp32 torch product and
time per launch (cold weights, hipGraph).  M=32 by default; env M, MODEL=7b|13b."""
import os, sys, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from hydrainfer_amd import _lib
from hydrainfer_amd._C.kernel import gemm

dev, dt = torch.device("cuda:0"), torch.bfloat16
M = int(os.environ.get("M", "32"))
hid, inter = (4096, 11008) if os.environ.get("MODEL", "7b") == "7b" else (5120, 13824)
sink = torch.zeros(4, dtype=torch.float32, device=dev)


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


opts = [o for o in os.environ.get("OPTS", "").split(",") if o]
for o in opts:
    k, v = o.split("=")
    assert _lib.lib().hx_debug_set_option(k.encode(), int(v)) == 0, o
tot = {"packed": 0.0, "xreg": 0.0, "read": 0.0}
for name, (N, K) in {"qkv": (3 * hid, hid), "o": (hid, hid), "gate_up": (2 * inter, hid), "down": (hid, inter)}.items():
    nc = 6
    ws = [(torch.randn((N, K), device=dev) * 0.02).to(dt) for _ in range(nc)]
    pk = [gemm.pack_weight(w) for w in ws]
    px = [gemm.pack_weight_xreg(w) for w in ws]
    x = torch.randn((M, K), device=dev).to(dt)
    a = torch.empty(gemm.workspace_floats(M, N, K), dtype=torch.float32, device=dev)
    b = torch.empty(max(gemm.xreg_workspace_floats(M, N, K), 1), dtype=torch.float32, device=dev)
    x_rm = x
    fs = None
    if os.environ.get("XFRAG", "1") == "1":   # x fragment-major, as the decode step hands it over
        x, fs = gemm.to_fragment_major(x), (M, K)
    sb = gemm.linear_decode_partial_xreg(x, px[0], N, b, frag_shape=fs)
    ref = x_rm.float() @ ws[0].float().t()
    got = b[: sb * M * N].view(sb, M, N).sum(0)
    err = (got - ref).abs().max().item() / ref.abs().max().item()
    sb2 = gemm.linear_decode_partial_xreg(x, px[0], N, b, frag_shape=fs)
    same = torch.equal(got, b[: sb2 * M * N].view(sb2, M, N).sum(0))
    t3 = same_rm = None
    if os.environ.get("ROW_MAJOR", "0") == "1" and M <= 32:      # the same kernel over the row-major tensor (RM = 1)
        assert _lib.lib().hx_debug_set_option(b"xreg_row_major", 1) == 0
        sb3 = gemm.linear_decode_partial_xreg(x, ws[0], N, b, frag_shape=fs)
        same_rm = sb3 == sb and torch.equal(got, b[: sb3 * M * N].view(sb3, M, N).sum(0))
        t3 = graph_time(lambda: [gemm.linear_decode_partial_xreg(x, ws[i % nc], N, b, frag_shape=fs) for i in range(12)], 12)
        assert _lib.lib().hx_debug_set_option(b"xreg_row_major", 0) == 0
    del ws
    t0 = graph_time(lambda: [gemm.linear_decode_partial_packed(x_rm, pk[i % nc], N, a) for i in range(12)], 12)
    t1 = graph_time(lambda: [gemm.linear_decode_partial_xreg(x, px[i % nc], N, b, frag_shape=fs) for i in range(12)], 12)
    nb = N * K * 2 // 8192 * 8192
    l = _lib.lib()
    t2 = float("nan") if not hasattr(l, "hx_debug_stream_read") else graph_time(lambda: [_lib.check(l.hx_debug_stream_read(px[i % nc].data_ptr(), nb, 0, 0, 8, 1, 1024, sink.data_ptr(),
                                                               _lib.current_stream()), "s") for i in range(12)], 12)
    tot["packed"] += t0; tot["xreg"] += t1; tot["read"] += t2
    print(f"{name:8s} N={N:6d} K={K:6d}: packed {t0:6.2f} us | xreg {t1:6.2f} us {N*K*2/t1/1e6:5.2f} TB/s slabs={sb} "
          f"rel.err {err:.1e} repeatable {same} | pure read {t2:6.2f} us"
          + (f" | row-major W {t3:6.2f} us, bit-identical {same_rm}" if t3 is not None else ""), flush=True)
    del pk, px
print(f"layer: packed {tot['packed']:.1f} us, xreg {tot['xreg']:.1f} us, pure read {tot['read']:.1f} us   opts={opts}")
