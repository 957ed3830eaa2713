#!/usr/bin/env python3
"""Timing experiment: the launches between two attention launches of a 7B decode layer (o, add+norm,
gate|up, silu*mul, down, add+norm, qkv) with the LDS-slice packed GEMM everywhere (today) vs the
activations-in-registers GEMM for gate|up and down, with and without the silu*mul launch (what a
fused gate|up epilogue would remove).  Cold weights, hipGraph.  env OPTS="xreg_stagger=0" etc. sets debug options."""
import os, sys, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from hydrainfer_amd import _lib
from hydrainfer_amd._C.kernel import activation, gemm, norm

dev, dt = torch.device("cuda:0"), torch.bfloat16
M = int(os.environ.get("M", "32"))
hid, inter = 4096, 11008
LAYERS, N = 4, 16
g = torch.Generator(device=dev).manual_seed(0)
rnd = lambda *s, sc=1.0: (torch.randn(s, device=dev, generator=g) * sc).to(dt)
W = []
for _ in range(LAYERS):
    o, gu, dn, qkv = rnd(hid, hid, sc=.02), rnd(2 * inter, hid, sc=.02), rnd(hid, inter, sc=.02), rnd(3 * hid, hid, sc=.02)
    W.append(dict(po=gemm.pack_weight(o), pgu=gemm.pack_weight(gu), pdn=gemm.pack_weight(dn), pqkv=gemm.pack_weight(qkv),
                  xgui=gemm.pack_weight_xreg(gu, interleave_halves=True), xdn=gemm.pack_weight_xreg(dn), xqkv=gemm.pack_weight_xreg(qkv), n1=rnd(hid), n2=rnd(hid)))
    del o, gu, dn, qkv
attn, h = rnd(M, hid), rnd(M, hid)
ws = torch.empty(gemm.workspace_floats(M, 2 * inter, hid), dtype=torch.float32, device=dev)
x, x2 = torch.empty_like(h), torch.empty_like(h)
act = torch.empty((M, inter), dtype=dt, device=dev)
for o_ in [o for o in os.environ.get("OPTS", "").split(",") if o]:
    k_, v_ = o_.split("=")
    assert _lib.lib().hx_debug_set_option(k_.encode(), int(v_)) == 0, o_


xf = torch.empty(gemm.fragment_major_elems(M, hid), dtype=dt, device=dev)
actf = torch.empty(gemm.fragment_major_elems(M, inter), dtype=dt, device=dev)


sync_areas = torch.zeros((2 * N, gemm.XREG_SYNC_WORDS), dtype=torch.int32, device=dev)
sync_i = [0]
ws2 = torch.empty_like(ws)


def next_sync():
    sync_i[0] += 1
    return sync_areas[sync_i[0] - 1]


def layer(w, mode):
    """packed: round-2-start launches (7 between two attention launches).  fused: gate|up + silu*mul one
    launch, down and qkv on the activations-in-registers kernel (6).  norm_fused: the two add+norm
    launches folded into the gate|up and qkv launches as well (4) — the product path."""
    s = gemm.linear_decode_partial_packed(attn, w["po"], hid, ws)
    if mode == "packed":
        norm.add_rms_norm_slabs(x, h, ws, s, w["n1"], 1e-5)
        s = gemm.linear_decode_partial_packed(x, w["pgu"], 2 * inter, ws)
        a = activation.silu_and_mul_slabs(ws, s, M, inter, dt)
        s = gemm.linear_decode_partial_packed(a, w["pdn"], hid, ws)
        norm.add_rms_norm_slabs(x2, h, ws, s, w["n2"], 1e-5)
        gemm.linear_decode_partial_packed(x2, w["pqkv"], 3 * hid, ws)
        return
    if mode == "fused":
        norm.add_rms_norm_slabs(xf, h, ws, s, w["n1"], 1e-5, fragment_major=True)
        gemm.gate_up_silu_xreg(xf, w["xgui"], inter, actf, frag_shape=(M, hid))
    else:
        gemm.norm_gate_up_silu_xreg(h, ws, s, w["n1"], 1e-5, xf, w["xgui"], inter, actf, next_sync())
    s = gemm.linear_decode_partial_xreg(actf, w["xdn"], hid, ws, frag_shape=(M, inter))
    if mode == "fused":
        norm.add_rms_norm_slabs(xf, h, ws, s, w["n2"], 1e-5, fragment_major=True)
        gemm.linear_decode_partial_xreg(xf, w["xqkv"], 3 * hid, ws, frag_shape=(M, hid))
    else:
        gemm.norm_linear_decode_xreg(h, ws, s, w["n2"], 1e-5, xf, w["xqkv"], 3 * hid, ws2, next_sync())


def timeit(mode, reps=7):
    def body():
        sync_i[0] = 0
        sync_areas.zero_()
        for i in range(N):
            layer(W[i % LAYERS], mode)
    s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        body()
    torch.cuda.current_stream().wait_stream(s)
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr):
        body()
    ts = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); gr.replay(); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / N * 1e3)
    return statistics.median(ts)


for mode in ("packed", "fused", "norm_fused", "fused", "norm_fused"):
    print(f"{mode:11s}: {timeit(mode):6.1f} us per layer (without attention)", flush=True)
