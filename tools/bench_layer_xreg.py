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
                  xgu=gemm.pack_weight_xreg(gu), xgui=gemm.pack_weight_xreg(gu, interleave_halves=True), xdn=gemm.pack_weight_xreg(dn), n1=rnd(hid), n2=rnd(hid)))
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


def layer(w, mode):
    """packed: today's launches.  xreg: gate|up and down on the activations-in-registers kernel,
    silu*mul still a launch.  fused: gate|up + silu*mul in one launch (the real product path)."""
    s = gemm.linear_decode_partial_packed(attn, w["po"], hid, ws)
    if mode == "packed":
        norm.add_rms_norm_slabs(x, h, ws, s, w["n1"], 1e-5)
        s = gemm.linear_decode_partial_packed(x, w["pgu"], 2 * inter, ws)
        a = activation.silu_and_mul_slabs(ws, s, M, inter, dt)
        s = gemm.linear_decode_partial_packed(a, w["pdn"], hid, ws)
    else:
        norm.add_rms_norm_slabs(xf, h, ws, s, w["n1"], 1e-5, fragment_major=True)
        if mode == "xreg":
            s = gemm.linear_decode_partial_xreg(xf, w["xgu"], 2 * inter, ws, frag_shape=(M, hid))
            a = activation.silu_and_mul_slabs(ws, s, M, inter, dt, fragment_major=True)
        else:
            gemm.gate_up_silu_xreg(xf, w["xgui"], inter, actf, frag_shape=(M, hid))
            a = actf
        s = gemm.linear_decode_partial_xreg(a, w["xdn"], hid, ws, frag_shape=(M, inter))
    norm.add_rms_norm_slabs(x2, h, ws, s, w["n2"], 1e-5)
    gemm.linear_decode_partial_packed(x2, w["pqkv"], 3 * hid, ws)


def timeit(mode, reps=7):
    body = lambda: [layer(W[i % LAYERS], mode) for i in range(N)]
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


for mode in ("packed", "xreg", "fused", "packed", "fused"):
    print(f"{mode:11s}: {timeit(mode):6.1f} us per layer (without attention)", flush=True)
