#!/usr/bin/env python3
"""Decode chain (one launch) vs the seven separate launches it replaces, 7B / 13B layer shapes,
cold weights (LAYERS distinct weight sets cycled, > Infinity Cache), inside a hipGraph.
    python tools/bench_chain.py [7b|13b] [M]        env HX_CHAIN_R="o,gu,down,qkv" to sweep item sizes"""
import os, sys, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from hydrainfer_amd._C.kernel import activation, gemm, norm

dev, dt = torch.device("cuda:0"), torch.bfloat16
model = sys.argv[1] if len(sys.argv) > 1 else "7b"
M = int(sys.argv[2]) if len(sys.argv) > 2 else 32
hid, inter = (4096, 11008) if model == "7b" else (5120, 13824)
q_size, qkv_n = hid, 3 * hid
LAYERS = 4
g = torch.Generator(device=dev).manual_seed(0)
rnd = lambda *s, sc=1.0: (torch.randn(s, device=dev, generator=g) * sc).to(dt)
W = [dict(o=rnd(hid, q_size, sc=.02), gu=rnd(2 * inter, hid, sc=.02), dn=rnd(hid, inter, sc=.02),
          qkv=rnd(qkv_n, hid, sc=.02), n1=rnd(hid), n2=rnd(hid)) for _ in range(LAYERS)]
for w in W:
    w.update(po=gemm.pack_weight(w["o"]), pgu=gemm.pack_weight(w["gu"]), pdn=gemm.pack_weight(w["dn"]), pqkv=gemm.pack_weight(w["qkv"]))
attn, h0 = rnd(M, q_size), rnd(M, hid)
wbytes = 2 * (hid * q_size + 2 * inter * hid + hid * inter + qkv_n * hid)

ws = torch.empty(max(gemm.workspace_floats(M, 2 * inter, hid), gemm.workspace_floats(M, hid, inter),
                     gemm.workspace_floats(M, qkv_n, hid)), dtype=torch.float32, device=dev)
x, x2, h = torch.empty_like(h0), torch.empty_like(h0), h0.clone()


def separate(w):
    s = gemm.linear_decode_partial_packed(attn, w["po"], hid, ws)
    norm.add_rms_norm_slabs(x, h, ws, s, w["n1"], 1e-5)
    s = gemm.linear_decode_partial_packed(x, w["pgu"], 2 * inter, ws)
    act = activation.silu_and_mul_slabs(ws, s, M, inter, dt)
    s = gemm.linear_decode_partial_packed(act, w["pdn"], hid, ws)
    norm.add_rms_norm_slabs(x2, h, ws, s, w["n2"], 1e-5)
    gemm.linear_decode_partial_packed(x2, w["pqkv"], qkv_n, ws)


cws = torch.empty(gemm.chain_workspace_floats(M, hid, inter, q_size), dtype=torch.float32, device=dev)
qkvp = torch.empty(gemm.workspace_floats(M, qkv_n, hid), dtype=torch.float32, device=dev)
hm, ho, xp, xn = (torch.empty_like(h0) for _ in range(4))
actb = torch.empty((M, inter), dtype=dt, device=dev)
N = 16
sync = torch.zeros((N, gemm.SYNC_WORDS), dtype=torch.int32, device=dev)


def chain(w, i):
    gemm.decode_chain(attn, h0, w["po"], w["pgu"], w["pdn"], w["pqkv"], inter, w["n1"], w["n2"], 1e-5, hm, ho, xp, actb, xn,
                      qkvp, cws, sync[i])


def timeit(body, reps=7):
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


def body_sep():
    for i in range(N):
        separate(W[i % LAYERS])


def body_chain():
    sync.zero_()
    for i in range(N):
        chain(W[i % LAYERS], i)


t_sep = timeit(body_sep)
t_ch = timeit(body_chain)
err = int(sync[:, gemm.SYNC_ERR].abs().sum())
print(f"{model} M={M}: separate {t_sep:.1f} us ({wbytes / t_sep / 1e6:.2f} TB/s)   chain {t_ch:.1f} us "
      f"({wbytes / t_ch / 1e6:.2f} TB/s)   HX_CHAIN_R={os.environ.get('HX_CHAIN_R', 'default')}  err={err}")
