import math, os, sys, torch
sys.path.insert(0, "/root/repo")
from hydrainfer_amd import _lib
from hydrainfer_amd._C.kernel.flash_attn import mha_varlen_fwd
dev, dt = torch.device("cuda:0"), torch.bfloat16
def case(B, n, kv, seed, H=32, D=128, bs=16):
    g = torch.Generator(device=dev).manual_seed(seed)
    rnd = lambda *s: torch.randn(s, generator=g, device=dev).to(dt)
    nb = (kv + bs - 1) // bs
    kc, vc, q = rnd(B * nb, bs, H, D), rnd(B * nb, bs, H, D), rnd(B * n, H, D)
    out = torch.empty_like(q)
    perm = torch.randperm(B * nb, generator=g, device=dev).to(torch.int32)
    cu_b = torch.arange(0, (B + 1) * nb, nb, dtype=torch.int32, device=dev)
    cu_q = torch.arange(0, (B + 1) * n, n, dtype=torch.int32, device=dev)
    cu_k = torch.arange(0, (B + 1) * kv, kv, dtype=torch.int32, device=dev)
    return lambda: mha_varlen_fwd(out, q, kc, vc, cu_q, cu_k, perm, cu_b, None, n, kv, 1 / math.sqrt(D), 0, -1, 0, 0)
def t_eager(fn, reps):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
def t_graph(fn, launches, reps=3):
    side = torch.cuda.Stream(device=dev); side.wait_stream(torch.cuda.current_stream(dev))
    with torch.cuda.stream(side): fn()
    torch.cuda.current_stream(dev).wait_stream(side)
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr):
        for _ in range(launches): fn()
    gr.replay(); torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); gr.replay(); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / launches * 1e3)
    return ts
l = _lib.lib()
for seed in (0, 11):
    for name, args in (("4x704", (4, 704, 704)), ("3x683of704", (3, 683, 704))):
        fn = case(*args, seed)
        for opts in ({"fwd_persistent": 0}, {"fwd_priority": 0, "fwd_pairing": 0}, {"fwd_priority": 1, "fwd_pairing": 0},
                     {"fwd_priority": 0, "fwd_pairing": 1}, {"fwd_priority": 1, "fwd_pairing": 1}):
            for k, val in {"fwd_persistent": 1, "fwd_priority": -1, "fwd_pairing": 1, **opts}.items():
                l.hx_debug_set_option(k.encode(), val)
            print(name, "seed", seed, opts, "graph10", ["%.1f" % x for x in t_graph(fn, 10)], "eager30 %.1f" % t_eager(fn, 30), flush=True)
