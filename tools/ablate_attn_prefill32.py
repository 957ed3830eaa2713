#!/usr/bin/env python3
"""Where a key tile's cycles go in attn_fwd32_kernel (make EXPERIMENTS=1): the launch with parts of the tile loop
removed (fwd_ablate; results are wrong, only the clock is read).  4 x 704 tokens, H = 32, D = 128, paged, bf16."""
import math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from hydrainfer_amd import _lib
from hydrainfer_amd._C.kernel.flash_attn import mha_varlen_fwd

dev, dt = torch.device("cuda:0"), torch.bfloat16
g = torch.Generator(device=dev).manual_seed(0)
rnd = lambda *s: torch.randn(s, generator=g, device=dev, dtype=torch.float32).to(dt)
B, n, H, D, bs = int(os.environ.get("B", 4)), 704, 32, 128, 16
nb = (n + bs - 1) // bs
kc, vc, q = rnd(B * nb, bs, H, D), rnd(B * nb, bs, H, D), rnd(B * n, H, D)
out = torch.empty_like(q)
perm = torch.randperm(B * nb, generator=g, device=dev).to(torch.int32)
cu_b = torch.arange(0, (B + 1) * nb, nb, dtype=torch.int32, device=dev)
cu = torch.arange(0, (B + 1) * n, n, dtype=torch.int32, device=dev)
fn = lambda: mha_varlen_fwd(out, q, kc, vc, cu, cu, perm, cu_b, None, n, n, 1 / math.sqrt(D), 0, -1, 0, 0)


def timeit(reps=5, n=10):
    """n launches per hipGraph replay (the host takes ~15 us to issue one launch through the Python shim)"""
    s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        fn()
    torch.cuda.current_stream().wait_stream(s)
    g_ = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g_):
        for _ in range(n):
            fn()
    g_.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        g_.replay()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / (reps * n) * 1e3


l = _lib.lib()
names = {0: "full kernel", 1: "no softmax arithmetic", 2: "no P V", 4: "no Q K", 8: "no staging, no barrier", 3: "no softmax, no P V",
         5: "no softmax, no Q K", 6: "no Q K, no P V (softmax + staging)", 7: "staging + barrier only", 9: "no softmax, no staging",
         15: "empty loop", 16: "no barrier (staging kept)", 32: "barrier kept, no tile requests", 48: "neither (= 8)", 256: "the launch alone (workgroups return at once)",
         79: "empty loop, no epilogue stores", 512: "full kernel, Q fragments not loaded (constants)"}
for k, nm in names.items():
    assert l.hx_debug_set_option(b"fwd_persistent", 0) == 0      # the ablations live in the per-item form of the kernel
    assert l.hx_debug_set_option(b"fwd_ablate", k) == 0
    print(f"  {k:2d} {nm:40s} {timeit():7.1f} us", flush=True)
l.hx_debug_set_option(b"fwd_ablate", 0)
l.hx_debug_set_option(b"fwd_persistent", 1)
