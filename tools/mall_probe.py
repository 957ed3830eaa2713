import os, sys, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from hydrainfer_amd._C.kernel import gemm
dev, dt = torch.device("cuda:0"), torch.bfloat16
def graph_time(fn, n_inner, reps=7):
    s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s): fn()
    torch.cuda.current_stream().wait_stream(s)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g): fn()
    ts = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / n_inner * 1e3)
    return statistics.median(ts)
for name, (N, K) in {"gate_up": (22016, 4096), "qkv": (12288, 4096), "down": (4096, 11008)}.items():
    for M in (16, 32):
        nc = 6
        ws = [gemm.pack_weight((torch.randn((N, K), device=dev) * 0.02).to(dt)) for _ in range(nc)]
        x = torch.randn((M, K), device=dev).to(dt)
        a = torch.empty(gemm.workspace_floats(M, N, K), dtype=torch.float32, device=dev)
        cold = graph_time(lambda: [gemm.linear_decode_partial_packed(x, ws[i % nc], N, a) for i in range(12)], 12)
        # every weight set read twice in a row: launches 2i and 2i+1 share a set
        twice = graph_time(lambda: [gemm.linear_decode_partial_packed(x, ws[(i // 2) % nc], N, a) for i in range(12)], 12)
        print(f"{name} M={M}: cold {cold:.1f} us/launch; pairs (cold+repeat) {twice:.1f} us/launch avg -> repeat launch ~{2*twice-cold:.1f} us", flush=True)
        del ws
