#!/usr/bin/env python3
"""Read-streaming ceiling of this GPU by access shape (hx_debug_stream_read): 1 GiB read once per
launch, median of 5, HIP events.  Columns: variant (rows x bytes per wave instruction), loads in
flight per wave, workgroups (x 4 waves), load policy."""
import os, sys, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from hydrainfer_amd import _lib

dev = torch.device("cuda:0")
N = 1 << 30
buf = torch.empty(N, dtype=torch.uint8, device=dev)
buf.random_(0, 255)
sink = torch.zeros(4, dtype=torch.float32, device=dev)
l = _lib.lib()
names = {0: "contig 1024B", 1: "8 x 128B", 2: "4 x 256B", 3: "2 x 512B", 4: "1 x 1024B"}


def run(variant, pitch, unroll, policy, wgs):
    ts = []
    for _ in range(6):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        _lib.check(l.hx_debug_stream_read(buf.data_ptr(), N, variant, pitch, unroll, policy, wgs, sink.data_ptr(),
                                          _lib.current_stream()), "stream")
        e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    return N / (statistics.median(ts[1:]) * 1e-3) / 1e12


S_OF = {1: 128, 2: 256, 3: 512, 4: 1024}
for variant, pitch in ((0, 0), (1, 16384), (2, 8192), (2, 16384), (3, 8192), (4, 8192), (4, 16384)):
    for unroll in (8, 16, 32):
        if variant and pitch < S_OF[variant] * unroll:
            continue   # a row block would hold less than one chunk: the kernel's index math needs >= 1
        row = []
        for wgs in (512, 1024, 2048):
            for policy in (0, 1):
                row.append(f"{run(variant, pitch, unroll, policy, wgs):.2f}")
        print(f"{names[variant]:13s} pitch {pitch:6d} U={unroll:2d}  wgs 512 (plain nt) / 1024 / 2048 TB/s: " + " ".join(row), flush=True)

# launch-size sweep of the best shape: time = fixed + bytes / rate, inside one hipGraph of 8 launches over
# 8 distinct regions (cold: the regions together exceed the Infinity Cache for sizes >= 48 MiB)
print("size sweep, contiguous, U=8, nt, 1024 wgs, 8 launches per graph over distinct regions:")
for mb in (8, 16, 32, 48, 96, 192, 384):
    n = mb << 20
    regions = [buf[(i * n) % (N - n):][:n] for i in range(8)] if n * 8 <= N else [buf[:n]] * 8
    def body():
        for r in regions:
            _lib.check(l.hx_debug_stream_read(r.data_ptr(), n, 0, 0, 8, 1, 1024, sink.data_ptr(), _lib.current_stream()), "stream")
    st = torch.cuda.Stream(); st.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(st):
        body()
    torch.cuda.current_stream().wait_stream(st)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        body()
    ts = []
    for _ in range(7):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / 8 * 1e3)
    t = statistics.median(ts)
    print(f"  {mb:4d} MiB: {t:7.2f} us per launch = {n / t / 1e6:.2f} TB/s", flush=True)
