#!/usr/bin/env python3
"""Does a read stream by LDS-DMA (global_load_lds_dwordx4, nothing written back to registers) reach more of the HBM peak
than the same stream through registers?  hx_debug_stream_read variant 5 against variant 0 (make EXPERIMENTS=1)."""
import os, sys, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from hydrainfer_amd import _lib
dev = torch.device("cuda:0")
N = 1 << 30
buf = torch.empty(N, dtype=torch.uint8, device=dev); buf.random_(0, 255)
sink = torch.zeros(4, dtype=torch.float32, device=dev)
l = _lib.lib()
def run(variant, unroll, policy, wgs):
    ts = []
    for _ in range(7):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        _lib.check(l.hx_debug_stream_read(buf.data_ptr(), N, variant, 0, unroll, policy, wgs, sink.data_ptr(), _lib.current_stream()), "stream")
        e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    return N / (statistics.median(ts[2:]) * 1e-3) / 1e12
for variant, name in ((0, "registers"), (5, "LDS-DMA")):
    for unroll in (4, 8, 16, 32):
        row = []
        for wgs in (256, 512, 1024, 2048):
            for policy in (0, 1):
                if variant == 5 and 4 * unroll * 1024 * (wgs // 256) > 160 * 1024 and False:
                    row.append("  - "); continue
                row.append(f"{run(variant, unroll, policy, wgs):.2f}")
        print(f"{name:10s} U={unroll:2d}  wgs 256 (plain nt) / 512 / 1024 / 2048 TB/s: " + " ".join(row), flush=True)
