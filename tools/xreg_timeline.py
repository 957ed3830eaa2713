#!/usr/bin/env python3
"""Phase timeline of the norm-fused GEMM launches inside the real decode step (hx_debug_set_option("xreg_timeline", 1)):
for the first and the last workgroup of the grid, microseconds after the launch's first workgroup entered: rows produced
/ prefetch issued, x flag seen, end of row group 1, 2, ..., end of the MFMA loop, end of the kernel."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from hydrainfer_amd import _lib
from hydrainfer_amd.model.llama import LLAVA_1_5_13B, LLAVA_1_5_7B, LlamaForCausalLM
from hydrainfer_amd.model.runner import DecodeRunner, RunnerConfig
dev = torch.device("cuda:0")
shape = LLAVA_1_5_13B if len(sys.argv) > 1 and sys.argv[1] == "13b" else LLAVA_1_5_7B
B = int(sys.argv[2]) if len(sys.argv) > 2 else 32          # 33 .. 64: the wide kernel's launches (stamps per unit of two row groups)
assert _lib.lib().hx_debug_set_option(b"xreg_timeline", 1) == 0
model = LlamaForCausalLM.random_init(shape, torch.bfloat16, dev, seed=0)
r = DecodeRunner(model, RunnerConfig(batch=B, prompt_len=704, n_generate=256, use_graph=True, executor="plan"), seed=0)
r.set_state(831, torch.randint(5, 30000, (B,), device=dev))
r.capture()
for _ in range(5):
    r.set_state(831); r.step(record=False)
torch.cuda.synchronize()
sync = model.xreg_sync.cpu().numpy().view(np.uint32)          # [L, 2, 512]
L = shape.num_hidden_layers
names = {0: "norm + gate|up + silu", 1: "norm + qkv (next layer)"}
labels = ["enter", "rows done / prefetch", "x flag seen", "rg1", "rg2", "rg3", "rg4", "rg5", "rg6", "-", "mfma loop done", "kernel end"]
for which in (0, 1):
    acc = []
    for l in range(2, L - 1):
        a = sync[l, which, 384:384 + 24].view(np.uint64).astype(np.int64)
        b = sync[l, which, 416:416 + 24].view(np.uint64).astype(np.int64)
        t0 = min(a[0], b[0])
        acc.append(([(x - t0) / 100 if x else float("nan") for x in a], [(x - t0) / 100 if x else float("nan") for x in b]))
    first = np.nanmean(np.array([x[0] for x in acc]), axis=0)
    last = np.nanmean(np.array([x[1] for x in acc]), axis=0)
    print(f"\n{names[which]}  (mean over layers, us after the earlier of the two workgroups entered)")
    for k, lab in enumerate(labels):
        if lab != "-" and not (np.isnan(first[k]) and np.isnan(last[k])):
            print(f"    {lab:22s} first wg {first[k]:7.2f}   last wg {last[k]:7.2f}")
