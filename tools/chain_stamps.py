#!/usr/bin/env python3
"""Where a chained decode layer's time goes (diagnostic stamps of the launch chain, hx_debug_set_option("chain_stamps")):
for every chained launch of a few middle layers, relative to the END of its predecessor (the moment the predecessor's
last workgroup raised the done flags):
    wg0 at wait   — when this launch's workgroup 0 had issued its prefetch and started to wait (negative = the
                    launch was running while its predecessor was still draining: the overlap a hipGraph cannot have)
    wg0 past wait — when workgroup 0 saw the flag (flag propagation latency)
    end           — when this launch's last workgroup raised its own flags (= its cost on the critical path)
Usage: chain_stamps.py [7b|13b] [ctx]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from hydrainfer_amd import _lib
from hydrainfer_amd.model.llama import LLAVA_1_5_13B, LLAVA_1_5_7B, LlamaForCausalLM
from hydrainfer_amd.model.runner import DecodeRunner, RunnerConfig

dev = torch.device("cuda:0")
shape = LLAVA_1_5_13B if len(sys.argv) > 1 and sys.argv[1] == "13b" else LLAVA_1_5_7B
ctx = int(sys.argv[2]) if len(sys.argv) > 2 else 832
assert _lib.lib().hx_debug_set_option(b"chain_stamps", 3) == 0
model = LlamaForCausalLM.random_init(shape, torch.bfloat16, dev, seed=0)
r = DecodeRunner(model, RunnerConfig(batch=32, prompt_len=704, n_generate=256, use_graph=True, executor="plan"), seed=0)
r.set_state(ctx - 1, torch.randint(5, 30000, (32,), device=dev))
r.capture()
for _ in range(5):
    r.set_state(ctx - 1)
    r.step(record=False)
torch.cuda.synchronize()
plan = r.graph
W = _lib.HX_PLAN_SYNC_BYTES_PER_LAUNCH // 4
sync = plan.sync.cpu().numpy().view(np.uint32)
n = plan.n_any_order + 1
stamp0 = 16 * 32 + 32 + 8 * 32
names = ["attention", "o proj", "norm+gate|up+silu", "down", "norm+qkv"]
rows = []
for i in range(n):
    st = sync[i * W + stamp0: i * W + stamp0 + 12].view(np.uint64)
    rows.append([int(x) for x in st[:4]] + [int(~st[4] & np.uint64(0xFFFFFFFFFFFFFFFF)), int(st[5])])   # + first / last workgroup entry
L = shape.num_hidden_layers
print(f"{'launch':28s} {'first wg in':>12s} {'last wg in':>11s} {'wg0 at wait':>12s} {'wg0 past wait':>14s} {'wg0 done':>10s} {'end':>8s}   (us after the predecessor's end)")
per = {k: [] for k in names}
for i in range(1, n):
    prev_end = rows[i - 1][2]
    a, b, e, s0, f_in, l_in = rows[i]
    name = names[(i - 1) % 5]
    layer = (i - 1) // 5
    vals = [(x - prev_end) / 100.0 for x in (f_in, l_in, a, b, s0, e)]
    per[name].append(vals)
    if L // 2 <= layer < L // 2 + 2:
        print(f"L{layer:02d} {name:24s} " + " ".join(f"{v:12.2f}" for v in vals))
print("\nmean over all layers:")
for k in names:
    v = np.array(per[k])
    print(f"    {k:24s} " + " ".join(f"{v[:, j].mean():12.2f}" for j in range(6)))
print("\nworkgroup end times of one middle layer's launches (count per 2-us bucket after the launch's first workgroup entry):")
for i in range(1 + 5 * (L // 2), 1 + 5 * (L // 2) + 5):
    h = sync[i * W + stamp0 + 16: i * W + stamp0 + 80]
    nz = [(2 * j, int(c)) for j, c in enumerate(h) if c]
    print(f"    {names[(i - 1) % 5]:20s} " + " ".join(f"{t}:{c}" for t, c in nz))
i = 1 + 5 * (L // 2)
hx_ = sync[i * W + stamp0 + 80: i * W + stamp0 + 112].astype(np.float64) / 32
hy_ = sync[i * W + stamp0 + 112: i * W + stamp0 + 144].astype(np.float64) / 32
print("attention, mean end time (us) by head   :", " ".join(f"{v:.0f}" for v in hx_))
print("attention, mean end time (us) by sequence:", " ".join(f"{v:.0f}" for v in hy_))
tot = (rows[n - 1][2] - rows[0][2]) / 100.0
print(f"\nchain from layer 0's qkv end to the last down projection's end: {tot:.1f} us = {tot / L:.2f} us per layer")
