#!/usr/bin/env python3
"""Does the PLACEMENT of the weights matter?  Separate processes of the same binary differ by up to 1.5 % per step while
rounds inside one process agree to 0.1 %, and the kernel traces put the difference in the weight-streaming launches, not
in attention.  Here: one process, several models with identical contents allocated one after the other (ARENA=1: each
model's packed decode weights live in ONE allocation), one runner each over the SAME KV pool, interleaved rounds.
    ab_models.py [n_models] [steps] [rounds]"""
import os, statistics, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from hydrainfer_amd.model.llama import LLAVA_1_5_7B, LlamaForCausalLM
from hydrainfer_amd.model.runner import DecodeRunner, RunnerConfig

dev = torch.device("cuda:0")
n_models = int(sys.argv[1]) if len(sys.argv) > 1 else 3
K = int(sys.argv[2]) if len(sys.argv) > 2 else 40
rounds = int(sys.argv[3]) if len(sys.argv) > 3 else 5
stride = 254 // (K - 1)
first = 705 + (254 - stride * (K - 1)) // 2
arena = os.environ.get("ARENA", "0") == "1"
runners, base = [], None
for i in range(n_models):
    if i and os.environ.get("JUNK", "1") == "1":       # move the next model's allocations
        junk = torch.empty((i * 777) << 20, dtype=torch.uint8, device=dev)
    model = LlamaForCausalLM.random_init(LLAVA_1_5_7B, torch.bfloat16, dev, seed=0)
    if arena and i == n_models - 1:
        model.prepare_decode(max_rows=32)
        total = sum(t.numel() for t in list(model.packed_x.values()) + list(model.packed.values()))
        buf = torch.empty(total + (1 << 20), dtype=torch.bfloat16, device=dev)
        off = 0
        for d in (model.packed_x, model.packed):
            for k in sorted(d, key=lambda k: (int(k.split(".")[0][1:]), k)):
                t = d[k]
                v = buf[off:off + t.numel()].view(t.shape)
                v.copy_(t)
                d[k] = v
                dw = model.dw.get(k) if d is model.packed_x else model.dw_lds.get(k)
                if dw is not None:
                    dw.packed = v
                    dw.desc.packed = v.data_ptr()
                off += (t.numel() + 511) // 512 * 512
    r = DecodeRunner(model, RunnerConfig(batch=32, prompt_len=704, n_generate=256, use_graph=True, executor="plan",
                                         advance_stride=stride), seed=0)
    if base is None:
        base = r
    else:
        r.pool = base.pool
        r.kv_caches = base.kv_caches
        for ap, bp in zip(r.decode_params.attention_params, base.decode_params.attention_params):
            ap.kv_cache = bp.kv_cache
    r.set_state(first - stride, torch.randint(5, 30000, (32,), device=dev))
    r.capture()
    runners.append(r)
res = [[] for _ in runners]
for rnd in range(rounds):
    for i, r in enumerate(runners):
        r.set_state(first - stride)
        r.step(record=False)
        r.set_state(first - stride)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for k in range(K):
            r.step(record=False)
        torch.cuda.synchronize()
        res[i].append((time.perf_counter() - t0) / K * 1e3)
for i, v in enumerate(res):
    ptr = runners[i].model.packed_x["l5.wgu"].data_ptr()
    print(f"model {i}{' (arena)' if arena and i == n_models - 1 else ''}: median {statistics.median(v):.4f} ms/step  all {' '.join(f'{x:.3f}' for x in v)}  l5.wgu @ {ptr:#x} (mod 2M {ptr % (2 << 20):#x})")
