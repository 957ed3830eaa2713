#!/usr/bin/env python3
"""In-process A/B of the decode-step executors (hipGraph / launch plan in stream order / chained launch plan): one
model, one runner per executor over the SAME KV pool and block tables, interleaved rounds of K steps at equally
spaced contexts, median ms per step.  Usage: ab_executors.py [7b|13b] [steps] [rounds]"""
import os, statistics, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from hydrainfer_amd.model.llama import LLAVA_1_5_13B, LLAVA_1_5_7B, LlamaForCausalLM
from hydrainfer_amd.model.runner import DecodeRunner, RunnerConfig

dev = torch.device("cuda:0")
shape = LLAVA_1_5_13B if len(sys.argv) > 1 and sys.argv[1] == "13b" else LLAVA_1_5_7B
K = int(sys.argv[2]) if len(sys.argv) > 2 else 40
rounds = int(sys.argv[3]) if len(sys.argv) > 3 else 7
stride = 254 // (K - 1)
first = 705 + (254 - stride * (K - 1)) // 2
model = LlamaForCausalLM.random_init(shape, torch.bfloat16, dev, seed=0)
execs = [e for e in os.environ.get("EXECS", "graph,plan").split(",")]
runners = {}
base = None
names = []
for i, ex in enumerate(execs):
    names.append(f"{ex}#{i}")
from hydrainfer_amd import _lib
for name, ex in zip(names, execs):
    # "plan:decode_hpw4=1": library options (hx_debug_set_option) in force while THIS runner's launches are recorded
    ex, *opts = ex.split(":")
    opts = [(o.split("=")[0].encode(), int(o.split("=")[1])) for o in opts]
    for k, v in opts:
        assert _lib.lib().hx_debug_set_option(k, v) == 0, k
    r = DecodeRunner(model, RunnerConfig(batch=32, prompt_len=704, n_generate=256, use_graph=True, executor=ex,
                                         advance_stride=stride), seed=0)
    if base is None:
        base = r
    else:       # share the first runner's pool: identical bytes and addresses for every executor
        r.pool = base.pool
        r.kv_caches = base.kv_caches
        for ap, bp in zip(r.decode_params.attention_params, base.decode_params.attention_params):
            ap.kv_cache = bp.kv_cache
    r.set_state(first - stride, torch.randint(5, 30000, (32,), device=dev))
    r.capture()
    for k, v in opts:
        _lib.lib().hx_debug_set_option(k, 0)
    runners[name] = r
execs = names
res = {ex: [] for ex in execs}
per_step = os.environ.get("PER_STEP") == "1"
between = os.environ.get("BETWEEN", "")
junk = torch.empty(1 << 30, dtype=torch.uint8, device=dev) if between else None
for rnd in range(rounds):
    for ex in execs:
        r = runners[ex]
        r.set_state(first - stride)
        r.step(record=False)
        if between == "memset":
            for _ in range(20):
                junk.zero_()
        elif between == "sleep":
            torch.cuda.synchronize(); time.sleep(0.2)
        r.set_state(first - stride)
        torch.cuda.synchronize()
        evs = [torch.cuda.Event(enable_timing=True) for _ in range(K + 1)] if per_step else None
        t0 = time.perf_counter()
        for k in range(K):
            if per_step:
                evs[k].record()
            r.step(record=False)
        if per_step:
            evs[K].record()
        torch.cuda.synchronize()
        res[ex].append((time.perf_counter() - t0) / K * 1e3)
        if per_step and rnd == rounds - 1:
            print(ex, "per-step ms:", " ".join(f"{evs[k].elapsed_time(evs[k + 1]):.3f}" for k in range(K)))
for ex in execs:
    v = res[ex]
    print(f"{ex:14s} median {statistics.median(v):.4f} ms/step   min {min(v):.4f}   all {' '.join(f'{x:.3f}' for x in v)}")
