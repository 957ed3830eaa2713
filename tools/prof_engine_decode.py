#!/usr/bin/env python3
"""Steady-state decode steps of the engine (32 running requests, hipGraph decode with one step of
look-ahead) on a 7B node: wall time per step against the bare decode graph, and a cProfile of the
host side — where does serving TPOT exceed the graph's step time?"""
import cProfile, os, pstats, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from hydrainfer_amd.engine.node import LocalCluster
from hydrainfer_amd.engine.request_processor import InstructionCreator
from hydrainfer_amd.engine.scheduler import BatchSchedulerConfig
from hydrainfer_amd.engine.serve import build_node, synthetic_requests, warm_library_gemms, quiet_gc
from hydrainfer_amd.model.llama import LlamaForCausalLM
from hydrainfer_amd.model.llava import LlavaLanguageModel

dev, dtype = torch.device("cuda:0"), torch.bfloat16
shape, _ = bench.model_shape(sys.argv[1] if len(sys.argv) > 1 else "7b")
batch = 32
lm = LlavaLanguageModel(LlamaForCausalLM.random_init(shape, dtype, dev, seed=0), image_token_id=32000)
per_req = (704 + 256 + 15) // 16 + 1
sched = BatchSchedulerConfig(priority="prefill", max_running_requests=batch, chunked_prefill=True,
                             token_budgets=2048, image_budgets=8)
node = build_node("EPD0", "EPD", lm, None, shape, dtype, dev, per_req * (batch + 2), batch + 2, 576, sched,
                  max_blocks_per_seq=per_req)
node.executor.fill_executor.graph_decoder.warmup([batch], kv_max=1024)
warm_library_gemms(lm, sched.token_budgets, batch)
cluster = LocalCluster([node])
creator = InstructionCreator(image_token_id=32000, n_image_tokens_per_image=576, block_size=16)
reqs = synthetic_requests(batch, 704, 256, 32000, None, seed=3)     # text-only prompts of 704 tokens
for r in reqs:
    cluster.add_request(creator.process(r))
with quiet_gc():
    for _ in range(40):                       # prefill chunks + first decode steps
        cluster.step()
    torch.cuda.synchronize()
    n = 120
    t0 = time.perf_counter()
    for _ in range(n):
        cluster.step()
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) / n * 1e3
    prof = cProfile.Profile()
    host = []
    prof.enable()
    for _ in range(40):
        t1 = time.perf_counter()
        cluster.step()
        host.append(time.perf_counter() - t1)
    prof.disable()
    torch.cuda.synchronize()
print(f"engine decode step (wall, 120 steps): {wall:.3f} ms;  host time inside step(): median {sorted(host)[len(host)//2]*1e3:.3f} ms")
pstats.Stats(prof).sort_stats("tottime").print_stats(18)
