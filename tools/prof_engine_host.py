#!/usr/bin/env python3
"""Where does the host time of an eager (non-graph) engine step go?  One image+text request at a
time on an idle 7B node: wall time of the encode step and the prefill step, and a cProfile of the
prefill step with the GPU work left asynchronous."""
import cProfile, dataclasses, os, pstats, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from hydrainfer_amd.engine.node import LocalCluster
from hydrainfer_amd.engine.request_processor import InstructionCreator
from hydrainfer_amd.engine.scheduler import BatchSchedulerConfig
from hydrainfer_amd.engine.serve import build_node, synthetic_requests
from hydrainfer_amd.model.llama import LlamaForCausalLM
from hydrainfer_amd.model.llava import LlavaLanguageModel

dev, dtype = torch.device("cuda:0"), torch.bfloat16
shape, _ = bench.model_shape(sys.argv[1] if len(sys.argv) > 1 else "7b")
lm = LlavaLanguageModel(LlamaForCausalLM.random_init(shape, dtype, dev, seed=0), image_token_id=32000)
vision, pixels = bench.make_vision(shape, dtype, dev)
node = build_node("EPD0", "EPD", lm, vision, shape, dtype, dev, 61 * 10, 10, 576,
                  BatchSchedulerConfig(max_running_requests=8, token_budgets=2048, image_budgets=8), max_blocks_per_seq=61)
cluster = LocalCluster([node])
creator = InstructionCreator(image_token_id=32000, n_image_tokens_per_image=576, block_size=16)
reqs = synthetic_requests(6, 128, 4, 32000, pixels, seed=3)
prof = cProfile.Profile()
for i, r in enumerate(reqs):
    cluster.add_request(creator.process(r))
    torch.cuda.synchronize(); t0 = time.perf_counter()
    cluster.step()                                   # ImageEmbed
    t_issue = time.perf_counter() - t0
    torch.cuda.synchronize(); t1 = time.perf_counter()
    cluster.step()                                   # EPMigrate (to self) -> skipped
    if i >= 3:
        prof.enable()
    t2 = time.perf_counter()
    cluster.step()                                   # prefill (704 tokens) + sample (syncs on the token)
    t3 = time.perf_counter()
    prof.disable()
    while not cluster.idle():
        cluster.step()
    print(f"request {i}: encode step host {t_issue * 1e3:.2f} ms (synced {(t1 - t0) * 1e3:.2f}), prefill step {(t3 - t2) * 1e3:.2f} ms")
pstats.Stats(prof).sort_stats("tottime").print_stats(14)
