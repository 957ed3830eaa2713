"""Serving-level measurement: the continuous-batching engine on one MI355X replaying a request
trace (all-at-once or Poisson) of LLaVA-1.5 image+text requests.
    python tools/bench_engine.py --model 7b --requests 32 --rate 0 --max-tokens 256"""
import argparse
import dataclasses
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", default="7b", choices=["7b", "13b", "tiny"])
    ap.add_argument("--requests", type=int, default=32)
    ap.add_argument("--rate", type=float, default=0.0, help="Poisson req/s; 0 = all at t=0")
    ap.add_argument("--n-text", type=int, default=128)
    ap.add_argument("--max-tokens", type=int, default=256)
    ap.add_argument("--max-running", type=int, default=32)
    ap.add_argument("--token-budget", type=int, default=2048)
    ap.add_argument("--image-budget", type=int, default=8)
    ap.add_argument("--topology", default="EPD", help="e.g. EPD or E,P,D (all on cuda:0)")
    ap.add_argument("--dtype", default="bf16")
    ap.add_argument("--no-chunk", action="store_true")
    ap.add_argument("--priority", default="prefill")
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--no-image", action="store_true", help="text-only requests")
    ap.add_argument("--repeat", type=int, default=1, help="replay the trace this many times in one process")
    args = ap.parse_args()

    from hydrainfer_amd.engine.request_processor import InstructionCreator
    from hydrainfer_amd.engine.scheduler import BatchSchedulerConfig
    from hydrainfer_amd.engine.node import LocalCluster
    from hydrainfer_amd.engine.serve import (build_node, poisson_arrivals, replay, synthetic_requests,
                                             warm_library_gemms)
    from hydrainfer_amd.model.clip import CLIP_VIT_L_14_336, ClipShape, LlavaVisionModel
    from hydrainfer_amd.model.llama import LLAVA_1_5_13B, LLAVA_1_5_7B, LlamaForCausalLM, LlamaShape
    from hydrainfer_amd.model.llava import LlavaLanguageModel

    dev = torch.device("cuda:0")
    dtype = {"bf16": torch.bfloat16, "fp16": torch.float16}[args.dtype]
    if args.model == "tiny":
        shape = LlamaShape(256, 512, 2, 2, 2, 128, 32064)
        cshape = ClipShape(hidden_size=128, intermediate_size=256, num_hidden_layers=3, num_attention_heads=2,
                           image_size=336, patch_size=14, projector_hidden_size=256)
    else:
        shape = LLAVA_1_5_7B if args.model == "7b" else LLAVA_1_5_13B
        cshape = dataclasses.replace(CLIP_VIT_L_14_336, projector_hidden_size=shape.hidden_size)
    lm = LlavaLanguageModel(LlamaForCausalLM.random_init(shape, dtype, dev, seed=0), image_token_id=32000)
    vision = LlavaVisionModel.random_init(cshape, dtype, dev, seed=1)
    rng = np.random.RandomState(0)
    img = rng.randint(0, 256, (336, 336, 3)).astype(np.float32) / 255.0
    mean = np.array([0.48145466, 0.4578275, 0.40821073], dtype=np.float32)
    std = np.array([0.26862954, 0.26130258, 0.27577711], dtype=np.float32)
    pixels = torch.from_numpy((img - mean) / std).permute(2, 0, 1)[None]

    per_req_blocks = (576 + args.n_text + args.max_tokens + 15) // 16 + 1
    kv_blocks = per_req_blocks * (2 * args.max_running + 2)
    sched = BatchSchedulerConfig(priority=args.priority, max_running_requests=args.max_running,
                                 chunked_prefill=not args.no_chunk, token_budgets=args.token_budget,
                                 image_budgets=args.image_budget)
    nodes = [build_node(f"{t}{k}", t, lm, vision, shape, dtype, dev, kv_blocks, 2 * args.max_running + 2, 576,
                        dataclasses.replace(sched), graph_decode=not args.no_graph,
                        max_blocks_per_seq=per_req_blocks) for k, t in enumerate(args.topology.split(","))]
    for node in nodes:      # capture the decode graphs outside the timed replay
        fe = node.executor.fill_executor
        if fe is not None and fe.graph_decoder is not None:
            fe.graph_decoder.warmup(list(range(4, args.max_running + 1, 4)), kv_max=1024)
        ie = node.executor.image_embed_executor
        if ie is not None and not args.no_graph:
            ie.warmup(pixels, args.image_budget)
    cluster = LocalCluster(nodes)
    creator = InstructionCreator(image_token_id=32000, n_image_tokens_per_image=576, block_size=16)

    warm_library_gemms(lm, args.token_budget, args.max_running, vision, pixels, args.image_budget)
    # warm-up: two short requests (kernel loading, allocator, workspace growth)
    warm = synthetic_requests(2, args.n_text, 4, 32000, pixels, seed=99)
    replay(cluster, creator, warm, [0.0, 0.0], dev)
    reqs = synthetic_requests(args.requests, args.n_text, args.max_tokens, 32000, None if args.no_image else pixels, seed=1)
    arrivals = poisson_arrivals(args.requests, args.rate, 0) if args.rate > 0 else [0.0] * args.requests
    for rep in range(args.repeat - 1):
        r0 = replay(cluster, creator, synthetic_requests(args.requests, args.n_text, args.max_tokens, 32000, pixels,
                                                         seed=10 + rep), arrivals, dev)
        print(json.dumps({k: r0[k] for k in ("wall_s", "output_tok_s", "ttft_p50_ms", "tpot_p50_ms")}), file=sys.stderr)
    res = replay(cluster, creator, reqs, arrivals, dev)
    res.update(model=args.model, topology=args.topology, rate=args.rate, max_running=args.max_running,
               token_budget=args.token_budget, chunked=not args.no_chunk, graph_decode=not args.no_graph)
    print(json.dumps(res))


if __name__ == "__main__":
    main()
