#!/usr/bin/env python3
"""Where do the first 300 ms of the 32-request burst go (bench.py's `serving` leg: TTFT p50 265-310 ms)?  The leg's node,
the same 32 requests at t = 0, every engine step timed: what it ran (images encoded, prefill tokens, decode rows), its
host time (step() returns) and — with SYNC=1 — its time including the GPU work.  SYNC=0 leaves the pipeline as it is
and prints only the host side and the first-token stamps."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from hydrainfer_amd.engine.node import LocalCluster
from hydrainfer_amd.engine.request_processor import InstructionCreator
from hydrainfer_amd.engine.scheduler import BatchSchedulerConfig
from hydrainfer_amd.engine.serve import ADMIT_PER_STEP, build_node, quiet_gc, synthetic_requests, warm_library_gemms
from hydrainfer_amd.model.llama import LlamaForCausalLM
from hydrainfer_amd.model.llava import LlavaLanguageModel

SYNC = os.environ.get("SYNC", "1") == "1"
PROFILE = os.environ.get("PROFILE") == "1"
B = int(os.environ.get("B", "32"))
dev, dtype = torch.device("cuda:0"), torch.bfloat16
shape, _ = bench.model_shape("7b")
model = LlamaForCausalLM.random_init(shape, dtype, dev, seed=0)
lm = LlavaLanguageModel(model, image_token_id=32000)
vision, pixels = bench.make_vision(shape, dtype, dev)
per_req = (576 + 128 + 256 + 15) // 16 + 1
sched = BatchSchedulerConfig(priority="prefill", max_running_requests=B, chunked_prefill=True, token_budgets=2048, image_budgets=8)
node = build_node("EPD0", "EPD", lm, vision, shape, dtype, dev, per_req * (B + 2), B + 2, 576, sched, max_blocks_per_seq=per_req)
node.executor.fill_executor.graph_decoder.warmup(list(range(4, B + 1, 4)), kv_max=1024)
node.executor.image_embed_executor.warmup(pixels, sched.image_budgets)
warm_library_gemms(lm, sched.token_budgets, B, vision, pixels, sched.image_budgets)
if os.environ.get("TUNE"):      # the four prefill projections autotuned for the chunk budget (serve.tune_library_gemms)
    from hydrainfer_amd.engine.serve import tune_library_gemms
    print("tuned:", tune_library_gemms(model, rows=(2048,)), flush=True)
if os.environ.get("VISION_EAGER"):      # A/B: the vision tower launched kernel by kernel instead of replayed from its hipGraph
    node.executor.image_embed_executor.use_graphs = False
cluster = LocalCluster([node])
creator = InstructionCreator(image_token_id=32000, n_image_tokens_per_image=576, block_size=16,
                             max_position_embeddings=shape.max_position_embeddings)
ex = node.executor
log = []
_fill, _emb = ex.execute_fill, ex.execute_image_embed


def fill(batch):
    n_tok = sum(len(inst.token_ids) for _, inst in batch)
    n_dec = sum(1 for _, inst in batch if len(inst.token_ids) == 1)
    log.append(("fill", len(batch), n_tok, n_dec))
    _fill(batch)


def emb(batch):
    log.append(("embed", len(batch), 0, 0))
    _emb(batch)


ex.execute_fill, ex.execute_image_embed = fill, emb
for rep in range(2):
    reqs = synthetic_requests(B, 128, 256 if rep else 4, 32000, pixels, seed=rep + 1)
    rcbs, rows = [], []
    torch.cuda.synchronize()
    with quiet_gc():
        t0 = time.perf_counter()
        nxt = 0
        while nxt < B or not cluster.idle():
            a0 = time.perf_counter()
            admitted = 0
            while nxt < B and admitted < ADMIT_PER_STEP:
                rcb = creator.process(reqs[nxt]); cluster.add_request(rcb); rcb.metric.arrival_time = t0
                rcbs.append(rcb); nxt += 1; admitted += 1
            a1 = time.perf_counter()
            del log[:]
            prof = None
            if PROFILE and rep and len(rows) < 8:
                import cProfile
                prof = cProfile.Profile(); prof.enable()
            cluster.step()
            a2 = time.perf_counter()
            if prof is not None:
                prof.disable()
                if a2 - a1 > 0.045:      # a stalled step: where was the host?
                    import io, pstats
                    buf = io.StringIO()
                    pstats.Stats(prof, stream=buf).sort_stats("tottime").print_stats(8)
                    print(f"--- step at {1e3 * (a0 - t0):.1f} ms took {1e3 * (a2 - a1):.1f} ms:", "\n".join(buf.getvalue().splitlines()[6:20]), flush=True)
            if SYNC:
                torch.cuda.synchronize()
            a3 = time.perf_counter()
            if rep and a3 - t0 < 0.45:
                rows.append((a0 - t0, a1 - a0, a2 - a1, a3 - a2, list(log)))
    if rep:
        print(f"SYNC={int(SYNC)}  t ms | admit ms | step host ms | gpu tail ms | what")
        for t, adm, host, tail, what in rows:
            print(f"{t * 1e3:7.1f} | {adm * 1e3:5.1f} | {host * 1e3:6.1f} | {tail * 1e3:6.1f} | {what}")
        ttft = sorted((r.metric.token_times[0] - t0) * 1e3 for r in rcbs)
        print("TTFT ms:", [round(x) for x in ttft])
