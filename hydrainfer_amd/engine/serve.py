"""Build a serving node (pools + models + scheduler + executors) and replay an arrival trace on it.
Used by bench.py's `serving` leg and tools/bench_engine.py; mirrors what
hydrainfer/cluster/epdnode.py:_update_engine assembles and what benchmark/benchmark.py drives."""
import contextlib
import gc
import os
import time
from typing import List, Optional, Tuple

import numpy as np
import torch

from hydrainfer_amd.engine.executor import BatchFillExecutor, BatchImageEmbedExecutor, InstructionExecutor
from hydrainfer_amd.engine.node import EPDNode, LocalCluster, NodeType
from hydrainfer_amd.engine.rcb import SamplingParameters
from hydrainfer_amd.engine.request_processor import InstructionCreator, TokenRequest
from hydrainfer_amd.engine.scheduler import BatchScheduler, BatchSchedulerConfig, BatchSchedulerContext
from hydrainfer_amd.memory.token_cache_manger import (TokenCacheBlockManager, TokenCacheBlockManagerConfig,
                                                      TokenCacheBlockManagerContext)

_DTYPE_NAMES = {torch.float16: "fp16", torch.bfloat16: "bf16", torch.float32: "fp32"}


def build_node(name: str, node_type: str, language_model, vision_model, lm_shape, dtype: torch.dtype,
               device: torch.device, kv_blocks: int, image_blocks: int, n_image_tokens: int,
               sched: BatchSchedulerConfig, rank: int = 0, graph_decode: bool = True,
               max_blocks_per_seq: int = 256, world_size: int = 1, eager_migrate: bool = True,
               release_prefill_weights: Optional[bool] = None) -> EPDNode:
    """release_prefill_weights (default: this node never prefills, i.e. a "D" node of parallel.epd_roles): keep only
    the packed decode layouts of the decoder weights — one copy in HBM instead of two."""
    nt = NodeType(node_type)
    if nt.has_language_model and nt.enable_decode:
        # every weight layout a decode batch of this node can need is built NOW: before the cache pools are
        # allocated (it counts against the same HBM) and never in the middle of serving
        if release_prefill_weights is None:
            release_prefill_weights = not nt.enable_prefill
        language_model.language_model.prepare_decode(max_rows=(sched.max_running_requests + 3) // 4 * 4,
                                                     keep_row_major=not release_prefill_weights)
    # one node of MI355Xs: every rank is a same-host peer, so pulls take the IPC path
    ctx = TokenCacheBlockManagerContext(rank=rank, rank2host={r: "localhost" for r in range(max(world_size, rank + 1))})
    kv = img = None
    if nt.has_kv_cache:
        kv = TokenCacheBlockManager(TokenCacheBlockManagerConfig(
            n_layers=lm_shape.num_hidden_layers, n_tokens=2, n_blocks=kv_blocks, block_size=16,
            n_heads=lm_shape.num_key_value_heads, head_size=lm_shape.head_dim, dtype=_DTYPE_NAMES[dtype],
            device=str(device)), ctx)
    if nt.has_image_cache:
        img = TokenCacheBlockManager(TokenCacheBlockManagerConfig(
            n_layers=1, n_tokens=1, n_blocks=image_blocks, block_size=n_image_tokens,
            n_heads=lm_shape.num_attention_heads, head_size=lm_shape.head_dim, dtype=_DTYPE_NAMES[dtype],
            device=str(device)), ctx)
    fill = None
    if nt.has_language_model:
        decoder = None
        if graph_decode and nt.enable_decode:
            from hydrainfer_amd.engine.graph_decode import GraphedDecoder
            decoder = GraphedDecoder(language_model, kv, max_batch=sched.max_running_requests,
                                     max_blocks_per_seq=max_blocks_per_seq)
        fill = BatchFillExecutor(language_model, kv, img, dtype, device, graph_decoder=decoder)
    emb = BatchImageEmbedExecutor(vision_model, img, lm_shape.num_attention_heads, lm_shape.head_dim, dtype,
                                  device, use_graphs=graph_decode) if nt.has_vision_model else None
    scheduler = BatchScheduler(sched, BatchSchedulerContext(kv, img))
    return EPDNode(name, nt, scheduler, InstructionExecutor(fill, emb), kv, img, eager_migrate=eager_migrate)


def warm_library_gemms(language_model, token_budget: int, max_decode_rows: int = 64, vision_model=None,
                       pixel_values: Optional[torch.Tensor] = None, image_budget: int = 8) -> None:
    """The prefill-side linears are library GEMMs, and the library picks (and lazily loads) a
    different kernel for different row counts: the first step with a new batch shape can stall for
    100-400 ms while a code object is loaded — seen as TTFT outliers in short serving runs.  Run
    every projection shape once over the row counts a chunked-prefill step can have."""
    model = language_model.language_model
    st, dev, dt = model.state, model.device, model.dtype
    rows = sorted(set(list(range(64, token_budget + 1, 64)) + [token_budget + r for r in (1, 8, max_decode_rows)]))
    for name in ("l0.wqkv", "l0.wo", "l0.wgu", "l0.wdown"):
        w = st[name]
        if w.device.type == "meta":      # decode-only node: no prefill GEMMs to warm
            continue
        x = torch.zeros((rows[-1], w.shape[1]), dtype=dt, device=dev)
        for m in rows:
            torch.matmul(x[:m], w.t())
    x = torch.zeros((max_decode_rows, st["lm_head"].shape[1]), dtype=dt, device=dev)
    for m in (1, 2, 4, 8, 16, 32, max_decode_rows):      # (NOT every row count: warming 64 shapes was followed by 70-400 ms
        torch.matmul(x[:m], st["lm_head"].t())           #  stalls in the first prefill chunks — the library re-loading kernels)
    if vision_model is not None and pixel_values is not None:      # the tower for 1 .. image_budget images
        px = pixel_values.to(device=dev, dtype=dt)
        for n in range(1, image_budget + 1):
            vision_model.forward(px.expand(n, -1, -1, -1))
    torch.cuda.synchronize(dev)


def tune_library_gemms(language_model, rows=(704,), vision_model=None, pixel_values: Optional[torch.Tensor] = None,
                       image_counts=(1,), rotating_buffer_mb: int = 512) -> dict:
    """Start-up autotuning of the LIBRARY GEMMs this deployment will run most often: the four prefill projections at
    `rows` tokens (the prompt lengths / chunk sizes to expect) and the vision tower for `image_counts` images per step.
    torch's TunableOp times the library's own candidate kernels for each of those shapes on THIS GPU (over rotating
    buffers larger than the Infinity Cache: the weights of a 7B layer are never cache-hot in the pipeline) and keeps the
    fastest; tuning is switched OFF again before this returns, so no later shape — a chunk of some other length — ever
    pauses to tune: it takes the library's default kernel as before.  ~1 s per shape at start-up; a 704-token prefill
    went 11.4 -> 10.6-10.8 ms (tools/bench_ttft_tunable.py), single-request TTFT 14.4 -> 13.4 ms.  The results file is kept
    out of the working directory.  CAUTION (round 6): the pass RUNS every candidate kernel of the library for each shape.
    The four 7B projections at 704 and at 2048 rows tune cleanly (a 2048-token chunk: 29.3 -> 24.9 ms); the vision
    tower for 8 images does NOT — one candidate faults (`Memory access fault by GPU`, tools/probes/tune_probe.py), which
    is why bench.py never passes `vision_model`.  Widen the set one shape at a time, under a timeout.
    Returns {"shapes": n, "seconds": t}."""
    import tempfile
    import torch.cuda.tunable as tunable
    model = getattr(language_model, "language_model", language_model)
    st, dev, dt = model.state, model.device, model.dtype
    t0 = time.perf_counter()
    tunable.set_filename(os.path.join(tempfile.gettempdir(), f"hx_tunableop_{os.getpid()}.csv"))
    if rotating_buffer_mb is not None:
        tunable.set_rotating_buffer_size(int(rotating_buffer_mb))
    tunable.enable(True)
    tunable.tuning_enable(True)
    n = 0
    try:
        for name in ("l0.wqkv", "l0.wo", "l0.wgu", "l0.wdown"):
            w = st[name]
            if w.device.type == "meta":          # decode-only node: no prefill GEMMs
                continue
            for m in rows:
                torch.matmul(torch.zeros((int(m), w.shape[1]), dtype=dt, device=dev), w.t())
                n += 1
        if vision_model is not None and pixel_values is not None:
            px = pixel_values.to(device=dev, dtype=dt)
            for k in image_counts:
                vision_model.forward(px.expand(int(k), -1, -1, -1))
                n += 6
        torch.cuda.synchronize(dev)
    finally:
        tunable.tuning_enable(False)             # tuned shapes keep their kernels; nothing else is ever tuned
    return {"shapes": n, "seconds": round(time.perf_counter() - t0, 1)}


def synthetic_requests(n: int, n_text: int, max_tokens: int, image_token_id: int, pixels: Optional[torch.Tensor],
                       vocab_text: Tuple[int, int] = (1000, 31999), seed: int = 0) -> List[TokenRequest]:
    """The benchmark request of SURVEY.md §8(d): one image + n_text random text ids, distinct per request."""
    out = []
    for i in range(n):
        g = torch.Generator().manual_seed(seed * 100003 + i)
        text = torch.randint(vocab_text[0], vocab_text[1], (n_text,), generator=g).tolist()
        ids = ([image_token_id] if pixels is not None else []) + text
        out.append(TokenRequest(request_id=i, token_ids=ids, pixel_values=pixels, image_size=(336, 336),
                                image_hash=(seed << 20) + i,       # distinct images: no prefix sharing
                                sampling_params=SamplingParameters(max_tokens=max_tokens)))
    return out


@contextlib.contextmanager
def quiet_gc():
    """A full (generation 2) collection walks every live request's instruction chain and lists and
    was seen to stall a prefill step for 37 ms on a 7B node (tools/prof_engine_host.py).  Serving
    loops run with the cyclic collector off — nothing the engine allocates per step is cyclic —
    after moving everything that already exists out of the collector's sight."""
    was_enabled = gc.isenabled()
    gc.collect()
    gc.freeze()
    gc.disable()
    try:
        yield
    finally:
        if was_enabled:
            gc.enable()
        gc.unfreeze()


def poisson_arrivals(n: int, rate: float, seed: int = 0) -> List[float]:
    """benchmark/timestamp.py:9-16 with the seeding of benchmark/benchmark.py:136-137."""
    rng = np.random.RandomState(seed)
    gaps = rng.exponential(1.0 / rate, n)
    return np.cumsum(gaps).tolist()


# Turning a request into its instruction chain costs 1-3 ms of host time (prefix hashes, one
# instruction object per generated token).  The reference does it on a pool of worker threads
# (request_processor.py:214-236); here at most this many arrivals are taken in between two engine
# steps, so a burst does not hold the first encode back until the whole burst has been processed —
# the GPU works on the first ones while the host prepares the next.
ADMIT_PER_STEP = 8


def replay(cluster: LocalCluster, creator: InstructionCreator, requests: List[TokenRequest],
           arrivals: List[float], device: torch.device) -> dict:
    """Wall-clock replay: a request enters at its arrival time (0 = all at once).  Returns the
    serving metrics of benchmark/metric.py: output tokens/s, TTFT and TPOT percentiles."""
    order = sorted(range(len(requests)), key=lambda i: arrivals[i])
    rcbs = [None] * len(requests)
    torch.cuda.synchronize(device)
    with quiet_gc():
        t0 = time.perf_counter()
        nxt, steps = 0, 0
        while nxt < len(order) or not cluster.idle():
            now = time.perf_counter() - t0
            admitted = 0
            while nxt < len(order) and arrivals[order[nxt]] <= now and admitted < ADMIT_PER_STEP:
                admitted += 1
                i = order[nxt]
                rcbs[i] = creator.process(requests[i])
                cluster.add_request(rcbs[i])
                rcbs[i].metric.arrival_time = t0 + arrivals[i]
                nxt += 1
            if cluster.step() == 0 and nxt < len(order):
                time.sleep(max(0.0, min(0.001, arrivals[order[nxt]] - (time.perf_counter() - t0))))
            steps += 1
        torch.cuda.synchronize(device)
        wall = time.perf_counter() - t0
    ttft = sorted(r.metric.token_times[0] - r.metric.arrival_time for r in rcbs)
    tpot = sorted((r.metric.token_times[-1] - r.metric.token_times[0]) / max(1, len(r.metric.token_times) - 1)
                  for r in rcbs)
    n_out = sum(len(r.output_token_ids) for r in rcbs)
    pct = lambda xs, p: xs[min(len(xs) - 1, int(p * len(xs)))]
    return {"requests": len(rcbs), "output_tokens": n_out, "wall_s": round(wall, 3),
            "output_tok_s": round(n_out / wall, 1), "steps": steps,
            "ttft_mean_ms": round(sum(ttft) / len(ttft) * 1e3, 2),
            "ttft_p50_ms": round(pct(ttft, 0.5) * 1e3, 2), "ttft_p99_ms": round(pct(ttft, 0.99) * 1e3, 2),
            "tpot_p50_ms": round(pct(tpot, 0.5) * 1e3, 3), "tpot_p99_ms": round(pct(tpot, 0.99) * 1e3, 3)}
