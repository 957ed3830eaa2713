"""Engine end to end on a tiny LLaVA: continuous batching with chunked prefill, prefix-cache hits,
image encode -> image cache -> prefill -> decode, and E/P/D disaggregation with block migration.

The same scripted arrival trace runs (a) on CPU with the oracle as the model and CPU tensors as
pools, (b) on the GPU through libhydra_hip.  Every sampled row's logits are compared; greedy tokens
must agree wherever the oracle's top-2 gap exceeds twice the tolerance (a request whose token
legitimately flips on a near-tie stops being compared from there on)."""
import numpy as np
import pytest
import torch

from hydrainfer_amd.engine import BatchSchedulerConfig, InstructionCreator, SamplingParameters, TokenRequest
from hydrainfer_amd.engine.node import LocalCluster
from tests.engine_util import (CpuPoolManager, LogitsTap, OracleLM, OracleVision, make_node, run_trace)
from tests.golden import cases as C

N_IMG_TOK = (C.TINY_CLIP["image_size"] // C.TINY_CLIP["patch_size"]) ** 2     # 16
BS = C.TINY_BLOCK_SIZE


def trace_requests():
    g = torch.Generator().manual_seed(4242)
    pixels = C.tiny_clip_pixels(2)
    text_len = [10, 45, 3, 28, 17, 33, 10, 8, 21]
    arrive = [0, 0, 0, 2, 5, 5, 9, 9, 12]
    out = []
    first = None
    for i, (n, a) in enumerate(zip(text_len, arrive)):
        text = torch.randint(0, C.TINY_IMAGE_TOKEN_ID, (n,), generator=g).tolist()
        has_image = i % 4 != 3
        ids = ([C.TINY_IMAGE_TOKEN_ID] if has_image else []) + text
        img = i % 2
        if i == 0:
            first = ids
        if i == 6:                      # repeats request 0: same image, same text -> prefix hit
            ids, img = list(first), 0
        out.append((a, TokenRequest(request_id=i, token_ids=ids,
                                    pixel_values=pixels[img:img + 1].clone() if has_image else None,
                                    image_size=(56, 56), image_hash=9000 + img,
                                    sampling_params=SamplingParameters(max_tokens=3 + (i * 2) % 6))))
    return out


def sched_cfg(chunked):
    return BatchSchedulerConfig(priority="prefill", max_running_requests=6, chunked_prefill=chunked,
                                token_budgets=40, image_budgets=2)


def creator():
    return InstructionCreator(image_token_id=C.TINY_IMAGE_TOKEN_ID, n_image_tokens_per_image=N_IMG_TOK,
                              block_size=BS, ignore_eos=True)


def shapes():
    from hydrainfer_amd.model.clip import ClipShape
    from hydrainfer_amd.model.llama import LlamaShape
    return LlamaShape(**C.TINY_LLAMA), ClipShape(**C.TINY_CLIP)


def clip_state(dt):
    from hydrainfer_amd.model.clip import random_state_dict
    return {k: v.to(dt) for k, v in random_state_dict(shapes()[1], seed=3, std=0.05).items()}


def oracle_cluster(dt, topology, chunked):
    """topology: list of node type strings, e.g. ['EPD'] or ['E', 'P', 'D']."""
    lshape, cshape = shapes()
    lm = OracleLM(lshape, C.tiny_llama_state_dict(dt), dt, C.TINY_IMAGE_TOKEN_ID)
    vision = OracleVision(cshape, clip_state(dt))
    rows, nodes = [], []
    for k, t in enumerate(topology):
        kv = CpuPoolManager(lshape.num_hidden_layers, 2, 48, BS, lshape.num_key_value_heads, lshape.head_dim, dt, 10 + k)
        img = CpuPoolManager(1, 1, 6, N_IMG_TOK, lshape.num_attention_heads, lshape.head_dim, dt, 20 + k)
        nodes.append(make_node(f"{t}{k}", t, lm, vision, kv, img, lshape, dt, torch.device("cpu"),
                               sched_cfg(chunked), rows))
    return LocalCluster(nodes), lm, rows


def hip_cluster(dt, dname, topology, chunked, graph_decode=False):
    from hydrainfer_amd.memory.token_cache_manger import (TokenCacheBlockManager, TokenCacheBlockManagerConfig,
                                                          TokenCacheBlockManagerContext)
    from hydrainfer_amd.model.clip import LlavaVisionModel
    from hydrainfer_amd.model.llama import LlamaForCausalLM
    from hydrainfer_amd.model.llava import LlavaLanguageModel
    dev = torch.device("cuda:0")
    lshape, cshape = shapes()
    lm = LogitsTap(LlavaLanguageModel(
        LlamaForCausalLM.from_reference_state_dict(lshape, C.tiny_llama_state_dict(dt), dt, dev),
        image_token_id=C.TINY_IMAGE_TOKEN_ID))
    vision = LlavaVisionModel(cshape, dt, dev, {k: v.to(dev) for k, v in clip_state(dt).items()})
    rows, nodes = [], []
    for k, t in enumerate(topology):
        ctx = TokenCacheBlockManagerContext(rank=0, rank2host={0: "localhost"})
        kv = TokenCacheBlockManager(TokenCacheBlockManagerConfig(
            n_layers=lshape.num_hidden_layers, n_tokens=2, n_blocks=48, block_size=BS,
            n_heads=lshape.num_key_value_heads, head_size=lshape.head_dim, dtype=dname, device="cuda:0"), ctx)
        img = TokenCacheBlockManager(TokenCacheBlockManagerConfig(
            n_layers=1, n_tokens=1, n_blocks=6, block_size=N_IMG_TOK, n_heads=lshape.num_attention_heads,
            head_size=lshape.head_dim, dtype=dname, device="cuda:0"), ctx)
        nodes.append(make_node(f"{t}{k}", t, lm, vision, kv, img, lshape, dt, dev, sched_cfg(chunked), rows,
                               graph_decode=graph_decode))
    return LocalCluster(nodes), lm, rows


def per_request_logits(rows, logits, n_requests):
    """rows[k] = request ids of the sampled rows of fill batch k (chunk heads included);
    returns request -> list of logits rows in generation order, chunk-head rows dropped later."""
    seq = [[] for _ in range(n_requests)]
    for ids, lg in zip(rows, logits):
        assert lg.shape[0] == len(ids)
        for j, r in enumerate(ids):
            seq[r].append(lg[j])
    return seq


def compare(run_a, run_b, reqs, tol):
    """run = (rcbs, rows, logits).  A chunk head samples a token that is thrown away, so a request
    contributes max_tokens kept rows plus one row per chunk head; the kept ones are the LAST
    max_tokens rows of the request in either run."""
    n_checked = n_flipped = 0
    for i, (_, r) in enumerate(reqs):
        k = r.sampling_params.max_tokens
        a = per_request_logits(run_a[1], run_a[2], len(reqs))[i][-k:]
        b = per_request_logits(run_b[1], run_b[2], len(reqs))[i][-k:]
        assert len(a) == len(b) == k
        ta, tb = run_a[0][i].output_token_ids, run_b[0][i].output_token_ids
        assert len(ta) == len(tb) == k
        for s in range(k):
            err = (a[s] - b[s]).abs().max().item()
            assert err <= tol, f"request {i} token {s}: logits differ by {err}"
            n_checked += 1
            if ta[s] != tb[s]:
                top = torch.topk(a[s], 2).values
                assert (top[0] - top[1]).item() <= 2 * tol, f"request {i} token {s}: greedy token differs"
                n_flipped += 1
                break
    assert n_checked >= 2 * len(reqs) and n_flipped <= len(reqs) // 2
    return n_checked


def drained(cluster):
    for node in cluster.nodes:
        for m in (node.kv_cache_block_manager, node.image_cache_block_manager):
            if m is not None:
                assert len(m.shared_cache.to_be_evicted) == m.n_blocks, f"{node.name}: blocks still pinned"
        assert node.batch_scheduler.migrating_cnt == 0


def run_oracle(dt, topology, chunked):
    cluster, lm, rows = oracle_cluster(dt, topology, chunked)
    reqs = trace_requests()
    rcbs = run_trace(cluster, creator(), reqs)
    drained(cluster)
    return (rcbs, rows, lm.logits), reqs, cluster


@pytest.mark.parametrize("topology", [["E", "P", "D"], ["EP", "D"], ["E", "P", "D", "D"]], ids="-".join)
def test_oracle_engine_disaggregated_equals_collocated(topology):
    """Host protocol on CPU: disaggregation only moves where the work happens."""
    dt = torch.float32
    base, reqs, _ = run_oracle(dt, ["EPD"], chunked=True)
    split, _, cluster = run_oracle(dt, topology, chunked=True)
    compare(base, split, reqs, tol=2e-4)
    d_nodes = [n for n in cluster.nodes if n.node_type.node_type == "D"]
    assert all(len(n.finished) > 0 for n in d_nodes)          # every D node served requests
    assert sum(len(n.finished) for n in cluster.nodes) == len(reqs)


def test_oracle_engine_prefix_hit_skips_tokens():
    run, reqs, cluster = run_oracle(torch.float32, ["EPD"], chunked=True)
    kv = cluster.nodes[0].kv_cache_block_manager
    assert kv.get_metrics().cache_hit_rate > 0
    # request 6 repeats request 0 and must generate the same tokens from partly cached blocks
    assert run[0][6].output_token_ids[:3] == run[0][0].output_token_ids[:3]


@pytest.mark.gpu
@pytest.mark.parametrize("dname", ["fp16", "bf16"])
@pytest.mark.parametrize("topology,chunked", [(["EPD"], True), (["EPD"], False), (["E", "P", "D"], True)],
                         ids=["epd-chunked", "epd-whole", "e-p-d"])
def test_hip_engine_matches_oracle_engine(dname, topology, chunked):
    dt = C.DTYPES[dname]
    base, reqs, _ = run_oracle(dt, topology, chunked)
    cluster, lm, rows = hip_cluster(dt, dname, topology, chunked)
    rcbs = run_trace(cluster, creator(), reqs)
    torch.cuda.synchronize()
    drained(cluster)
    compare(base, (rcbs, rows, lm.logits), reqs, tol=1.5e-1 if dname == "bf16" else 2e-2)


@pytest.mark.gpu
@pytest.mark.parametrize("dname", ["fp16", "bf16"])
@pytest.mark.parametrize("topology", [["EPD"], ["EP", "D"]], ids="-".join)
def test_hip_engine_graph_decode_equals_eager(dname, topology):
    """Decode-only batches replayed from hipGraphs (padded batch, static metadata buffer) generate
    exactly the tokens of the eager engine."""
    dt = C.DTYPES[dname]
    reqs = trace_requests()
    eager_cluster, _, _ = hip_cluster(dt, dname, topology, True)
    eager = run_trace(eager_cluster, creator(), reqs)
    graph_cluster, _, _ = hip_cluster(dt, dname, topology, True, graph_decode=True)
    graphed = run_trace(graph_cluster, creator(), reqs)
    torch.cuda.synchronize()
    decoders = [n.executor.fill_executor.graph_decoder for n in graph_cluster.nodes
                if n.executor.fill_executor is not None and n.executor.fill_executor.graph_decoder is not None]
    assert decoders and sum(len(d.graphs) for d in decoders) > 0, "no decode batch went through a graph"
    for i in range(len(reqs)):
        assert graphed[i].output_token_ids == eager[i].output_token_ids, f"request {i}"
