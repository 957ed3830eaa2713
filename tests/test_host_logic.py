"""CPU: integer host logic of the product (block allocator, v2p, prefix hashes,
AttentionParametersBuilder) bit-exact against the reference-generated trace G7 and the
metadata recorded in G2."""
import numpy as np
import torch

from hydrainfer_amd.layer.causal_attention import AttentionParametersBuilder
from hydrainfer_amd.memory.block_allocator import BlockAllocator
from hydrainfer_amd.memory.shared_cache import SharedCache, SharedCacheConfig, compute_hash
from hydrainfer_amd.memory.token_cache import VirtualTokenCache
from hydrainfer_amd.memory import token_cache_manger as tcm
from tests.golden import cases as C
from tests.util import load_golden

FIELDS = ("q_cu_seq_lens", "kv_cu_seq_lens", "paged_kv_last_page_len", "new_cache_slots",
          "block_tables", "cu_blocks_lens")


def _check_params(p, g, prefix):
    for f in FIELDS:
        got = getattr(p, f)
        assert got.dtype == torch.int32
        np.testing.assert_array_equal(got.numpy(), g[prefix + f], err_msg=prefix + f)
    np.testing.assert_array_equal(
        np.array([p.num_sequences, int(p.all_sequences_decode), p.q_max_seq_len, p.kv_max_seq_len]),
        g[prefix + "scalars"])


def test_attention_parameters_builder_matches_reference_g2():
    g = load_golden("g2_paged_attention")
    for i, case in enumerate(C.paged_attention_cases()):
        *_, reqs = C.paged_attention_inputs(case, seed=i)
        b = AttentionParametersBuilder(case["n_heads"], case["n_kv_heads"], case["head_dim"],
                                       case["block_size"], torch.device("cpu"))
        for r in reqs:
            b.add_request(r["q_len"], r["kv_len"], r["new_cache_slots"], r["block_table"])
        b.add_kv_cache(None)
        _check_params(b.build_attention_parameters()[0], g, C.case_name("pattn", i) + "_")


def test_continuous_batching_trace_g7():
    g = load_golden("g7_trace")
    cfg = C.TraceConfig()
    bs = cfg.block_size
    alloc = BlockAllocator(cfg.n_blocks)
    shared = SharedCache(SharedCacheConfig(n_blocks=cfg.n_blocks))
    caches = [VirtualTokenCache(vid=r + 1, n_blocks_of_cache_manager=cfg.n_blocks)
              for r in range(cfg.n_requests)]

    hashes = np.stack([np.array(compute_hash(C.trace_token_ids(cfg, r), bs, -1), dtype=np.uint64)
                       for r in range(cfg.n_requests)])
    np.testing.assert_array_equal(hashes, g["trace_hashes"])

    def step(q_lens, tag):
        b = AttentionParametersBuilder(32, 32, 128, bs, torch.device("cpu"))
        for r, q_len in enumerate(q_lens):
            vc = caches[r]
            old = vc.n_cache_tokens
            tcm.realloc(alloc, shared, vc, old + q_len, bs)
            slots = tcm.v2p(vc.block_table, list(range(old, old + q_len)), bs)
            b.add_request(q_len, vc.n_cache_tokens, slots, vc.block_table)
        b.add_kv_cache(None)
        _check_params(b.build_attention_parameters()[0], g, f"trace_{tag}_")

    step([cfg.prompt_len] * cfg.n_requests, "prefill")
    for d in range(cfg.n_decode):
        step([1] * cfg.n_requests, f"decode{d}")
    for r in (3, 17):
        shared.unpin(caches[r].block_table)
        alloc.free(caches[r].block_table)
        caches[r].block_table, caches[r].n_cache_tokens = [], 0
    tcm.realloc(alloc, shared, caches[3], 40, bs)
    tcm.realloc(alloc, shared, caches[17], 700, bs)
    np.testing.assert_array_equal(np.array(caches[3].block_table, dtype=np.int32), g["trace_realloc_3"])
    np.testing.assert_array_equal(np.array(caches[17].block_table, dtype=np.int32), g["trace_realloc_17"])
    np.testing.assert_array_equal(np.array(alloc.free_blocks[-8:], dtype=np.int32),
                                  g["trace_free_blocks_tail"])


def test_block_allocator_edges():
    a = BlockAllocator(4)
    assert a.allocate(0) == []
    assert a.allocate(3) == [2, 1, 0]
    assert a.allocate(5) == [3]          # at most n: returns what is left (block_allocator.py:25-32)
    assert a.allocate(1) == []
    a.free([1, 3])
    assert a.allocate(1) == [3]          # LIFO
    assert a.get_num_avaiable_blocks() == 1
    m = a.get_metrics()
    assert (m.n_used_blocks, m.n_total_blocks) == (3, 4)


def test_realloc_shrink_and_prefix_match():
    bs = 16
    alloc, shared = BlockAllocator(8), SharedCache(SharedCacheConfig(n_blocks=8))
    vc = VirtualTokenCache(vid=1, n_blocks_of_cache_manager=8)
    tcm.realloc(alloc, shared, vc, 40, bs)
    assert vc.block_table == [2, 1, 0] and vc.n_cache_tokens == 40
    hashes = compute_hash(list(range(40)), bs, -1)
    assert len(hashes) == 2
    shared.insert(hashes, vc.block_table[:2])
    assert shared.match(hashes + [123]) == [2, 1, -1]
    tcm.realloc(alloc, shared, vc, 16, bs)
    assert vc.block_table == [2] and shared.get_num_avaiable_blocks() == 7
    assert tcm.v2p([5, 9], [0, 15, 16, 31], bs) == [80, 95, 144, 159]


def test_ipc_safe_pool_sizing():
    blk = 8 << 20                                    # 7B: 8 MiB per block across layers and k/v
    f = tcm.ipc_safe_n_blocks
    assert f(1920, blk) == 2048                      # 15.0 GiB -> 16 GiB (window [14, 16) GiB)
    assert f(1536, blk) == 1536                      # 12 GiB is fine
    assert f(2048, blk) == 2048 and f(2049, blk) == 2049
    assert f(3840, blk) == 4096                      # 30 GiB -> 32 GiB
    assert f(1, 4096) == 1 and f(7, 4096) == 8       # generic: 7/8 of the next power of two


def test_instruction_creator_rejects_positions_beyond_the_rotary_table():
    """cos_sin has max_position_embeddings rows and the kernels index it unchecked: a request whose
    prompt + generation runs past it is refused at admission instead of reading out of bounds."""
    import pytest
    from hydrainfer_amd.engine import InstructionCreator, SamplingParameters, TokenRequest
    creator = InstructionCreator(image_token_id=32000, n_image_tokens_per_image=576, block_size=16,
                                 max_position_embeddings=1024)
    ok = TokenRequest(1, [5] * 1000, None, (8, 8), 0, SamplingParameters(max_tokens=25))
    creator.process(ok)                                     # positions 0..1023
    with pytest.raises(ValueError):
        creator.process(TokenRequest(2, [5] * 1000, None, (8, 8), 0, SamplingParameters(max_tokens=26)))
    with pytest.raises(ValueError):                         # the image placeholder expands to 576 tokens
        creator.process(TokenRequest(3, [32000] + [5] * 440, object(), (8, 8), 1, SamplingParameters(max_tokens=10)))
