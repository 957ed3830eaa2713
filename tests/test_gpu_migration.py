"""GPU: block migration (gather-copy kernel, IPC handle wire format, pack/unpack)."""
import multiprocessing as mp
import os

import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _pools(seed=0, src_blocks=10, dst_blocks=7, dtype=torch.float16):
    g = torch.Generator().manual_seed(seed)
    src = torch.randn((3, 2, src_blocks, 16, 4, 64), generator=g).to(dtype)
    dst = torch.randn((3, 2, dst_blocks, 16, 4, 64), generator=g).to(dtype)
    return src, dst


def test_migrate_blocks_local_matches_oracle():
    from hydrainfer_amd._C.data_transfer import block_migration as bm
    from oracle import ops
    for dtype in (torch.float16, torch.bfloat16, torch.float32):
        src, dst = _pools(dtype=dtype)
        s_tbl, d_tbl = [9, 0, 4, 3], [1, 6, 2, 0]
        want = dst.clone()
        ops.migrate_blocks(s_tbl, d_tbl, src, want)
        sd, dd = src.to(DEV), dst.to(DEV)
        bm.migrate_blocks_local(s_tbl, d_tbl, sd, dd)
        torch.cuda.synchronize()
        assert torch.equal(dd.cpu(), want)


def test_migrate_blocks_same_process_ipc_handle_roundtrip_format():
    from hydrainfer_amd._C.data_transfer import block_migration as bm
    src, _ = _pools()
    sd = src.to(DEV)
    h = bm.get_ipc_mem_handle(sd)
    assert isinstance(h, list) and len(h) == 72 and all(isinstance(x, int) and 0 <= x < 256 for x in h)


def test_pack_unpack_roundtrip_many_pairs():
    from hydrainfer_amd._C.data_transfer import block_migration as bm
    g = torch.Generator().manual_seed(1)
    n_blocks = 1200  # > HX_MIGRATE_MAX_PAIRS: exercises the chunked launch
    pool = torch.randn((2, 2, n_blocks, 16, 1, 16), generator=g).to(torch.float16).to(DEV)
    table = torch.randperm(n_blocks, generator=g)[:1000].tolist()
    staging = torch.empty((2, 2, len(table), 16, 1, 16), dtype=torch.float16, device=DEV)
    bm.pack_blocks(table, pool, staging)
    torch.cuda.synchronize()
    assert torch.equal(staging, pool[:, :, table])
    pool2 = torch.zeros_like(pool)
    bm.unpack_blocks(table, staging, pool2)
    torch.cuda.synchronize()
    assert torch.equal(pool2[:, :, table], pool[:, :, table])
    rest = sorted(set(range(n_blocks)) - set(table))
    assert float(pool2[:, :, rest].float().abs().sum()) == 0


def _skewed(t, skew):
    """The same values in a pool whose planes lie `skew` bytes apart (memory/kv_pool.py)."""
    from hydrainfer_amd.memory import kv_pool
    p = kv_pool.allocate_kv_pool(tuple(t.shape), t.dtype, DEV, fill="zeros", skew_bytes=skew)
    p.copy_(t)
    return p


@pytest.mark.parametrize("skews", [(768, 768), (768, 0), (0, 4352), (256, 1280)])
def test_migration_between_pools_with_planes_apart(skews):
    """hx_migrate_blocks_planes / hx_pack_blocks_planes / hx_unpack_blocks_planes: pools whose (layer, k/v) planes are
    not back to back, every mix with the reference's contiguous pool, against oracle.ops.migrate_blocks; the bytes
    between the planes are never written."""
    from hydrainfer_amd._C.data_transfer import block_migration as bm
    from hydrainfer_amd.memory import kv_pool
    from oracle import ops
    src, dst = _pools(seed=3)
    s_tbl, d_tbl = [9, 0, 4, 3], [1, 6, 2, 0]
    want = dst.clone()
    ops.migrate_blocks(s_tbl, d_tbl, src, want)
    sd, dd = _skewed(src, skews[0]), _skewed(dst, skews[1])
    assert kv_pool.plane_bytes_of(sd) == 10 * 16 * 4 * 64 * 2 + skews[0]
    bm.migrate_blocks_local(s_tbl, d_tbl, sd, dd)
    torch.cuda.synchronize()
    assert torch.equal(dd.cpu(), want)
    # through the handle (same process: the exported base is reused): the source's plane stride travels in the handle
    dd2 = _skewed(dst, skews[1])
    h = bm.get_ipc_mem_handle(sd)
    assert len(h) == (80 if skews[0] else 72) and bm.handle_plane_bytes(h) == (kv_pool.plane_bytes_of(sd) if skews[0] else 0)
    bm.migrate_blocks(s_tbl, d_tbl, h, dd2, src.shape[2])
    torch.cuda.synchronize()
    assert torch.equal(dd2.cpu(), want)
    # pack / unpack (the RCCL path's two halves)
    staging = torch.empty((3, 2, len(s_tbl), 16, 4, 64), dtype=src.dtype, device=DEV)
    bm.pack_blocks(s_tbl, sd, staging)
    dd3 = _skewed(dst, skews[1])
    bm.unpack_blocks(d_tbl, staging, dd3)
    torch.cuda.synchronize()
    assert torch.equal(staging.cpu(), src[:, :, s_tbl]) and torch.equal(dd3.cpu(), want)
    if skews[1]:      # the spare bytes behind every plane of the destination: still the zeros they were allocated with
        n = 7 * 16 * 4 * 64
        flat = torch.as_strided(dd3, (6, skews[1] // 2), (n + skews[1] // 2, 1), dd3.storage_offset() + n)
        assert float(flat.float().abs().sum()) == 0


def test_a_pool_that_is_neither_layout_is_refused():
    from hydrainfer_amd import _lib
    from hydrainfer_amd._C.data_transfer import block_migration as bm
    src, dst = _pools()
    sd = src.to(DEV)
    bad = dst.to(DEV).transpose(3, 4)          # blocks no longer contiguous
    with pytest.raises(_lib.HydraHipError):
        bm.migrate_blocks_local([0], [0], sd, bad)
    with pytest.raises(_lib.HydraHipError):
        bm.pack_blocks([0], bad, torch.empty(1 << 20, dtype=torch.float16, device=DEV))


def _ipc_child(handle, src_tbl, dst_tbl, src_n_blocks, q):
    try:
        import torch
        from hydrainfer_amd._C.data_transfer import block_migration as bm
        dst = torch.zeros((3, 2, 7, 16, 4, 64), dtype=torch.float16, device="cuda:0")
        bm.migrate_blocks(src_tbl, dst_tbl, handle, dst, src_n_blocks)
        bm.migrate_blocks(src_tbl, dst_tbl, handle, dst, src_n_blocks)  # handle re-use: cached mapping
        torch.cuda.synchronize()
        q.put(dst.cpu().numpy().tobytes())  # plain bytes: no fd passing through the queue
    except Exception as e:  # pragma: no cover
        q.put(repr(e))


def test_migrate_blocks_across_processes_via_ipc_handle():
    """The reference's transport (communication.py:23-45): the receiver process maps the
    sender's pool from 64 handle bytes passed through Python and pulls blocks."""
    from hydrainfer_amd._C.data_transfer import block_migration as bm
    from oracle import ops
    src, _ = _pools()
    sd = src.to(DEV)
    torch.cuda.synchronize()
    handle = bm.get_ipc_mem_handle(sd)
    s_tbl, d_tbl = [9, 0, 4], [1, 6, 2]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_ipc_child, args=(handle, s_tbl, d_tbl, src.shape[2], q))
    p.start()
    got = q.get(timeout=180)
    p.join(timeout=60)
    assert not isinstance(got, str), got
    want = torch.zeros((3, 2, 7, 16, 4, 64), dtype=torch.float16)
    ops.migrate_blocks(s_tbl, d_tbl, src, want)
    assert got == want.numpy().tobytes()


def test_decode_advance_matches_builder():
    """hx_decode_advance == one AttentionParametersBuilder pass for an all-decode batch."""
    import ctypes
    from hydrainfer_amd import _lib
    from hydrainfer_amd.memory import token_cache_manger as tcm
    bs, B = 16, 300
    g = torch.Generator().manual_seed(0)
    lens = torch.randint(1, 200, (B,), generator=g).tolist()
    tables, cu_b, nxt = [], [0], 0
    for l in lens:
        nb = (l + 1 + bs - 1) // bs
        tables += list(range(nxt, nxt + nb))[::-1]
        nxt += nb
        cu_b.append(cu_b[-1] + nb)
    pos = torch.tensor([l - 1 for l in lens], dtype=torch.int32, device=DEV)
    kvl = torch.tensor(lens, dtype=torch.int32, device=DEV)
    cu_k = torch.zeros(B + 1, dtype=torch.int32, device=DEV)
    slots = torch.zeros(B, dtype=torch.int32, device=DEV)
    bt = torch.tensor(tables, dtype=torch.int32, device=DEV)
    cub = torch.tensor(cu_b, dtype=torch.int32, device=DEV)
    _lib.check(_lib.lib().hx_decode_advance(pos.data_ptr(), kvl.data_ptr(), cu_k.data_ptr(),
                                            slots.data_ptr(), bt.data_ptr(), cub.data_ptr(), B, bs, 1,
                                            _lib.current_stream()), "decode_advance")
    torch.cuda.synchronize()
    want_slots = [tcm.v2p(tables[cu_b[i]:cu_b[i + 1]], [lens[i]], bs)[0] for i in range(B)]
    assert slots.tolist() == want_slots
    assert kvl.tolist() == [l + 1 for l in lens]
    assert pos.tolist() == lens
    assert cu_k.tolist() == [0] + torch.tensor([l + 1 for l in lens]).cumsum(0).tolist()


def test_token_cache_block_manager_end_to_end():
    """TokenCacheBlockManager (hydrainfer/memory/token_cache_manger.py:51-179 mirror): pool
    layout, allocation, v2p, per-layer caches feeding set_kv_cache, and a same-process P->D
    migration through the manager's IPC backend (handle bytes -> cached mapping -> one gather
    kernel on the migrate stream)."""
    from hydrainfer_amd.memory import (KVCache, TokenCacheBlockManager, TokenCacheBlockManagerConfig,
                                       TokenCacheBlockManagerContext)
    cfg = dict(n_layers=3, n_tokens=2, block_size=16, n_heads=4, head_size=64, dtype="fp16", device=DEV)
    ctx = TokenCacheBlockManagerContext(rank=0, rank2host={0: "h", 1: "h"})
    p_mgr = TokenCacheBlockManager(TokenCacheBlockManagerConfig(n_blocks=12, **cfg), ctx)
    d_mgr = TokenCacheBlockManager(TokenCacheBlockManagerConfig(n_blocks=9, **cfg),
                                   TokenCacheBlockManagerContext(rank=1, rank2host={0: "h", 1: "h"}))
    # 64 handle bytes + 8 offset bytes + 8 bytes of plane stride: the manager's pool keeps its (layer, k/v) planes
    # KV_POOL_SKEW_BYTES apart (memory/kv_pool.py)
    from hydrainfer_amd.memory import kv_pool
    from hydrainfer_amd._C.data_transfer.block_migration import handle_plane_bytes
    assert p_mgr.cache_tensor.shape == (3, 2, 12, 16, 4, 64) and len(p_mgr.memory_handle) == 80
    assert handle_plane_bytes(p_mgr.memory_handle) == 12 * 16 * 4 * 64 * 2 + kv_pool.KV_POOL_SKEW_BYTES
    assert not p_mgr.cache_tensor.is_contiguous() and p_mgr.cache_tensor[1, 1].is_contiguous()
    src = p_mgr.allocate_virtual_cache()
    p_mgr.realloc(src, 40)
    assert src.block_table == [2, 1, 0] and p_mgr.v2p(src, [0, 17, 39]) == [32, 17, 7]
    # prefill-side: write 40 tokens into every layer through KVCache views of the pool
    slots = torch.tensor(p_mgr.v2p(src, list(range(40))), dtype=torch.int32, device=DEV)
    for l in range(3):
        kv = KVCache.from_token_cache(p_mgr.get_layer_cache(l))
        k = torch.randn((40, 4, 64), device=DEV).half()
        kv.set_kv_cache(slots, k, -k)
    dst = d_mgr.allocate_virtual_cache()
    d_mgr.realloc(dst, 40)
    dst_before = d_mgr.cache_tensor.clone()
    d_mgr.migrate_blocks(src, dst, is_send=False)       # pull model: receiver moves the data
    p_mgr.migrate_blocks(src, dst, is_send=True)        # sender side is a no-op for ipc
    d_mgr.synchronize()
    for s, d in zip(src.block_table, dst.block_table):
        assert torch.equal(d_mgr.cache_tensor[:, :, d], p_mgr.cache_tensor[:, :, s])
    rest = [b for b in range(9) if b not in dst.block_table]
    assert torch.equal(d_mgr.cache_tensor[:, :, rest], dst_before[:, :, rest])
    d_mgr.realloc(dst, 10)
    assert len(dst.block_table) == 1
    # reference accounting (token_cache_manger.py:90-91): allocator free list (6) + unpinned
    # shared-cache blocks (8) — free blocks are counted by both, exactly as in the reference
    assert d_mgr.get_num_avaiable_blocks() == 14
