"""CPU, world_size 2, gloo: the N>1 control flow of the replica layout (barrier, max-over-ranks
timing, handle exchange) and the packed point-to-point migration protocol of
memory/communication.py::RCCLBackend with the HIP pack/unpack kernels replaced by their index
semantics (oracle) — the GPU data path itself is covered by tests/test_gpu_migration.py."""
import os
import socket

import pytest
import torch
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    try:
        os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank),
                          MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        import torch.distributed as dist
        from hydrainfer_amd import parallel
        ctx = parallel.init_from_env("gloo")
        assert ctx.enabled and ctx.world_size == 2
        # contract: timed region bracketed by barriers, MAX over ranks
        ctx.barrier()
        assert ctx.max_over_ranks(1.0 + rank) == 2.0
        assert ctx.sum_over_ranks(32.0) == 64.0
        handles = ctx.all_gather_object({"rank": rank, "handle": [rank] * 72, "table": [3 + rank, 1]})
        assert [h["rank"] for h in handles] == [0, 1]
        peer = parallel.migration_peer(rank, world)
        assert handles[peer]["handle"] == [peer] * 72

        # packed migration protocol over send/recv: rank 0 (P) -> rank 1 (D)
        from hydrainfer_amd._C.data_transfer import block_migration as bm
        from hydrainfer_amd.memory.communication import RCCLBackend
        from hydrainfer_amd.memory.token_cache import VirtualTokenCache

        def pack(table, cache, staging):     # index semantics of hx_pack_blocks
            staging.view(cache.shape[0], cache.shape[1], len(table), *cache.shape[3:]).copy_(cache[:, :, table])

        def unpack(table, staging, cache):   # index semantics of hx_unpack_blocks
            cache[:, :, table] = staging.view(cache.shape[0], cache.shape[1], len(table), *cache.shape[3:])
        bm.pack_blocks, bm.unpack_blocks = pack, unpack
        g = torch.Generator().manual_seed(5)
        src_pool = torch.randn((2, 2, 6, 4, 2, 8), generator=g)   # same values on both ranks
        pool = src_pool.clone() if rank == 0 else torch.zeros((2, 2, 9, 4, 2, 8))
        backend = RCCLBackend(None, pool)
        src = VirtualTokenCache(vid=1, n_blocks_of_cache_manager=6, n_cache_tokens=12, block_table=[5, 0, 2], rank=0)
        dst = VirtualTokenCache(vid=7, n_blocks_of_cache_manager=9, n_cache_tokens=12, block_table=[8, 1, 4], rank=1)
        backend.migrate_blocks(src, dst, is_send=(rank == 0))
        if rank == 1:
            for s, d in zip(src.block_table, dst.block_table):
                assert torch.equal(pool[:, :, d], src_pool[:, :, s])
            untouched = [b for b in range(9) if b not in dst.block_table]
            assert float(pool[:, :, untouched].abs().sum()) == 0
        ctx.shutdown()
        q.put((rank, "ok"))
    except Exception as e:  # pragma: no cover
        import traceback
        q.put((rank, traceback.format_exc()))


def test_two_rank_gloo_control_flow_and_packed_migration():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    results = [q.get(timeout=240) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    assert sorted(results) == [(0, "ok"), (1, "ok")], results


def test_epd_roles():
    from hydrainfer_amd.parallel import epd_roles
    assert epd_roles(1) == ["EPD"] and epd_roles(2) == ["EP", "D"] and epd_roles(3) == ["E", "P", "D"]
    assert epd_roles(8).count("D") == 4 and epd_roles(8).count("E") == 2 and epd_roles(8).count("P") == 2
