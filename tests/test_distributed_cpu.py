"""CPU, world_size 2, gloo: the N>1 control flow of the replica layout (barrier, max-over-ranks
timing, handle exchange) and the packed point-to-point migration protocol of
memory/communication.py::RCCLBackend with the HIP pack/unpack kernels replaced by their index
semantics (oracle) — the GPU data path itself is covered by tests/test_gpu_migration.py."""
import os
import socket
import time

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
        # agreement without a collective (bench.py's exit when an optional leg is wedged on some rank): any rank's flag
        # is seen by all, through the rendezvous store
        assert ctx.any_rank_flagged(False, "t0") is False
        assert ctx.any_rank_flagged(rank == 1, "t1") is True
        # bench.py's last exchange: every rank's view of its optional legs, as strings through the store
        assert ctx.gather_via_store("g0", f"view of rank {rank}") == ["view of rank 0", "view of rank 1"]
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
        # node start-up: the communicators of the send/recv hops are created under a bound (two hosts -> send/recv backend)
        from hydrainfer_amd.memory.communication import (CommunicationBackendManager, CommunicationBackendManagerConfig,
                                                         CommunicationBackendManagerContext)
        mgr = CommunicationBackendManager(CommunicationBackendManagerConfig(),
                                          CommunicationBackendManagerContext(None, pool, pool.shape[2], {0: "hostA", 1: "hostB"}))
        assert mgr.connect_peers(rank, [(0, 1), (0, 1), (1, 1)], timeout_s=60) == 1
        same = CommunicationBackendManager(CommunicationBackendManagerConfig(),
                                           CommunicationBackendManagerContext(None, pool, pool.shape[2], {0: "h", 1: "h"}))
        assert same.connect_peers(rank, [(0, 1)], timeout_s=60) == 0       # the IPC pull has no sender half
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


# ---------------------------------------------------------------------------------------------------------------
# negative paths of the send/recv migration: they must fail LOUDLY and BOUNDED on the first
# multi-GPU run, not hang it.  Reference: hydrainfer/memory/communication.py:50-74 (waits for ever),
# hydrainfer/cluster/epdnode.py:362-447.
# ---------------------------------------------------------------------------------------------------------------
def _stub_pack_unpack():
    from hydrainfer_amd._C.data_transfer import block_migration as bm

    def pack(table, cache, staging):     # index semantics of hx_pack_blocks
        staging.view(cache.shape[0], cache.shape[1], len(table), *cache.shape[3:]).copy_(cache[:, :, table])

    def unpack(table, staging, cache):   # index semantics of hx_unpack_blocks
        cache[:, :, table] = staging.view(cache.shape[0], cache.shape[1], len(table), *cache.shape[3:])
    bm.pack_blocks, bm.unpack_blocks = pack, unpack


def _pull_worker(rank, world, port, q, sender_dies, rank2host):
    """rank 0 = sender (P), rank 1 = receiver (D).  The transfer goes through CommunicationBackendManager: with
    rank2host naming two different hosts it must pick the send/recv backend (RCCL on GPUs, gloo here)."""
    try:
        os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank),
                          MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        import torch.distributed as dist
        from hydrainfer_amd import parallel
        from hydrainfer_amd.memory.communication import (CommunicationBackendManager, CommunicationBackendManagerConfig,
                                                         CommunicationBackendManagerContext, IPCHandleMemoryBackend,
                                                         MigrationTimeout, RCCLBackend)
        from hydrainfer_amd.memory.token_cache import VirtualTokenCache
        ctx = parallel.init_from_env("gloo")
        _stub_pack_unpack()
        g = torch.Generator().manual_seed(5)
        src_pool = torch.randn((2, 2, 6, 4, 2, 8), generator=g)
        pool = src_pool.clone() if rank == 0 else torch.zeros((2, 2, 9, 4, 2, 8))
        mgr = CommunicationBackendManager(CommunicationBackendManagerConfig(),
                                          CommunicationBackendManagerContext(None, pool, pool.shape[2], rank2host))
        same = rank2host[0] == rank2host[1]
        assert mgr.in_same_machine(0, 1) == same and mgr.needs_sender(0, 1) == (not same)
        assert isinstance(mgr.intranode_backend, IPCHandleMemoryBackend) and isinstance(mgr.internode_backend, RCCLBackend)
        assert not mgr.in_same_machine(0, 7)                     # a rank nobody told us about is never "local"
        mgr.internode_backend.timeout_s = 3.0
        src = VirtualTokenCache(vid=1, n_blocks_of_cache_manager=6, n_cache_tokens=12, block_table=[5, 0, 2], rank=0)
        dst = VirtualTokenCache(vid=7, n_blocks_of_cache_manager=9, n_cache_tokens=12, block_table=[8, 1, 4], rank=1)
        ctx.barrier()
        if not sender_dies:      # start-up: the pair's communicator is created under a bound, before any transfer
            mgr.internode_backend.connect(peer=1 - rank, my_rank=rank, timeout_s=20.0)
        if rank == 0 and sender_dies:
            q.put((rank, "died")); q.close(); q.join_thread()      # flush the feeder thread: _exit would drop the item
            os._exit(17)                                         # the sender is gone before its half of the transfer
        t0 = time.monotonic()
        try:
            mgr.migrate_blocks(src, dst, is_send=(rank == 0))
        except MigrationTimeout as e:
            q.put((rank, f"timeout after {time.monotonic() - t0:.1f}s")); q.close(); q.join_thread()
            os._exit(3)                                          # the rank fails, non-zero, without collective teardown
        if rank == 1:
            for s_, d_ in zip(src.block_table, dst.block_table):
                assert torch.equal(pool[:, :, d_], src_pool[:, :, s_])
        ctx.shutdown()
        q.put((rank, "ok"))
    except Exception:  # pragma: no cover
        import traceback
        q.put((rank, traceback.format_exc()))


def _run_pull(sender_dies, rank2host):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_pull_worker, args=(r, 2, port, q, sender_dies, rank2host)) for r in range(2)]
    for p in procs:
        p.start()
    results = dict(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
    return results, [p.exitcode for p in procs]


def test_send_recv_backend_is_selected_by_rank2host_and_moves_the_blocks():
    """Ranks on different hosts: CommunicationBackendManager routes the pull through the packed send/recv
    backend (RCCL on GPUs; gloo stands in on CPU), both halves run, the blocks arrive."""
    results, codes = _run_pull(False, {0: "node-a", 1: "node-b"})
    assert results == {0: "ok", 1: "ok"} and codes == [0, 0], (results, codes)


def test_pull_whose_sender_died_fails_bounded_and_a_fresh_attempt_succeeds():
    """The sender exits before its half of a send/recv transfer: the receiver's wait is bounded (3 s here), it
    raises MigrationTimeout and its rank exits non-zero — no hang.  Recovery is a FRESH pair of processes (spawned
    children, nothing is re-exec'd): the same transfer then goes through."""
    import time as _t
    t0 = _t.monotonic()
    results, codes = _run_pull(True, {0: "node-a", 1: "node-b"})
    assert results[0] == "died" and results[1].startswith("timeout"), results
    assert codes[0] == 17 and codes[1] == 3, codes
    assert _t.monotonic() - t0 < 60
    results, codes = _run_pull(False, {0: "node-a", 1: "node-b"})
    assert results == {0: "ok", 1: "ok"} and codes == [0, 0], (results, codes)
