"""-m gpu, needs >= 2 GPUs (skipped on the one-GPU boxes): the send/recv transfer path of
memory/communication.py::RCCLBackend on real devices — pack kernel -> ONE dist.send / dist.recv over
RCCL (backend "nccl") -> unpack kernel, rank 0 (a P node's pool on cuda:0) to rank 1 (a D node's
pool on cuda:1) — byte for byte against oracle.ops.migrate_blocks.  Reference path:
hydrainfer/memory/communication.py:50-74 (one P2POp per (block, layer, k/v) view there)."""
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


def _worker(rank, port, q):
    try:
        import torch.distributed as dist
        from hydrainfer_amd.memory.communication import RCCLBackend
        from hydrainfer_amd.memory.token_cache import VirtualTokenCache
        from oracle import ops
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        dev = torch.device(f"cuda:{rank}")
        torch.cuda.set_device(dev)
        dist.init_process_group("nccl", rank=rank, world_size=2, init_method=f"tcp://127.0.0.1:{port}",
                                device_id=dev)
        g = torch.Generator().manual_seed(11)
        # LLaVA-1.5-7B block geometry on 4 layers: (L, 2, n_blocks, 16, 32, 128) bf16, pools of different size
        src_cpu = torch.randn((4, 2, 24, 16, 32, 128), generator=g).to(torch.bfloat16)
        dst_cpu = torch.randn((4, 2, 40, 16, 32, 128), generator=g).to(torch.bfloat16)
        src_tbl = [23, 0, 7, 7 + 1, 15, 3, 19, 11, 2, 21]
        dst_tbl = [39, 5, 0, 17, 18, 30, 8, 1, 22, 33]
        pool = (src_cpu if rank == 0 else dst_cpu).to(dev)
        stream = torch.cuda.Stream(device=dev)
        backend = RCCLBackend(stream, pool)
        src = VirtualTokenCache(vid=1, n_blocks_of_cache_manager=24, n_cache_tokens=16 * len(src_tbl),
                                block_table=src_tbl, rank=0)
        dst = VirtualTokenCache(vid=2, n_blocks_of_cache_manager=40, n_cache_tokens=16 * len(dst_tbl),
                                block_table=dst_tbl, rank=1)
        for _ in range(2):                              # second call reuses the staging buffer
            backend.migrate_blocks(src, dst, is_send=(rank == 0))
            stream.synchronize()
        if rank == 1:
            want = dst_cpu.clone()
            ops.migrate_blocks(src_tbl, dst_tbl, src_cpu, want)
            assert torch.equal(pool.cpu().view(torch.int16), want.view(torch.int16)), "migrated bytes differ"
        else:
            assert torch.equal(pool.cpu().view(torch.int16), src_cpu.view(torch.int16)), "the sender's pool changed"
        dist.barrier()
        dist.destroy_process_group()
        q.put((rank, "ok"))
    except Exception:  # pragma: no cover
        import traceback
        q.put((rank, traceback.format_exc()))


@pytest.mark.gpu
@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs two GPUs (RCCL refuses two ranks on one device)")
def test_rccl_backend_moves_blocks_between_two_gpus():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    results = [q.get(timeout=300) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    assert sorted(results) == [(0, "ok"), (1, "ok")], results
