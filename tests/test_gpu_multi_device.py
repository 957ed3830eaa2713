"""-m gpu, tests that switch themselves on when the box has >= 2 GPUs (round-4 review, item 2) and are collected-and-
skipped on the one-GPU boxes — nothing else has to change when a multi-GPU node appears:

  * the PRIMARY intra-node transport of the reference (hydrainfer/memory/communication.py:23-45 over
    csrc/data_transfer/block_migration.cpp:194-245): a D process on cuda:0 maps the pool of a P process that lives on
    cuda:1 from its 64 handle bytes and PULLS blocks over xGMI with ONE hx_migrate_blocks launch — byte for byte
    against oracle.ops.migrate_blocks, GB/s printed (one xGMI link is ~153 GB/s);
  * `bench.py --gpus 2 --model tiny` on two real devices over RCCL (no HX_SINGLE_DEVICE, backend "nccl"): the driver's
    launch contract with the P -> D `migration` and EP + D `disaggregated` legs in the one JSON line and an empty
    top-level `legs_failed`.
(The E+P+D engine with one process per device is tests/test_gpu_distributed_engine.py::test_engine_nodes_one_process_per_gpu;
the RCCL send/recv fallback is tests/test_gpu_rccl_migration.py.)"""
import json
import os
import subprocess
import sys

import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
needs_two = pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs two GPUs (enables itself on a multi-GPU box)")

# LLaVA-1.5-7B block geometry (16 tokens x 32 heads x 128 x bf16 = 128 KiB per block and tensor) on 8 layers; the 44 blocks
# of one 704-token prompt: 8 x 2 x 44 x 128 KiB = 92 MB per pull
SHAPE_SRC, SHAPE_DST = (8, 2, 64, 16, 32, 128), (8, 2, 96, 16, 32, 128)


def _tables():
    g = torch.Generator().manual_seed(3)
    return torch.randperm(64, generator=g)[:44].tolist(), torch.randperm(96, generator=g)[:44].tolist()


def _fill(shape, seed):
    # deterministic bytes that both processes can rebuild without shipping 100 MB through a queue
    g = torch.Generator().manual_seed(seed)
    return torch.randint(-32768, 32767, shape, generator=g, dtype=torch.int16).view(torch.bfloat16)


def _owner(device_index, handle_q, done_evt):
    """The P side: owns the source pool on cuda:<device_index>, publishes its handle, stays alive until the pull is done."""
    try:
        from hydrainfer_amd._C.data_transfer import block_migration as bm
        dev = torch.device(f"cuda:{device_index}")
        torch.cuda.set_device(dev)
        pool = _fill(SHAPE_SRC, 21).to(dev)
        torch.cuda.synchronize(dev)
        handle_q.put(bm.get_ipc_mem_handle(pool))
        done_evt.wait(timeout=300)
        # the pull must not have written to the source
        ok = torch.equal(pool.cpu().view(torch.int16), _fill(SHAPE_SRC, 21).view(torch.int16))
        handle_q.put("unchanged" if ok else "the source pool changed")
    except Exception:  # pragma: no cover
        import traceback
        handle_q.put(traceback.format_exc())


@needs_two
def test_ipc_pull_from_pool_on_another_gpu():
    from hydrainfer_amd._C.data_transfer import block_migration as bm
    from oracle import ops
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    ctx = mp.get_context("spawn")
    hq, done = ctx.Queue(), ctx.Event()
    owner = ctx.Process(target=_owner, args=(1, hq, done))
    owner.start()
    try:
        handle = hq.get(timeout=300)
        assert isinstance(handle, list), handle
        dev = torch.device("cuda:0")
        torch.cuda.set_device(dev)
        dst_cpu = _fill(SHAPE_DST, 22)
        dst = dst_cpu.to(dev)
        src_tbl, dst_tbl = _tables()
        bm.migrate_blocks(src_tbl, dst_tbl, handle, dst, SHAPE_SRC[2])          # first call maps the peer pool
        torch.cuda.synchronize(dev)
        want = dst_cpu.clone()
        ops.migrate_blocks(src_tbl, dst_tbl, _fill(SHAPE_SRC, 21), want)
        assert torch.equal(dst.cpu().view(torch.int16), want.view(torch.int16)), "pulled bytes differ from the oracle"
        # rate of the pull (cached mapping), HIP events on the launch stream
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        reps = 5
        e0.record()
        for _ in range(reps):
            bm.migrate_blocks(src_tbl, dst_tbl, handle, dst, SHAPE_SRC[2])
        e1.record()
        e1.synchronize()
        nbytes = SHAPE_SRC[0] * 2 * len(src_tbl) * 16 * 32 * 128 * 2
        gbs = nbytes * reps / (e0.elapsed_time(e1) * 1e-3) / 1e9
        print(f"\nIPC pull cuda:1 -> cuda:0: {nbytes / 1e6:.1f} MB per launch, {gbs:.1f} GB/s (one xGMI link ~153 GB/s)")
        assert torch.equal(dst.cpu().view(torch.int16), want.view(torch.int16))
        assert gbs > 10, f"{gbs:.1f} GB/s: the pull is not going over a direct link"
    finally:
        done.set()
        try:
            verdict = hq.get(timeout=120)
        finally:
            owner.join(timeout=60)
            if owner.is_alive():
                owner.kill()
    assert verdict == "unchanged", verdict


@needs_two
def test_bench_two_ranks_on_two_devices_over_rccl():
    env = {k: v for k, v in os.environ.items()
           if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "HX_SINGLE_DEVICE", "HX_DIST_BACKEND")}
    env.update(HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--model", "tiny", "--batch", "8",
                        "--steps", "8", "--warmup", "2", "--rate", "40", "--no-cpu-baseline", "--no-13b"],
                       capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 8 and d["value"] > 0
    assert d["legs_failed"] == [], d["legs_failed"]
    mg, dg = d["migration"], d["disaggregated"]
    assert mg is not None and "error" not in mg, mg
    assert dg is not None and "error" not in dg, dg
    assert dg["roles"] == ["EP", "D"] and dg["output_tokens"] == 8 * 256
