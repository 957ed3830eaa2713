"""E / P / D engine nodes in separate processes sharing one GPU: request state over the store
mailbox, image and KV blocks pulled through the IPC-mapped peer pools by hx_migrate_blocks, decode
replayed from hipGraphs — the multi-GPU serving path with every GPU standing in as cuda:0."""
import os
import socket

import pytest
import torch
import torch.multiprocessing as mp

from tests.golden import cases as C


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, roles, port, q, per_device=False):
    try:
        import time
        import torch.distributed as dist
        from hydrainfer_amd._C.data_transfer import block_migration as bm
        from hydrainfer_amd.engine.distributed import RankEngine, replay_distributed
        from hydrainfer_amd.engine.node import LocalCluster
        from hydrainfer_amd.engine.scheduler import BatchSchedulerConfig
        from hydrainfer_amd.engine.serve import build_node
        from hydrainfer_amd.model.clip import ClipShape, LlavaVisionModel, random_state_dict
        from hydrainfer_amd.model.llama import LlamaForCausalLM, LlamaShape
        from hydrainfer_amd.model.llava import LlavaLanguageModel
        from tests.engine_util import run_trace
        from tests.test_engine_e2e import N_IMG_TOK, creator, trace_requests
        world = len(roles)
        if os.environ.get("HX_TEST_STACKS"):     # debugging aid: dump all stacks of a stuck rank
            import faulthandler
            faulthandler.dump_traceback_later(60, file=open(f"{os.environ['HX_TEST_STACKS']}_{rank}.txt", "w"))
        dist.init_process_group("gloo", rank=rank, world_size=world, init_method=f"tcp://127.0.0.1:{port}")
        # per_device: one process per GPU (BASELINE configs[3]); otherwise every rank stands in on cuda:0
        dev, dt = torch.device(f"cuda:{rank}" if per_device else "cuda:0"), torch.float16
        torch.cuda.set_device(dev)
        lshape, cshape = LlamaShape(**C.TINY_LLAMA), ClipShape(**C.TINY_CLIP)
        lm = LlavaLanguageModel(LlamaForCausalLM.from_reference_state_dict(lshape, C.tiny_llama_state_dict(dt), dt, dev),
                                image_token_id=C.TINY_IMAGE_TOKEN_ID)
        vision = LlavaVisionModel(cshape, dt, dev, {k: v.to(dt).to(dev) for k, v in
                                                    random_state_dict(cshape, seed=3, std=0.05).items()})
        sched = BatchSchedulerConfig(priority="prefill", max_running_requests=6, chunked_prefill=True,
                                     token_budgets=40, image_budgets=2)

        def node(role, r, graph, model=None):
            return build_node(f"{role}{r}", role, model or lm, vision, lshape, dt, dev, 96, 14, N_IMG_TOK, sched,
                              rank=r, graph_decode=graph, max_blocks_per_seq=8, world_size=world)

        engine = RankEngine(rank, roles, node(roles[rank], rank, True), None)
        n = engine.node
        pools = [None] * world
        dist.all_gather_object(pools, {
            "kv": n.kv_cache_block_manager.memory_handle if n.kv_cache_block_manager else None,
            "image": n.image_cache_block_manager.memory_handle if n.image_cache_block_manager else None})
        for r, role in enumerate(roles):          # map peers' pools before any graph exists
            if r != rank and n.node_type.enable_prefill and "E" in role:
                bm._open(pools[r]["image"])
            if r != rank and n.node_type.enable_decode and "P" in role:
                bm._open(pools[r]["kv"])
        reqs = [r for _, r in trace_requests()]
        box = [time.perf_counter() + 0.1]
        dist.broadcast_object_list(box, src=0)
        engine.open_mailbox("t")
        dist.barrier()
        mine = replay_distributed(engine, creator(), reqs, [0.01 * i for i in range(len(reqs))], box[0], dev,
                                  deadline_s=120)
        for m in (n.kv_cache_block_manager, n.image_cache_block_manager):
            if m is not None:
                pinned = m.n_blocks - len(m.shared_cache.to_be_evicted)
                assert pinned == (1 if m is n.kv_cache_block_manager and n.node_type.enable_decode else 0)
        allr = [None] * world
        dist.all_gather_object(allr, mine)
        if rank == 0:
            merged = {}
            for m in allr:
                merged.update(m)
            # The same trace through ONE process (same kernels, other batch compositions), with the
            # logits of every sampled row kept.  north_star: greedy tokens identical — asserted
            # wherever the arithmetic can decide it: a request's tokens must equal the single-process
            # run's up to the first step whose top-1 margin there is within 2x the logit tolerance
            # of the tiny model (fp16 2e-2, DESIGN.md section 2: batch composition changes split-K /
            # tile order, not the math); past such a near-tie the continuations legitimately differ.
            from tests.engine_util import LogitsTap
            from tests.test_engine_e2e import per_request_logits
            tap, rows = LogitsTap(lm), []
            ref_node = node("EPD", 0, False, model=tap)
            fe = ref_node.executor.fill_executor
            real = fe.execute

            def execute(batch):
                rows.append([rcb.request_id for rcb, inst in batch if inst.sample])
                real(batch)
            fe.execute = execute
            single = run_trace(LocalCluster([ref_node]), creator(), trace_requests())
            seq = per_request_logits(rows, tap.logits, len(reqs))
            tol = 2e-2
            n_equal = n_near_tie = 0
            for i, r in enumerate(reqs):
                k = r.sampling_params.max_tokens
                got, want = merged[i]["tokens"], single[i].output_token_ids
                assert len(got) == len(want) == k
                assert len(merged[i]["pd_transfer"]) == 2 or len(merged[i]["ep_transfer"]) == 2
                kept = seq[i][-k:]                         # chunk heads sample rows that are thrown away
                for s_ in range(k):
                    if got[s_] != want[s_]:
                        top = torch.topk(kept[s_], 2).values
                        margin = (top[0] - top[1]).item()
                        assert margin <= 2 * tol, (f"request {i} token {s_}: {got[s_]} != {want[s_]} although the "
                                                   f"single-process top-1 margin is {margin:.4f}")
                        n_near_tie += 1
                        break
                else:
                    n_equal += 1
            assert n_equal >= (len(reqs) + 1) // 2, f"only {n_equal} of {len(reqs)} requests token-identical"
        dist.barrier()
        dist.destroy_process_group()
        q.put((rank, "ok"))
    except Exception:  # pragma: no cover
        import traceback
        if os.environ.get("HX_TEST_STACKS"):
            open(f"{os.environ['HX_TEST_STACKS']}_exc_{rank}.txt", "w").write(traceback.format_exc())
        q.put((rank, traceback.format_exc()))


def _run(roles, per_device):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, roles, port, q, per_device)) for r in range(len(roles))]
    for p in procs:
        p.start()
    try:
        results = [q.get(timeout=300) for _ in procs]
    finally:
        for p in procs:
            p.join(timeout=30)
            if p.is_alive():
                p.kill()
    bad = [f"rank {r}: {msg[-1500:]}" for r, msg in results if msg != "ok"]
    assert not bad, "\n".join(bad)


@pytest.mark.gpu
@pytest.mark.parametrize("roles", [["EP", "D"], ["E", "P", "D"]], ids="-".join)
def test_engine_nodes_in_processes_share_one_gpu(roles):
    _run(roles, per_device=False)


@pytest.mark.gpu
@pytest.mark.parametrize("roles", [["EP", "D"], ["E", "P", "D"]], ids="-".join)
def test_engine_nodes_one_process_per_gpu(roles):
    """BASELINE configs[3] on real devices: rank r owns cuda:r, image blocks E -> P and KV blocks P -> D are pulled over
    xGMI through the IPC-mapped peer pools (hydrainfer/cluster/epdnode.py:362-447).  Enables itself on a box with enough
    GPUs; the one-GPU boxes skip it."""
    if torch.cuda.device_count() < len(roles):
        pytest.skip(f"needs {len(roles)} GPUs, this box has {torch.cuda.device_count()}")
    _run(roles, per_device=True)


# ---------------------------------------------------------------------------------------------------------------------
# A P rank dies in the middle of a trace, with the real kernels and real IPC pulls (the CPU twin with the closed-form
# model: tests/test_distributed_engine_cpu.py::test_a_rank_dies_mid_trace_and_only_its_requests_end).  The rank that
# dies has touched the GPU: it leaves with os._exit (non-zero), nothing is re-exec'd, nobody waits for it.
# ---------------------------------------------------------------------------------------------------------------------
def _kill_worker(rank, roles, port, q, victim):
    try:
        import time
        os.environ["HX_PEER_DEAD_AFTER_S"] = "2.0"
        import torch.distributed as dist
        from hydrainfer_amd._C.data_transfer import block_migration as bm
        from hydrainfer_amd.engine.distributed import RankEngine, replay_distributed
        from hydrainfer_amd.engine.scheduler import BatchSchedulerConfig
        from hydrainfer_amd.engine.serve import build_node
        from hydrainfer_amd.model.clip import ClipShape, LlavaVisionModel, random_state_dict
        from hydrainfer_amd.model.llama import LlamaForCausalLM, LlamaShape
        from hydrainfer_amd.model.llava import LlavaLanguageModel
        from tests.test_engine_e2e import N_IMG_TOK, creator, trace_requests
        world = len(roles)
        dist.init_process_group("gloo", rank=rank, world_size=world, init_method=f"tcp://127.0.0.1:{port}")
        dev, dt = torch.device("cuda:0"), torch.float16
        torch.cuda.set_device(dev)
        lshape, cshape = LlamaShape(**C.TINY_LLAMA), ClipShape(**C.TINY_CLIP)
        lm = LlavaLanguageModel(LlamaForCausalLM.from_reference_state_dict(lshape, C.tiny_llama_state_dict(dt), dt, dev),
                                image_token_id=C.TINY_IMAGE_TOKEN_ID)
        vision = LlavaVisionModel(cshape, dt, dev, {k: v.to(dt).to(dev) for k, v in
                                                    random_state_dict(cshape, seed=3, std=0.05).items()})
        sched = BatchSchedulerConfig(priority="prefill", max_running_requests=6, chunked_prefill=True,
                                     token_budgets=40, image_budgets=2)
        engine = RankEngine(rank, roles, build_node(f"{roles[rank]}{rank}", roles[rank], lm, vision, lshape, dt, dev, 96, 14,
                                                    N_IMG_TOK, sched, rank=rank, graph_decode=True, max_blocks_per_seq=8,
                                                    world_size=world), None)
        n = engine.node
        pools = [None] * world
        dist.all_gather_object(pools, {
            "kv": n.kv_cache_block_manager.memory_handle if n.kv_cache_block_manager else None,
            "image": n.image_cache_block_manager.memory_handle if n.image_cache_block_manager else None})
        for r, role in enumerate(roles):          # map peers' pools before any graph exists
            if r != rank and n.node_type.enable_prefill and "E" in role:
                bm._open(pools[r]["image"])
            if r != rank and n.node_type.enable_decode and "P" in role:
                bm._open(pools[r]["kv"])
        # warm-up epoch with everybody alive: graph captures, lazy library loads (seconds on a fresh box) happen HERE, so
        # that the timed epoch below runs at the engine's real pace and "late" really is after the death
        box = [time.perf_counter() + 0.1]
        dist.broadcast_object_list(box, src=0)
        engine.open_mailbox("warm")
        dist.barrier()
        warm = [r for _, r in trace_requests()]
        replay_distributed(engine, creator(), warm, [0.01 * i for i in range(len(warm))], box[0], dev, deadline_s=120)
        dist.barrier()
        if rank == victim:
            real_step, real_deliver = engine.step, engine._deliver

            def step():
                s_ = engine.node.batch_scheduler
                if getattr(engine, "n_freed", 0) >= 2 and (s_.waiting or s_.running or engine.held):
                    torch.cuda.synchronize(dev)
                    os._exit(17)
                return real_step()

            def deliver(src, kind, payload):
                if kind == "free":
                    engine.n_freed = getattr(engine, "n_freed", 0) + 1
                real_deliver(src, kind, payload)
            engine.step, engine._deliver = step, deliver
        base = [r for _, r in trace_requests()]
        late = [r for _, r in trace_requests()]
        for i, r in enumerate(late):
            r.request_id = len(base) + i
        reqs = base + late
        arrivals = [0.01 * i for i in range(len(base))] + [6.0 + 0.01 * i for i in range(len(late))]
        box = [time.perf_counter() + 0.1]
        dist.broadcast_object_list(box, src=0)
        engine.open_mailbox("kill")
        dist.barrier()
        mine = replay_distributed(engine, creator(), reqs, arrivals, box[0], dev, deadline_s=120)
        pinned = []
        for m in (n.kv_cache_block_manager, n.image_cache_block_manager):
            if m is not None:
                pinned.append(m.n_blocks - len(m.shared_cache.to_be_evicted)
                              - (1 if m is n.kv_cache_block_manager and n.node_type.enable_decode else 0))   # the decoder's pad block
        state = {"held": len(engine.held), "migrating": n.batch_scheduler.migrating_cnt, "dead": sorted(engine.dead),
                 "reaped": engine.n_reaped, "pinned": pinned, "n_base": len(base),
                 "max_tokens": {r.request_id: r.sampling_params.max_tokens for r in reqs}}
        q.put((rank, "ok", {k: {"tokens": v["tokens"], "path": v["path"], "failed": v.get("failed")} for k, v in mine.items()}, state))
        store = dist.distributed_c10d._get_default_store()
        store.add("kill/left", 1)
        t_end = time.monotonic() + 30
        while rank == 0 and store.add("kill/left", 0) < world - 1 and time.monotonic() < t_end:
            time.sleep(0.05)
    except Exception:  # pragma: no cover
        import traceback
        q.put((rank, traceback.format_exc(), None, None))
    q.close()
    q.join_thread()
    torch.cuda.synchronize()
    os._exit(0)


@pytest.mark.gpu
def test_a_prefill_rank_dies_mid_trace_on_the_gpu():
    roles, victim = ["E", "P", "P", "D"], 1
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_kill_worker, args=(r, roles, port, q, victim)) for r in range(len(roles))]
    for p in procs:
        p.start()
    results = {}
    try:
        for _ in range(len(roles) - 1):
            rank, status, mine, state = q.get(timeout=300)
            assert status == "ok", (rank, status[-2000:])
            results[rank] = (mine, state)
    finally:
        for p in procs:
            p.join(timeout=60)
            if p.is_alive():
                p.kill()
    assert procs[victim].exitcode == 17 and sorted(results) == [0, 2, 3]
    finished, failed = {}, {}
    for rank, (mine, state) in results.items():
        for rid, r in mine.items():
            assert rid not in finished and rid not in failed, f"request {rid} reported twice"
            (failed if r["failed"] else finished)[rid] = r
        assert state["dead"] == [victim] and state["held"] == 0 and state["migrating"] == 0 and not any(state["pinned"]), (rank, state)
    state = results[0][1]
    n_base, max_tokens = state["n_base"], state["max_tokens"]
    reaped = sum(st["reaped"] for _, st in results.values())
    assert len(finished) + len(failed) + reaped == len(max_tokens), (sorted(finished), sorted(failed), reaped)
    assert len(failed) + reaped > 0
    for rid, r in finished.items():
        assert len(r["tokens"]) == max_tokens[rid] and roles[r["path"][-1]] == "D"
    for rid in range(n_base, len(max_tokens)):      # entered after the death was noticed: over the surviving P rank, all finish
        assert rid in finished and victim not in finished[rid]["path"], (rid, finished.get(rid))
    assert any(victim in r["path"] for r in finished.values()), "nothing went through the victim before it died"
