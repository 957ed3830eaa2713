"""One engine node per process over gloo (engine/distributed.py): request state on the wire and
the migrate / pull / free protocol between E, P and D ranks.  CPU only: the models are the
deterministic stand-in of tests.golden.cases.engine_trace_sample, so every request's tokens are
known in closed form whatever the interleaving; the pools are CPU tensors whose cross-process
pull is a no-op (the GPU pull is covered by tests/test_gpu_migration.py and the single-GPU
multi-rank run of tests/test_gpu_distributed_engine.py)."""
import os
import socket
from types import SimpleNamespace as NS

import pytest
import torch
import torch.multiprocessing as mp

from tests.golden import cases as C

N_IMG, BS, IMAGE_TOKEN = 40, 16, 32000


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _requests(n=14):
    from hydrainfer_amd.engine import SamplingParameters, TokenRequest
    g = torch.Generator().manual_seed(31)
    reqs = []
    for i in range(n):
        text = torch.randint(1000, 31999, (5 + (i * 7) % 40,), generator=g).tolist()
        has_image = i % 5 != 4
        reqs.append(TokenRequest(i, ([IMAGE_TOKEN] if has_image else []) + text,
                                 torch.full((1, 3, 2, 2), float(i)) if has_image else None, (8, 8), 4000 + i,
                                 SamplingParameters(max_tokens=2 + (i * 3) % 9)))
    return reqs


def expected_tokens(req):
    """Closed form of the stand-in model: the prompt's last token samples f(id, pos), and so on."""
    ids = []
    for t in req.token_ids:
        ids += [IMAGE_TOKEN] * N_IMG if t == IMAGE_TOKEN else [t]
    out, last, pos = [], ids[-1], len(ids) - 1
    for _ in range(req.sampling_params.max_tokens):
        last = C.engine_trace_sample(last, pos)
        out.append(last)
        pos += 1
    return out


def _build_engine(rank, roles, group, sendrecv=False):
    from hydrainfer_amd.engine import BatchSchedulerConfig
    from hydrainfer_amd.engine.distributed import RankEngine
    from tests.engine_util import CpuPoolManager, make_node

    class Pool(CpuPoolManager):
        """CPU stand-in of a cache pool.  sendrecv=False: the IPC pull (one-sided, a no-op here).
        sendrecv=True: the send/recv transfer of memory/communication.py::RCCLBackend — the sender's
        half posts the blocks with dist.send, the receiver's half takes them with dist.recv, over
        gloo: a pull whose sender half is never run hangs this test."""

        def needs_sender(self, src, dst):
            return sendrecv

        def migrate_blocks(self, src, dst, is_send=False):
            import torch.distributed as dist
            assert len(src.block_table) == len(dst.block_table) and src.memory_handle
            if sendrecv:
                if is_send:
                    assert src.rank == rank and dst.rank != rank
                    dist.send(self.cache_tensor[:, :, src.block_table].contiguous(), dst=dst.rank)
                    self.sent = getattr(self, "sent", 0) + len(src.block_table)
                    return
                assert dst.rank == rank and src.rank != rank
                buf = torch.empty_like(self.cache_tensor[:, :, dst.block_table])
                dist.recv(buf, src=src.rank)
                self.cache_tensor[:, :, dst.block_table] = buf
            else:
                assert not is_send, "the IPC pull has no sender half"
            self.pulled = getattr(self, "pulled", 0) + len(src.block_table)

    shape = NS(num_hidden_layers=1, num_attention_heads=1, num_key_value_heads=1, head_dim=8)

    class LM:
        image_token_id = IMAGE_TOKEN
        language_model = NS(shape=shape)

        def forward(self, ids, feats, pos, params):
            i, p = ids.tolist(), pos.tolist()
            return torch.tensor([C.engine_trace_sample(i[j], p[j]) for j in params.selected_token_ids.tolist()])

    class Vision:
        def forward(self, px):
            return torch.zeros(px.shape[0], N_IMG, 8)

    kv, img = Pool(1, 2, 64, BS, 1, 8), Pool(1, 1, 10, N_IMG, 1, 8)
    kv.rank = img.rank = rank          # virtual caches carry the rank that owns their blocks
    cfg = BatchSchedulerConfig(max_running_requests=4, token_budgets=64, image_budgets=2)
    node = make_node(f"{roles[rank]}{rank}", roles[rank], LM(), Vision(), kv, img, shape, torch.float32,
                     torch.device("cpu"), cfg)
    return RankEngine(rank, roles, node, group), kv, img


def _worker(rank, roles, port, q, sendrecv=False):
    try:
        import time
        import torch.distributed as dist
        from hydrainfer_amd.engine import InstructionCreator
        from hydrainfer_amd.engine.distributed import replay_distributed, summarize
        world = len(roles)
        if world > 1:
            dist.init_process_group("gloo", rank=rank, world_size=world, init_method=f"tcp://127.0.0.1:{port}")
        engine, kv, img = _build_engine(rank, roles, None, sendrecv)
        reqs = _requests(14 if world < 8 else 40)
        arrivals = [0.002 * i for i in range(len(reqs))]
        box = [time.perf_counter() + 0.05]
        if world > 1:
            dist.broadcast_object_list(box, src=0)
        creator = InstructionCreator(IMAGE_TOKEN, N_IMG, BS)
        engine.open_mailbox("run0")
        if world > 1:
            dist.barrier()
        mine = replay_distributed(engine, creator, reqs, arrivals, box[0], deadline_s=120)
        # every block of every pool is free again and nothing is waiting for a FREE
        for m in (engine.node.kv_cache_block_manager, engine.node.image_cache_block_manager):
            if m is not None:
                assert len(m.shared_cache.to_be_evicted) == m.n_blocks
        assert not engine.held and engine.node.batch_scheduler.migrating_cnt == 0
        allr = [None] * world
        if world > 1:
            dist.all_gather_object(allr, (mine, getattr(kv, "pulled", 0), getattr(img, "pulled", 0),
                                          getattr(kv, "sent", 0) + getattr(img, "sent", 0)))
            dist.barrier()
            dist.destroy_process_group()
        else:
            allr = [(mine, 0, 0, 0)]
        if rank == 0:
            merged = {}
            for m, _, _, _ in allr:
                merged.update(m)
            assert sorted(merged) == list(range(len(reqs)))
            for i, r in enumerate(reqs):
                assert merged[i]["tokens"] == expected_tokens(r), f"request {i}"
            s = summarize(merged, box[0])
            assert s["requests"] == len(reqs) and s["output_tokens"] == sum(r.sampling_params.max_tokens for r in reqs)
            d_ranks = [r for r, t in enumerate(roles) if "D" in t]
            for r in range(world):           # requests finish on D ranks only
                assert (len(allr[r][0]) > 0) == (r in d_ranks), f"rank {r} finished {len(allr[r][0])}"
            _check_routing(roles, reqs, merged, {r: allr[r][0] for r in range(world)})
            if world > 1:
                assert sum(a[1] for a in allr) > 0          # KV blocks were pulled P -> D
                if "E" in roles:
                    assert sum(a[2] for a in allr) > 0      # image blocks were pulled E -> P
                if sendrecv:                                # every pulled block was sent by its owner's half
                    assert sum(a[3] for a in allr) == sum(a[1] + a[2] for a in allr) > 0
        q.put((rank, "ok"))
    except Exception:  # pragma: no cover
        import traceback
        q.put((rank, traceback.format_exc()))


def _check_routing(roles, reqs, merged, finished_on):
    """What only a many-to-many topology shows (BASELINE configs[4] = 2E + 2P + 4D): the front door's two round robins
    (hydrainfer/cluster/cluster.py:178-184), every hop's round robin over ALL its downstream nodes
    (cluster/epdnode.py:56-75,419-420), every D rank finishing requests pulled from EVERY P pool."""
    from collections import Counter
    e = [r for r, t in enumerate(roles) if "E" in t]
    p_ = [r for r, t in enumerate(roles) if "P" in t]
    d = [r for r, t in enumerate(roles) if "D" in t]
    seen = [0, 0]
    for i, r in enumerate(reqs):
        has_image = r.pixel_values is not None
        door = (e if has_image else p_)[seen[has_image] % len(e if has_image else p_)]
        seen[has_image] += 1
        path = merged[i]["path"]
        assert path[0] == door, f"request {i}: entered at rank {path[0]}, the front door says {door}"
        # E -> P -> D, a node never hands a request to itself, and the last owner is a D rank that finished it
        want = ["E"] * has_image + ["P", "D"]
        stages, k = [], 0
        for rank in path:
            while k < len(want) and want[k] in roles[rank]:
                k += 1
            stages.append(k)
        assert k == len(want) and all(a != b for a, b in zip(path, path[1:])), (i, path)
        assert i in finished_on[path[-1]], (i, path)
    hops = Counter((a, b) for m in merged.values() for a, b in zip(m["path"], m["path"][1:]))
    for senders, receivers in ((e, p_), (p_, d)):
        for s_ in senders:
            counts = [hops[(s_, r)] for r in receivers if r != s_]
            if "EPD" == roles[s_] or not counts or sum(counts) == 0:
                continue
            # a round robin over the downstream nodes: every one of them is visited, evenly
            assert min(counts) > 0 and max(counts) - min(counts) <= 1, (s_, receivers, counts)
    if len(p_) > 1 and len(d) > 1:
        for rank in d:              # a D rank holds KV pulled from BOTH P pools
            assert {m["path"][-2] for i, m in merged.items() if m["path"][-1] == rank} == set(p_), rank


def test_rcb_wire_round_trip():
    from hydrainfer_amd.engine import InstructionCreator
    from hydrainfer_amd.engine.distributed import rcb_from_wire, rcb_to_wire
    import pickle
    creator = InstructionCreator(IMAGE_TOKEN, N_IMG, BS)
    rcb = creator.process(_requests()[0])
    first = rcb.current_instruction()
    rcb.step(); rcb.step()                      # ImageEmbed, EPMigrate done -> at PullCache
    fill = rcb.current_instruction().next
    fill.chunk_prefill(17)                      # a chunked prefill must survive the trip
    rcb.output_token_ids = [5]
    back = rcb_from_wire(pickle.loads(pickle.dumps(rcb_to_wire(rcb))))
    a, b = list(rcb.instructions)[3:], list(back.instructions)[1:]   # from the current instruction on
    assert [repr(x) for x in a] == [repr(x) for x in b]
    for x, y in zip(a, b):
        for f in ("token_ids", "position_ids", "cache_ids", "sample", "hashes", "is_chunked",
                  "image_token_cache_ids", "image_token_mask"):
            assert getattr(x, f, None) == getattr(y, f, None), f
    fills = [x for x in b if hasattr(x, "token_ids")]
    assert repr(fills[0].sample_dst) == "EM" and fills[1].sample_dst is fills[2] and fills[-1].sample_dst is None
    assert repr(back.current_instruction()) == "PR" and back.output_token_ids == [5]
    assert back.sampling_params.max_tokens == rcb.sampling_params.max_tokens and first is not None


HYBRID_POOL = ["E", "E", "P", "P", "D", "D", "D", "D"]          # BASELINE configs[4], hydrainfer/config/cluster/hybrid.yaml


@pytest.mark.parametrize("roles", [["EPD"], ["EP", "D"], ["E", "P", "D"], ["E", "P", "D", "D"], HYBRID_POOL], ids="-".join)
def test_distributed_engine_protocol(roles):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, roles, port, q)) for r in range(len(roles))]
    for p in procs:
        p.start()
    results = [q.get(timeout=240) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    assert sorted(results) == [(r, "ok") for r in range(len(roles))], results


@pytest.mark.parametrize("roles", [["EP", "D"], ["E", "P", "D"], ["E", "P", "D", "D"], HYBRID_POOL], ids="-".join)
def test_distributed_engine_send_recv_pull(roles):
    """The transfer path of ranks that cannot map each other's pool (different hosts, or
    intranode_migrate_backend='nccl'): the receiver asks the sender for its half of the send/recv
    pair before waiting in recv (hydrainfer/cluster/epdnode.py:362-378,394-400)."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, roles, port, q, True)) for r in range(len(roles))]
    for p in procs:
        p.start()
    results = [q.get(timeout=240) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    assert sorted(results) == [(r, "ok") for r in range(len(roles))], results


# ---------------------------------------------------------------------------------------------------------------------
# A rank dies in the middle of a trace (hydrainfer/cluster/epdnode.py:428-442: a hand-over that fails ends THAT request —
# blocks freed on both sides, a None token to its stream — never the node that noticed).
# ---------------------------------------------------------------------------------------------------------------------
N_EARLY, N_LATE, T_LATE = 14, 6, 3.0


def _kill_worker(rank, roles, port, q, victim, die_after):
    try:
        import time
        os.environ["HX_PEER_DEAD_AFTER_S"] = "1.0"
        import torch.distributed as dist
        from hydrainfer_amd.engine import InstructionCreator
        from hydrainfer_amd.engine.distributed import RankEngine, replay_distributed
        world = len(roles)
        dist.init_process_group("gloo", rank=rank, world_size=world, init_method=f"tcp://127.0.0.1:{port}")
        engine, kv, img = _build_engine(rank, roles, None)
        if rank == victim:
            real_step = engine.step

            def step():
                # dies once it has handed `die_after` requests on to D ranks while others are still queued or being pulled
                handed = sum(1 for r in engine.node.finished) + len(engine.held) + getattr(engine, "n_freed", 0)
                s_ = engine.node.batch_scheduler
                if handed >= die_after and (s_.waiting or s_.running):
                    os._exit(17)              # a fresh child process that never touched a GPU: it just stops
                return real_step()
            real_deliver = engine._deliver

            def deliver(src, kind, payload):
                if kind == "free":
                    engine.n_freed = getattr(engine, "n_freed", 0) + 1
                real_deliver(src, kind, payload)
            engine._deliver, engine.step = deliver, step
        reqs = _requests(N_EARLY + N_LATE)
        arrivals = [0.002 * i for i in range(N_EARLY)] + [T_LATE + 0.002 * i for i in range(N_LATE)]
        box = [time.perf_counter() + 0.05]
        dist.broadcast_object_list(box, src=0)
        engine.open_mailbox("kill")
        dist.barrier()
        try:
            mine = replay_distributed(engine, InstructionCreator(IMAGE_TOKEN, N_IMG, BS), reqs, arrivals, box[0], deadline_s=40)
        except TimeoutError as e:
            n = engine.node
            raise TimeoutError(f"{e}: finished {[r.request_id for r in n.finished]} failed {[(r.request_id, r.failed) for r in n.failed]} "
                               f"held {list(engine.held)} dead {engine.dead} reaped {engine.reaped} n_reaped {engine.n_reaped} total "
                               f"{engine.total_finished} waiting {[r.request_id for r in n.batch_scheduler.waiting]} running "
                               f"{[(r.request_id, repr(r.current_instruction())) for r in n.batch_scheduler.running]} registered "
                               f"{[(rid, engine.mailbox.owner_of(rid, er)) for rid, _, er in engine.mailbox.registered()]}")
        sch = engine.node.batch_scheduler
        state = {"queued": [(r.request_id, repr(r.current_instruction()), r.path) for r in list(sch.waiting) + list(sch.running)],
                 "held": len(engine.held), "migrating": engine.node.batch_scheduler.migrating_cnt, "dead": sorted(engine.dead),
                 "reaped": engine.n_reaped, "pinned": [m.n_blocks - len(m.shared_cache.to_be_evicted)
                                                       for m in (engine.node.kv_cache_block_manager, engine.node.image_cache_block_manager)
                                                       if m is not None]}
        q.put((rank, "ok", mine, state))
        # rank 0 hosts the store every rank's mailbox lives on: it leaves last
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
    os._exit(0)          # (no collective teardown: one rank of the group is gone)


def test_a_rank_dies_mid_trace_and_only_its_requests_end():
    roles, victim = ["E", "P", "P", "D", "D"], 1
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_kill_worker, args=(r, roles, port, q, victim, 3)) for r in range(len(roles))]
    for p in procs:
        p.start()
    results = {}
    for _ in range(len(roles) - 1):
        rank, status, mine, state = q.get(timeout=240)
        assert status == "ok", (rank, status)
        results[rank] = (mine, state)
    for p in procs:
        p.join(timeout=60)
    assert procs[victim].exitcode == 17 and sorted(results) == [0, 2, 3, 4]
    reqs = _requests(N_EARLY + N_LATE)
    finished, failed = {}, {}
    for rank, (mine, state) in results.items():
        for rid, r in mine.items():
            assert rid not in finished and rid not in failed, f"request {rid} reported twice"
            (failed if r.get("failed") else finished)[rid] = r
        # every survivor noticed, holds nothing for the dead rank, and has all its blocks back
        assert state["dead"] == [victim] and state["held"] == 0 and state["migrating"] == 0 and not any(state["pinned"]), \
            (rank, str(state), sorted(mine), [(k, v.get("failed"), v["path"]) for k, v in mine.items()])
    reaped = sum(state["reaped"] for _, state in results.values())
    # every request ended exactly one way: finished, terminated by the rank that held it, or written off with the dead rank
    assert len(finished) + len(failed) + reaped == len(reqs), (sorted(finished), sorted(failed), reaped)
    assert len(failed) + reaped > 0, "the death cost nothing?"
    for rid, r in finished.items():
        assert r["tokens"] == expected_tokens(reqs[rid]), rid
        assert roles[r["path"][-1]] == "D"
    for rid, r in failed.items():
        assert r["tokens"] != expected_tokens(reqs[rid]) and "died" in r["failed"], (rid, r["failed"])
    # requests that entered after the death was noticed run over the surviving P rank only, and all of them finish
    for rid in range(N_EARLY, N_EARLY + N_LATE):
        assert rid in finished and victim not in finished[rid]["path"], (rid, finished.get(rid))
    assert any(victim in r["path"] for r in finished.values()), "nothing went through the victim before it died"
