"""One engine node per process (one process per GPU), E/P/D roles by rank
(parallel.epd_roles), migration by IPC peer reads — the multi-GPU form of engine.node.

What the reference does with Ray actor RPCs (epdnode.py:362-447: `migrate.remote`,
`pull_virtual_cache.remote`, `free_migrate_request.remote`) is done here with a mailbox on the
process group's key-value store (the TCPStore torch.distributed already runs): a sender numbers
and posts a pickled message, a receiver polls its own slots once per engine step.  Nodes are
never in lock-step — a D rank's 6 ms decode steps do not wait for a P rank's 40 ms prefill step.
Block tables and request state travel on the host; KV / image blocks never do — the receiver
reads them out of the sender's pool with hx_migrate_blocks (xGMI peer reads through the IPC
mapping).  The two store round trips per step are host time that the decode look-ahead hides."""
import dataclasses
import pickle
import time
from typing import Dict, List, Optional, Tuple

import torch
import torch.distributed as dist

from hydrainfer_amd.engine.isa import (EmptyInstruction, EPMigrate, Fill, ImageEmbedFill, InstructionListBuilder,
                                       PDMigrate, PullCache, TextFill)
from hydrainfer_amd.engine.node import EPDNode, NodeType
from hydrainfer_amd.engine.rcb import (RequestControlBlock, RequestMetaData, RequestMetric, SamplingParameters,
                                       ScenarioType)
from hydrainfer_amd.memory.token_cache import VirtualTokenCache


# ---------------------------------------------------------------- request state on the wire
def _cache_to_wire(vc: Optional[VirtualTokenCache]):
    return None if vc is None else dataclasses.asdict(vc)


def _cache_from_wire(d) -> Optional[VirtualTokenCache]:
    return None if d is None else VirtualTokenCache(**d)


def rcb_to_wire(rcb: RequestControlBlock) -> dict:
    """Everything the next stage needs: the instructions from the current one on (a flat list —
    the linked chain would pickle recursively), the block tables + IPC handles of the caches, the
    tokens so far and the timing stamps (CLOCK_MONOTONIC is shared by the processes of a node)."""
    insts = []
    inst = rcb.current_instruction()
    while inst is not None and inst.next is not None:          # stop at the tail sentinel
        if isinstance(inst, ImageEmbedFill):
            insts.append(("EF", inst.image_token_cache_ids, inst.image_token_mask, inst.token_ids,
                          inst.position_ids, inst.cache_ids, inst.sample, inst.hashes, inst.is_chunked))
        elif isinstance(inst, TextFill):
            insts.append(("TF", inst.token_ids, inst.position_ids, inst.cache_ids, inst.sample, inst.hashes,
                          inst.is_chunked))
        elif isinstance(inst, PullCache):
            insts.append(("PR", inst.hop))
        elif isinstance(inst, EPMigrate):
            insts.append(("EPMR",))
        elif isinstance(inst, PDMigrate):
            insts.append(("PDMR",))
        elif isinstance(inst, EmptyInstruction):
            insts.append(("EM",))
        else:
            raise RuntimeError(f"{inst!r} cannot migrate")
        inst = inst.next
    md = rcb.request_metadata
    return {"request_id": rcb.request_id, "instructions": insts,
            "sampling": (rcb.sampling_params.max_tokens, list(rcb.sampling_params.eos_token_ids)),
            "metadata": None if md is None else dataclasses.astuple(md),
            "kv": _cache_to_wire(rcb.virtual_kv_cache), "image": _cache_to_wire(rcb.virtual_image_cache),
            "output_token_ids": list(rcb.output_token_ids), "scenario": int(rcb.scenario_type or 0),
            "metric": dataclasses.asdict(rcb.metric), "stream_rank": rcb.stream_rank,
            "path": list(getattr(rcb, "path", []))}


def rcb_from_wire(w: dict) -> RequestControlBlock:
    rcb = RequestControlBlock()
    rcb.request_id = w["request_id"]
    rcb.sampling_params = SamplingParameters(w["sampling"][0], list(w["sampling"][1]))
    if w["metadata"] is not None:
        rcb.request_metadata = RequestMetaData(*w["metadata"])
    rcb.virtual_kv_cache, rcb.virtual_image_cache = _cache_from_wire(w["kv"]), _cache_from_wire(w["image"])
    rcb.output_token_ids = list(w["output_token_ids"])
    rcb.scenario_type = ScenarioType(w["scenario"])
    rcb.metric = RequestMetric(**w["metric"])
    rcb.stream_rank = w.get("stream_rank")
    rcb.path = list(w.get("path", []))       # the ranks that have owned this request, in order
    b = InstructionListBuilder()
    fills: List[Fill] = []
    for rec in w["instructions"]:
        kind = rec[0]
        if kind == "EF":
            inst = ImageEmbedFill(rec[1], rec[2], rec[3], rec[4], rec[5], rec[6], None, rec[7])
            inst.is_chunked = rec[8]
        elif kind == "TF":
            inst = TextFill(rec[1], rec[2], rec[3], rec[4], None, rec[5])
            inst.is_chunked = rec[6]
        elif kind == "PR":
            inst = PullCache()
            inst.hop = rec[1]
        else:
            inst = {"EPMR": EPMigrate, "PDMR": PDMigrate, "EM": EmptyInstruction}[kind]()
        if isinstance(inst, Fill):
            fills.append(inst)
        b.append(inst)
    for cur, nxt in zip(fills, fills[1:] + [None]):
        # a chunk head's sample is thrown away (isa.py); every other fill feeds the next one
        cur.sample_dst = EmptyInstruction() if cur.is_chunked else nxt
    rcb.instructions = b.build_instruction_list()
    return rcb


# ---------------------------------------------------------------- peers
class RemoteNode:
    """What an EPDNode sees of a node that lives in another process."""

    def __init__(self, rank: int, node_type: str, engine: "RankEngine", tpot_slo: float = 0.4):
        self.rank, self.node_type, self.engine, self.tpot_slo = rank, NodeType(node_type), engine, tpot_slo
        self.name = f"{node_type}@{rank}"

    # sender side, phase 1 (epdnode.py:412-441)
    def migrate(self, src_node: EPDNode, rcb: RequestControlBlock) -> None:
        self.engine.held[rcb.request_id] = rcb                   # blocks stay pinned until FREE
        self.engine.outbox.append((self.rank, "migrate", rcb_to_wire(rcb)))

    # receiver side, phase 3 of a send/recv transfer (epdnode.py:394-400 -> :362-378): the request goes
    # out AT ONCE, not with the step's outbox — the receiver is about to wait in recv for the data
    def pull_virtual_cache(self, which: str, src_cache: VirtualTokenCache, dst_cache: VirtualTokenCache) -> None:
        self.engine.mailbox.send(self.rank, "pull", (which, _cache_to_wire(src_cache), _cache_to_wire(dst_cache)))

    # receiver side, phase 4 (epdnode.py:443-446) — addressed to the sender
    def free_migrate_request(self, rcb: RequestControlBlock) -> None:
        self.engine.outbox.append((self.rank, "free", rcb.request_id))


class MailboxTokenProcessor:
    """OutputTokenProcessor of a request that is being served by another rank's front end: every sampled token goes to
    that rank's mailbox with the step's other messages (the reference: ZmqOutputTokenProcessor pushing to the API
    server's PULL socket, hydrainfer/engine/output_token_processor.py:92-140)."""

    def __init__(self, engine: "RankEngine", dst_rank: int, request_id):
        self.engine, self.dst_rank, self.request_id = engine, dst_rank, request_id

    def append_token_id(self, token_id: int, is_last_token: bool = False) -> None:
        self.engine.outbox.append((self.dst_rank, "token", (self.request_id, int(token_id), bool(is_last_token))))


class StoreMailbox:
    """Numbered per-destination slots on a torch.distributed store.  `epoch` separates runs."""

    def __init__(self, store, rank: int, epoch: str):
        self.store, self.rank, self.epoch = store, rank, epoch
        self.next_slot = 1

    def send(self, dst: int, kind: str, payload) -> None:
        n = self.store.add(f"{self.epoch}/n/{dst}", 1)
        self.store.set(f"{self.epoch}/m/{dst}/{n}", pickle.dumps((self.rank, kind, payload)))

    def poll(self) -> List[Tuple[int, str, object]]:
        out = []
        while self.store.check([f"{self.epoch}/m/{self.rank}/{self.next_slot}"]):
            out.append(pickle.loads(self.store.get(f"{self.epoch}/m/{self.rank}/{self.next_slot}")))
            self.next_slot += 1
        return out

    def add_finished(self, k: int) -> int:
        return self.store.add(f"{self.epoch}/finished", k)


class LocalMailbox:
    """world_size 1."""

    def __init__(self):
        self.finished = 0

    def send(self, dst, kind, payload):
        raise RuntimeError("a single node has nobody to write to")

    def poll(self):
        return []

    def add_finished(self, k: int) -> int:
        self.finished += k
        return self.finished


class RankEngine:
    """The node of this rank + its mailbox."""

    def __init__(self, rank: int, roles: List[str], node: EPDNode, group=None):
        self.rank, self.roles, self.node, self.group = rank, roles, node, group
        self.world = len(roles)
        self.outbox: List[Tuple[int, str, object]] = []
        self.held: Dict[object, RequestControlBlock] = {}
        self.peers = {r: (node if r == rank else RemoteNode(r, t, self)) for r, t in enumerate(roles)}
        nt = node.node_type
        p_nodes = [self.peers[r] for r, t in enumerate(roles) if "P" in t]
        d_nodes = [self.peers[r] for r, t in enumerate(roles) if "D" in t]
        node.connect(p_nodes if nt.enable_encode else [], d_nodes if nt.enable_prefill else [])
        self.mailbox = LocalMailbox()
        self.reported = self.n_exchanges = self.total_finished = 0
        # serving front end on this rank (entrypoint/api_server.py): request id -> the OutputTokenProcessor that streams it
        self.token_handlers: Dict[object, object] = {}
        self.creator = None          # InstructionCreator for requests submitted by another rank's front end
        self._n_submitted = [0, 0]   # front door: requests so far [text-only, with image] (one round robin per kind)

    # ---- serving: a front end on ONE rank, requests entering where the routing rule says, tokens coming back ----------
    def submit(self, request, processor, creator, request_index: int = 0) -> None:
        """Front-end side: start `request` on the rank cluster.py:178-184 picks (image requests round-robin over the E
        ranks, text-only ones over the P ranks — two balancers, each with its own cursor); its tokens — sampled on
        whichever ranks run its prefill and decode — are delivered to `processor` on THIS rank."""
        self.token_handlers[request.request_id] = processor
        has_image = request.pixel_values is not None
        dst = entry_rank(self._n_submitted[has_image], self.roles, has_image)
        self._n_submitted[has_image] += 1
        if dst == self.rank:
            try:
                self._start(request, creator, self.rank)
            except Exception:
                self.token_handlers.pop(request.request_id, None)
                raise
        else:
            self.outbox.append((dst, "submit", (request, self.rank)))

    def _start(self, request, creator, stream_rank: int) -> None:
        rcb = creator.process(request)
        rcb.stream_rank = stream_rank
        rcb.path = [self.rank]
        self._attach_stream(rcb)
        self.node.add_request(rcb)

    def _attach_stream(self, rcb: RequestControlBlock) -> None:
        if rcb.stream_rank is None:
            return
        if rcb.stream_rank == self.rank:
            h = self.token_handlers.get(rcb.request_id)
            if h is not None:
                rcb.register_output_token_processor(_LocalStream(self, rcb.request_id))
        else:
            rcb.register_output_token_processor(MailboxTokenProcessor(self, rcb.stream_rank, rcb.request_id))

    def _token(self, request_id, token: int, last: bool) -> None:
        h = self.token_handlers.get(request_id)
        if h is not None:
            h.append_token_id(token, last)
            if last:
                del self.token_handlers[request_id]

    def connect_transfer_peers(self, timeout_s: Optional[float] = None) -> int:
        """Call on every rank once at start-up, before the first request: the send/recv communicators of every E -> P
        (image blocks) and P -> D (KV blocks) hop this rank takes part in are created under a bound
        (CommunicationBackendManager.connect_peers).  Nothing to do for the IPC pull backend — the intra-node default."""
        n = 0
        ep = [(e, p_) for e, te in enumerate(self.roles) if "E" in te for p_, tp in enumerate(self.roles) if "P" in tp]
        pd = [(p_, d) for p_, tp in enumerate(self.roles) if "P" in tp for d, td in enumerate(self.roles) if "D" in td]
        for manager, pairs in ((self.node.image_cache_block_manager, ep), (self.node.kv_cache_block_manager, pd)):
            mm = getattr(manager, "migrate_manager", None)
            if mm is not None and self.world > 1:
                n += mm.connect_peers(self.rank, pairs, timeout_s)
        return n

    def open_mailbox(self, epoch: str) -> None:
        """Call on every rank before a run (same epoch everywhere)."""
        if self.world > 1:
            self.mailbox = StoreMailbox(dist.distributed_c10d._get_default_store(), self.rank, epoch)
        else:
            self.mailbox = LocalMailbox()
        self.reported = len(self.node.finished)
        self.n_exchanges = self.total_finished = 0

    def _deliver(self, src_rank: int, kind: str, payload) -> None:
        if kind == "migrate":
            rcb = rcb_from_wire(payload)
            rcb.path.append(self.rank)
            self._attach_stream(rcb)
            self.node.migrate(self.peers[src_rank], rcb)
        elif kind == "token":
            self._token(*payload)
        elif kind == "submit":
            request, stream_rank = payload
            if self.creator is None:
                raise RuntimeError("a front end submitted a request to this rank, but RankEngine.creator is not set")
            try:
                self._start(request, self.creator, stream_rank)
            except Exception as e:      # e.g. prompt + max_tokens past the rotary table: the stream must end, loudly
                self.outbox.append((stream_rank, "failed", (request.request_id, repr(e))))
        elif kind == "failed":
            h = self.token_handlers.pop(payload[0], None)
            if h is not None and hasattr(h, "fail"):
                h.fail(RuntimeError(payload[1]))
        elif kind == "pull":
            which, src, dst = payload
            self.node.pull_virtual_cache(which, _cache_from_wire(src), _cache_from_wire(dst))
        elif kind == "free":
            rcb = self.held.pop(payload)
            self.node.free_migrate_request(rcb)
            rcb.release_instructions()          # this process's copy is dead (the receiver rebuilt its own)
        else:
            raise RuntimeError(kind)

    def exchange(self) -> int:
        """Post what this step produced, take what has arrived; returns the number of requests
        finished cluster-wide in this run."""
        for dst, kind, payload in self.outbox:
            self.mailbox.send(dst, kind, payload)
        self.outbox = []
        for src, kind, payload in self.mailbox.poll():
            self._deliver(src, kind, payload)
        done = len(self.node.finished) - self.reported
        self.reported += done
        self.n_exchanges += 1
        if done or self.n_exchanges % 8 == 0:          # the end-of-run test is not latency critical
            self.total_finished = self.mailbox.add_finished(done)
        return self.total_finished

    def step(self) -> int:
        self.node.step()
        return self.exchange()


class _LocalStream:
    """Tokens sampled on the front end's own rank."""

    def __init__(self, engine: RankEngine, request_id):
        self.engine, self.request_id = engine, request_id

    def append_token_id(self, token_id: int, is_last_token: bool = False) -> None:
        self.engine._token(self.request_id, int(token_id), bool(is_last_token))


def entry_rank(kind_ordinal: int, roles: List[str], has_image: bool) -> int:
    """cluster.py:178-184: image requests round-robin over the E nodes (`ebalancer`), text-only ones over the P nodes
    (`pbalancer`) — two independent cursors, so `kind_ordinal` counts the requests OF THE SAME KIND seen so far."""
    ranks = [r for r, t in enumerate(roles) if ("E" if has_image else "P") in t]
    return ranks[kind_ordinal % len(ranks)]


def entry_ranks(requests, roles: List[str]) -> List[int]:
    """The front door applied to a whole trace in arrival (list) order."""
    seen, out = [0, 0], []
    for r in requests:
        has_image = r.pixel_values is not None
        out.append(entry_rank(seen[has_image], roles, has_image))
        seen[has_image] += 1
    return out


def replay_distributed(engine: RankEngine, creator, requests, arrivals: List[float], t0: float,
                       device: Optional[torch.device] = None, deadline_s: float = 600.0) -> dict:
    """Every rank runs this with the same request list; a request enters at the rank
    `entry_rank` names.  Returns this rank's finished requests' metrics."""
    door = entry_ranks(requests, engine.roles)
    mine = sorted((i for i in range(len(requests)) if door[i] == engine.rank), key=lambda i: arrivals[i])
    from hydrainfer_amd.engine.serve import ADMIT_PER_STEP, quiet_gc
    nxt, total = 0, len(requests)
    first_finished = len(engine.node.finished)
    with quiet_gc():
        while True:
            now = time.perf_counter() - t0
            admitted = 0
            while nxt < len(mine) and arrivals[mine[nxt]] <= now and admitted < ADMIT_PER_STEP:
                admitted += 1
                i = mine[nxt]
                rcb = creator.process(requests[i])
                rcb.path = [engine.rank]
                engine.node.add_request(rcb)
                rcb.metric.arrival_time = t0 + arrivals[i]
                nxt += 1
            if engine.step() >= total:
                break
            if now > deadline_s:
                raise TimeoutError(f"rank {engine.rank}: trace not drained after {deadline_s} s")
            if engine.node.idle():
                time.sleep(0.001)       # an idle rank polls its mailbox about 1000 times a second
    t_wait = time.perf_counter()
    while engine.held and time.perf_counter() - t_wait < 10.0:     # FREEs still on their way
        engine.exchange()
        time.sleep(0.0005)
    if device is not None and device.type == "cuda":
        torch.cuda.synchronize(device)
    return {r.request_id: {"arrival": r.metric.arrival_time, "token_times": list(r.metric.token_times),
                           "tokens": list(r.output_token_ids), "ep_transfer": list(r.metric.ep_transfer),
                           "pd_transfer": list(r.metric.pd_transfer), "path": list(getattr(r, "path", []))}
            for r in engine.node.finished[first_finished:]}


def summarize(per_request: Dict[int, dict], t0: float) -> dict:
    rs = list(per_request.values())
    end = max(r["token_times"][-1] for r in rs)
    n_out = sum(len(r["tokens"]) for r in rs)
    ttft = sorted(r["token_times"][0] - r["arrival"] for r in rs)
    tpot = sorted((r["token_times"][-1] - r["token_times"][0]) / max(1, len(r["token_times"]) - 1) for r in rs)
    hop = lambda key: sorted(r[key][1] - r[key][0] for r in rs if len(r[key]) == 2)
    pct = lambda xs, p: xs[min(len(xs) - 1, int(p * len(xs)))] if xs else None
    ms = lambda v: None if v is None else round(v * 1e3, 3)
    pairs: Dict[str, int] = {}
    for r in rs:            # hand-overs per (sender rank -> receiver rank) pair: the many-to-many routing at a glance
        for a, b in zip(r.get("path", []), r.get("path", [])[1:]):
            pairs[f"{a}->{b}"] = pairs.get(f"{a}->{b}", 0) + 1
    return {"requests": len(rs), "output_tokens": n_out, "wall_s": round(end - t0, 3), "pulls_per_pair": pairs,
            "output_tok_s": round(n_out / (end - t0), 1),
            "ttft_mean_ms": ms(sum(ttft) / len(ttft)),
            "ttft_p50_ms": ms(pct(ttft, 0.5)), "ttft_p99_ms": ms(pct(ttft, 0.99)),
            "tpot_p50_ms": ms(pct(tpot, 0.5)), "tpot_p99_ms": ms(pct(tpot, 0.99)),
            "ep_pull_p50_ms": ms(pct(hop("ep_transfer"), 0.5)), "pd_pull_p50_ms": ms(pct(hop("pd_transfer"), 0.5))}
