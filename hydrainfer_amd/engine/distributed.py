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
import os
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
        self.engine.held_dst[rcb.request_id] = self.rank
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

    def fail(self, exc: BaseException) -> None:
        """The engine terminated the request (EPDNode.terminate): the front end's stream must end — the reference
        pushes `(request_id, None)` over the same zmq socket (hydrainfer/cluster/epdnode.py:440-442)."""
        self.engine.outbox.append((self.dst_rank, "failed", (self.request_id, str(exc)[:300])))


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

    def claim(self, what: str) -> bool:
        """True for exactly one caller per `what` (cluster-wide)."""
        return self.store.add(f"{self.epoch}/claim/{what}", 1) == 1

    # ---- liveness: a counter per rank, bumped by its step loop; `dead/<r>` once a peer has given up on r
    def beat(self) -> None:
        self.store.add(f"{self.epoch}/hb/{self.rank}", 1)

    def beats_of(self, rank: int) -> int:
        return self.store.add(f"{self.epoch}/hb/{rank}", 0)

    def mark_dead(self, rank: int) -> None:
        self.store.set(f"{self.epoch}/dead/{rank}", b"1")

    def marked_dead(self, rank: int) -> bool:
        return self.store.check([f"{self.epoch}/dead/{rank}"])

    def ack_dead(self, rank: int) -> None:
        """This rank has taken everything `rank` had posted before it died."""
        self.store.set(f"{self.epoch}/ack/{rank}/{self.rank}", b"1")

    def acked_dead(self, rank: int, by: int) -> bool:
        return self.store.check([f"{self.epoch}/ack/{rank}/{by}"])

    # ---- who holds a request: the registry every rank can read when a rank has died
    def register(self, request_id, stream_rank) -> None:
        """At the front door: the request exists and `self.rank` owns it."""
        n = self.store.add(f"{self.epoch}/nreq", 1)
        self.store.set(f"{self.epoch}/req/{n}", pickle.dumps((request_id, stream_rank, self.rank)))

    def set_owner(self, request_id, owner: int) -> None:
        """owner = a rank, or -1 once the request has finished or been terminated."""
        self.store.set(f"{self.epoch}/own/{request_id}", str(owner).encode())

    def owner_of(self, request_id, entry_rank: int) -> int:
        key = f"{self.epoch}/own/{request_id}"
        return int(self.store.get(key)) if self.store.check([key]) else entry_rank

    def registered(self):
        """(request id, stream rank, entry rank) of every request that has entered, in entry order."""
        n = self.store.add(f"{self.epoch}/nreq", 0)
        return [pickle.loads(self.store.get(f"{self.epoch}/req/{i}")) for i in range(1, n + 1)]


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

    def beat(self):
        pass

    def claim(self, what):
        return True

    def register(self, request_id, stream_rank):
        pass

    def set_owner(self, request_id, owner):
        pass


class RankDeclaredDead(RuntimeError):
    """The other ranks have given up on this one (no heartbeat for dead_after_s): its requests have been terminated and
    its blocks written off — it must not go on."""


class RankEngine:
    """The node of this rank + its mailbox."""

    def __init__(self, rank: int, roles: List[str], node: EPDNode, group=None):
        self.rank, self.roles, self.node, self.group = rank, roles, node, group
        self.world = len(roles)
        self.outbox: List[Tuple[int, str, object]] = []
        self.held: Dict[object, RequestControlBlock] = {}
        self.held_dst: Dict[object, int] = {}          # request id -> the rank it has been handed to (until its FREE)
        # FAILURE SEMANTICS (hydrainfer/cluster/epdnode.py:428-442: a hand-over that fails ends THAT request — blocks freed,
        # a None token to its stream — never the node).  Every rank bumps a heartbeat on the store; a rank silent for
        # dead_after_s is declared dead by whoever notices: its peers stop routing to it, end the requests they had handed
        # to it or were about to pull from it, and the lowest live rank ends the ones that died WITH it (the store's
        # ownership registry says which) — the run's accounting closes and every stream ends.
        self.dead: set = set()
        self.reaped: set = set()
        self.dead_after_s = float(os.environ.get("HX_PEER_DEAD_AFTER_S", "10"))
        self._beat_at = self._checked_at = 0.0
        self._seen: Dict[int, Tuple[int, float]] = {}
        self.n_reaped = 0
        node.peer_alive = lambda peer: not (isinstance(peer, RemoteNode) and peer.rank in self.dead)
        self.peers = {r: (node if r == rank else RemoteNode(r, t, self)) for r, t in enumerate(roles)}
        nt = node.node_type
        p_nodes = [self.peers[r] for r, t in enumerate(roles) if "P" in t]
        d_nodes = [self.peers[r] for r, t in enumerate(roles) if "D" in t]
        node.connect(p_nodes if nt.enable_encode else [], d_nodes if nt.enable_prefill else [])
        self.mailbox = LocalMailbox()
        self.reported = self.reported_failed = self.n_exchanges = self.total_finished = 0
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
        dst = entry_rank(self._n_submitted[has_image], self.roles, has_image, self.dead)
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
        self._attach_stream(rcb)
        self.admit(rcb)

    def admit(self, rcb: RequestControlBlock) -> None:
        """A request enters the cluster at this rank."""
        rcb.path = [self.rank]
        self.mailbox.register(rcb.request_id, rcb.stream_rank)
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
        self.reported, self.reported_failed = len(self.node.finished), len(self.node.failed)
        self.n_exchanges = self.total_finished = 0
        self.dead, self.reaped, self._seen = set(), set(), {}
        self._beat_at = self._checked_at = 0.0
        self.mailbox.beat()

    def _deliver(self, src_rank: int, kind: str, payload) -> None:
        if kind == "migrate":
            rcb = rcb_from_wire(payload)
            rcb.path.append(self.rank)
            self._attach_stream(rcb)
            self.mailbox.set_owner(rcb.request_id, self.rank)      # from here on its fate is this rank's to report
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
            rcb = self.held.pop(payload, None)
            self.held_dst.pop(payload, None)
            if rcb is None:          # written off already (the receiver had been declared dead, or pulled twice)
                return
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
        self._check_peers()
        for rcb in self.node.finished[self.reported:] + self.node.failed[self.reported_failed:]:
            self.mailbox.set_owner(rcb.request_id, -1)
        done = (len(self.node.finished) - self.reported) + (len(self.node.failed) - self.reported_failed)
        self.reported, self.reported_failed = len(self.node.finished), len(self.node.failed)
        self.n_exchanges += 1
        if done or self.n_exchanges % 8 == 0:          # the end-of-run test is not latency critical
            self.total_finished = self.mailbox.add_finished(done)
        return self.total_finished

    # ---- liveness -------------------------------------------------------------------------------------------------
    def _check_peers(self) -> None:
        if self.world == 1 or not isinstance(self.mailbox, StoreMailbox):
            return
        now = time.monotonic()
        period = min(0.1, self.dead_after_s / 8)
        if now - self._beat_at >= period:
            self._beat_at = now
            self.mailbox.beat()
        if now - self._checked_at < 2 * period:
            return
        self._checked_at = now
        if self.mailbox.marked_dead(self.rank):
            raise RankDeclaredDead(f"rank {self.rank}: declared dead by its peers (no heartbeat for {self.dead_after_s:g} s)")
        for r in range(self.world):
            if r == self.rank or r in self.dead:
                continue
            beats = self.mailbox.beats_of(r)
            last = self._seen.get(r)
            if last is None or last[0] != beats:
                self._seen[r] = (beats, now)
            elif beats > 0 and now - last[1] > self.dead_after_s:      # (a rank that has not started its loop yet is not dead)
                self._on_dead(r)
        self._reap()

    def _on_dead(self, x: int) -> None:
        """Rank x has stopped: what THIS rank owes the requests that involved it."""
        self.dead.add(x)
        self.mailbox.mark_dead(x)
        peer = self.peers[x]
        self.node.ep_loadbalancer.remove_worker(peer)      # hand-overs go on round-robin over the nodes that are left
        self.node.pd_loadbalancer.remove_worker(peer)
        for src, kind, payload in self.mailbox.poll():     # everything x posted before it died is in the slots by now
            self._deliver(src, kind, payload)
        for rid, dst in list(self.held_dst.items()):       # handed to x, never freed by it
            if dst != x:
                continue
            rcb = self.held.pop(rid)
            del self.held_dst[rid]
            self.node.batch_scheduler.migrating_release()
            if self.mailbox.owner_of(rid, self.rank) == x:
                # x had taken it over and died with it: the reaper ends it; only the blocks come back here
                self.node._free_cache(rcb)
                rcb.release_instructions()
            else:
                self.node.terminate(rcb, f"rank {x} ({self.roles[x]}) died before it took the request over")
        # (requests queued here to PULL from x end when their PullCache comes up: EPDNode._execute_pull_cache, peer_alive)
        self.mailbox.ack_dead(x)

    def _reap(self) -> None:
        """The lowest live rank ends the requests that died with a dead rank — once every live rank has taken what that
        rank had posted (a request handed over just before the death belongs to its receiver by then)."""
        live = [r for r in range(self.world) if r not in self.dead]
        if not live or live[0] != self.rank:
            return
        for x in sorted(self.dead - self.reaped):
            if not all(r == self.rank or self.mailbox.acked_dead(x, r) for r in live):
                continue
            lost = 0
            for rid, stream_rank, entry_rank in self.mailbox.registered():
                if self.mailbox.owner_of(rid, entry_rank) != x:
                    continue
                self.mailbox.set_owner(rid, -1)
                lost += 1
                reason = f"rank {x} ({self.roles[x]}) died holding the request"
                if stream_rank == self.rank:
                    self._deliver(self.rank, "failed", (rid, reason))
                elif stream_rank is not None and stream_rank not in self.dead:
                    self.outbox.append((stream_rank, "failed", (rid, reason)))
            self.reaped.add(x)
            self.n_reaped += lost
            if lost:
                self.total_finished = self.mailbox.add_finished(lost)

    def step(self) -> int:
        self.node.step()
        return self.exchange()


class _LocalStream:
    """Tokens sampled on the front end's own rank."""

    def __init__(self, engine: RankEngine, request_id):
        self.engine, self.request_id = engine, request_id

    def append_token_id(self, token_id: int, is_last_token: bool = False) -> None:
        self.engine._token(self.request_id, int(token_id), bool(is_last_token))

    def fail(self, exc: BaseException) -> None:
        self.engine._deliver(self.engine.rank, "failed", (self.request_id, str(exc)[:300]))


def entry_rank(kind_ordinal: int, roles: List[str], has_image: bool, dead=()) -> int:
    """cluster.py:178-184: image requests round-robin over the E nodes (`ebalancer`), text-only ones over the P nodes
    (`pbalancer`) — two independent cursors, so `kind_ordinal` counts the requests OF THE SAME KIND seen so far.
    `dead`: ranks the caller has given up on — the round robin goes over the others."""
    ranks = [r for r, t in enumerate(roles) if ("E" if has_image else "P") in t]
    live = [r for r in ranks if r not in dead] or ranks
    return live[kind_ordinal % len(live)]


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
    kind = lambda i: "E" if requests[i].pixel_values is not None else "P"
    # a request whose front-door rank has died enters at a live rank of the same kind instead: every such rank offers, the
    # store lets exactly one admit it (StoreMailbox.claim)
    mine = sorted((i for i in range(len(requests)) if door[i] == engine.rank or kind(i) in engine.roles[engine.rank]),
                  key=lambda i: arrivals[i])
    from hydrainfer_amd.engine.serve import ADMIT_PER_STEP, quiet_gc
    nxt, total = 0, len(requests)
    orphans: List[int] = []
    first_finished, first_failed = len(engine.node.finished), len(engine.node.failed)
    with quiet_gc():
        while True:
            now = time.perf_counter() - t0
            admitted = 0
            while nxt < len(mine) and arrivals[mine[nxt]] <= now and admitted < ADMIT_PER_STEP:
                i = mine[nxt]
                nxt += 1
                if door[i] != engine.rank:
                    orphans.append(i)          # somebody else's: looked at again below, should that rank die
                    continue
                admitted += 1
                if not engine.mailbox.claim(f"door/{i}"):      # (taken over by another rank that had given up on this one)
                    continue
                rcb = creator.process(requests[i])
                engine.admit(rcb)
                rcb.metric.arrival_time = t0 + arrivals[i]
            if engine.dead and orphans:
                for i in [i for i in orphans if door[i] in engine.dead]:
                    orphans.remove(i)
                    if engine.mailbox.claim(f"door/{i}"):
                        rcb = creator.process(requests[i])
                        engine.admit(rcb)
                        rcb.metric.arrival_time = t0 + arrivals[i]
            if engine.step() >= total:
                break
            if now > deadline_s:
                raise TimeoutError(f"rank {engine.rank}: trace not drained after {deadline_s} s")
            if engine.node.idle():
                time.sleep(0.001)       # an idle rank polls its mailbox about 1000 times a second
    t_wait = time.perf_counter()
    while engine.held and time.perf_counter() - t_wait < max(10.0, 2 * engine.dead_after_s):     # FREEs still on their way
        engine.exchange()
        time.sleep(0.0005)
    if device is not None and device.type == "cuda":
        torch.cuda.synchronize(device)
    out = {r.request_id: {"arrival": r.metric.arrival_time, "token_times": list(r.metric.token_times),
                          "tokens": list(r.output_token_ids), "ep_transfer": list(r.metric.ep_transfer),
                          "pd_transfer": list(r.metric.pd_transfer), "path": list(getattr(r, "path", []))}
           for r in engine.node.finished[first_finished:]}
    for r in engine.node.failed[first_failed:]:      # terminated here (a pull that failed twice, a peer that died)
        out[r.request_id] = {"arrival": r.metric.arrival_time, "token_times": list(r.metric.token_times),
                             "tokens": list(r.output_token_ids), "ep_transfer": [], "pd_transfer": [],
                             "path": list(getattr(r, "path", [])), "failed": r.failed}
    return out


def summarize(per_request: Dict[int, dict], t0: float) -> dict:
    failed = [r for r in per_request.values() if r.get("failed")]
    rs = [r for r in per_request.values() if not r.get("failed")]
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
    return {"requests": len(rs), "requests_terminated": len(failed), "output_tokens": n_out, "wall_s": round(end - t0, 3),
            "pulls_per_pair": pairs,
            "output_tok_s": round(n_out / (end - t0), 1),
            "ttft_mean_ms": ms(sum(ttft) / len(ttft)),
            "ttft_p50_ms": ms(pct(ttft, 0.5)), "ttft_p99_ms": ms(pct(ttft, 0.99)),
            "tpot_p50_ms": ms(pct(tpot, 0.5)), "tpot_p99_ms": ms(pct(tpot, 0.99)),
            "ep_pull_p50_ms": ms(pct(hop("ep_transfer"), 0.5)), "pd_pull_p50_ms": ms(pct(hop("pd_transfer"), 0.5))}
