"""E/P/D node step loop — mirror of hydrainfer/cluster/epdnode.py:238-447 (AsyncEPDNode.step and
the four-phase migrate protocol) and hydrainfer/cluster/migrate.py:5-24 (NodeType), without the
Ray/asyncio control plane: nodes of one `LocalCluster` live in one process and call each other
directly, which keeps the protocol order (1 sender announces, 2 receiver queues a PullCache,
3 receiver allocates + pulls, 4 sender frees) and lets it run under test.

Data path of a pull = TokenCacheBlockManager.migrate_blocks -> hx_migrate_blocks (one gather
kernel over the peer pool); the control plane moves block tables only."""
import copy
import time
from typing import Dict, List

from hydrainfer_amd.engine.executor import InstructionExecutor
from hydrainfer_amd.engine.isa import (EmptyInstruction, EPMigrate, Fill, ImageEmbed, MigrateRequest,
                                       PullCache)
from hydrainfer_amd.engine.rcb import BatchRequest, RequestControlBlock, ScenarioType
from hydrainfer_amd.engine.scheduler import BatchScheduler
from hydrainfer_amd.utils.profiler import profile


class NodeType:
    def __init__(self, node_type: str = "EPD"):
        assert node_type in ("E", "P", "D", "EP", "ED", "PD", "EPD"), f"invalid node type {node_type}"
        self.node_type = node_type
        self.enable_encode = "E" in node_type
        self.enable_prefill = "P" in node_type
        self.enable_decode = "D" in node_type
        self.has_kv_cache = self.has_language_model = "P" in node_type or "D" in node_type
        self.has_image_cache = "E" in node_type or "P" in node_type
        self.has_vision_model = "E" in node_type

    def __eq__(self, other):
        return self.node_type == (other.node_type if isinstance(other, NodeType) else other)


class RoundRobin:
    """LoadBalancer(policy='round') keyed by scenario with fall-through to any non-empty key
    (cluster/loadbalancer.py:14-68)."""

    def __init__(self):
        self.workers: Dict[int, list] = {int(s): [] for s in ScenarioType}
        self.cursor: Dict[int, int] = {int(s): 0 for s in ScenarioType}
        self.lost_workers = 0

    def register_worker(self, key, worker) -> None:
        self.workers[int(key)].append(worker)

    def remove_worker(self, worker) -> None:
        """A downstream node died: the round robin goes on over the others."""
        self.lost_workers += 1
        for key, ws in self.workers.items():
            if worker in ws:
                ws.remove(worker)
                self.cursor[key] = self.cursor[key] % len(ws) if ws else 0

    def _choice(self, key: int):
        ws = self.workers[key]
        if not ws:
            return None
        w = ws[self.cursor[key]]
        self.cursor[key] = (self.cursor[key] + 1) % len(ws)
        return w

    def choice(self, key):
        w = self._choice(int(key))
        if w is not None:
            return w
        for k in self.workers:
            w = self._choice(k)
            if w is not None:
                return w
        return None


MAX_MIGRATE_ATTEMPTS = 2      # hydrainfer/cluster/epdnode.py:428: "max_retries = 2", then the request is terminated


class EPDNode:
    def __init__(self, name: str, node_type: NodeType, scheduler: BatchScheduler,
                 executor: InstructionExecutor, kv_cache_block_manager, image_cache_block_manager,
                 tpot_slo: float = 0.4, eager_migrate: bool = False):
        self.name, self.node_type, self.tpot_slo = name, node_type, tpot_slo
        # The reference spends one whole engine step on every EPMigrate / PDMigrate instruction
        # (the request is re-queued, scheduled, and only then handed over or — on a collocated
        # node — skipped).  With eager_migrate a request whose encode / prefill has just run is
        # handed over in the SAME step: one step (a decode step's worth of TTFT) less per hop.
        self.eager_migrate = eager_migrate
        self.batch_scheduler = scheduler
        self.executor = executor
        self.kv_cache_block_manager = kv_cache_block_manager
        self.image_cache_block_manager = image_cache_block_manager
        self.ep_loadbalancer, self.pd_loadbalancer = RoundRobin(), RoundRobin()
        self.finished: List[RequestControlBlock] = []
        self.failed: List[RequestControlBlock] = []      # requests this node terminated (terminate())
        self.peer_alive = None      # callable(node) -> bool, set by engine.distributed.RankEngine (a dead process's pool is gone)

    # ---- migrate graph (epdnode.py:60-75): strict traffic only to nodes with a tight TPOT SLO
    def connect(self, ep_targets: List["EPDNode"], pd_targets: List["EPDNode"]) -> None:
        for targets, lb in ((ep_targets, self.ep_loadbalancer), (pd_targets, self.pd_loadbalancer)):
            for node in targets:
                if node.tpot_slo < 0.05:
                    lb.register_worker(ScenarioType.Strict, node)
                lb.register_worker(ScenarioType.Relaxed, node)

    def add_request(self, rcb: RequestControlBlock) -> None:
        rcb.metric.arrival_time = time.perf_counter()
        self.batch_scheduler.schedule_new(rcb)

    def idle(self) -> bool:
        s = self.batch_scheduler
        return not s.waiting and not s.running

    # ---- one engine step (epdnode.py:238-337)
    def step(self) -> int:
        fe = self.executor.fill_executor
        if fe is not None and fe.graph_decoder is not None:
            # the steady state of a decode batch runs without the scheduler (executor.DecodeCohort); anything else —
            # an arrival, a request about to finish, a hand-over — ends it and takes the general path below
            n = fe.cohort_step(self.batch_scheduler)
            if n:
                if fe.ended_rows is not None:
                    # an end-of-sequence id came back: what the loop at the end of a general step does, row by row in batch
                    # order — the first finished request makes the pending launch's tokens known, which may end others
                    rows, fe.ended_rows = fe.ended_rows, None
                    now = time.perf_counter()
                    keep = []
                    for rcb in rows:
                        if rcb.is_finished():
                            self.executor.resolve_pending()
                            rcb.metric.finished_time = now
                            self._free_cache(rcb)
                            rcb.release_instructions()
                            self.finished.append(rcb)
                        else:
                            keep.append(rcb)
                    self.batch_scheduler.running = keep
                return n
        with profile("schedule"):
            batch = self.batch_scheduler.step()
        if len(batch) == 0:
            return 0
        fill, embed, empty, migrate, pull = (BatchRequest() for _ in range(5))
        for rcb, inst in batch:
            if isinstance(inst, Fill):
                fill.append(rcb)
            elif isinstance(inst, EmptyInstruction):
                empty.append(rcb)
            elif isinstance(inst, ImageEmbed):
                embed.append(rcb)
            elif isinstance(inst, MigrateRequest):
                migrate.append(rcb)
            elif isinstance(inst, PullCache):
                pull.append(rcb)
            else:
                raise RuntimeError(f"unsupported instruction {type(inst)}")

        now = time.perf_counter()
        for rcb, inst in fill:
            phase = rcb.metric.prefill_execute if len(inst.token_ids) > 1 else rcb.metric.decode_execute
            if not phase:
                phase.append(now)
        for rcb, _ in embed:
            rcb.metric.encode_execute.append(now)

        with profile("encode"):
            self.executor.execute_image_embed(embed)
        with profile("fill"):
            self.executor.execute_fill(fill)
        self.executor.execute_empty(empty)

        now = time.perf_counter()
        for rcb, _ in embed:
            rcb.metric.encode_execute.append(now)
        for group in (embed, fill, empty, pull):
            for rcb in group.rcbs:
                if self.eager_migrate and group is not pull and not rcb.is_finished() \
                        and isinstance(rcb.current_instruction(), MigrateRequest):
                    migrate.append(rcb)
                    continue
                if rcb.is_finished():
                    # its last tokens may still be in flight (decode look-ahead); they must be on
                    # the host, and the step that wrote its blocks done, before the blocks go back
                    self.executor.resolve_pending()
                    rcb.metric.finished_time = now
                    self._free_cache(rcb)
                    rcb.release_instructions()
                    self.finished.append(rcb)
                else:
                    self.batch_scheduler.schedule_running(rcb)
        # The reference runs the two migrate handlers as asyncio tasks that only get the loop once
        # step() has returned (epdnode.py:286-297,345-347), i.e. AFTER the re-queue above: a request
        # that stays on this node re-enters `running` behind this step's other requests, and a
        # pulled request is already re-queued when its PullCache completes.  Same order here.
        with profile("migrate+pull"):
            self._execute_batch_migrate(migrate)
            self._execute_pull_cache(pull)
        for m in (self.kv_cache_block_manager, self.image_cache_block_manager):
            if m is not None:
                m.synchronize()
        return len(batch)

    # ---- 1. sender: hand the request (with its block tables) to the next stage
    def _execute_batch_migrate(self, batch: BatchRequest) -> None:
        for rcb, inst in batch:
            rcb.step()
            assert isinstance(rcb.current_instruction(), PullCache)
            rcb.current_instruction().hop = "ep" if isinstance(inst, EPMigrate) else "pd"
            lb = self.ep_loadbalancer if isinstance(inst, EPMigrate) else self.pd_loadbalancer
            node = lb.choice(rcb.scenario_type)
            if node is None and lb.lost_workers:
                # every downstream node of this hop has died: the reference's hand-over RPC would fail twice and the
                # request be terminated (epdnode.py:428-442) — nowhere to retry, so it ends here
                self.terminate(rcb, f"{self.name}: no live {'prefill' if isinstance(inst, EPMigrate) else 'decode'} node to hand over to")
                continue
            if node is None or node is self:
                rcb.step()                          # stays here: skip the pull as well
                self.batch_scheduler.schedule_running(rcb)
                continue
            self.batch_scheduler.migrating_acquire()
            self.executor.resolve_pending()
            self._drain_compute()                   # the peer reads these blocks on another stream
            node.migrate(self, rcb)

    def _drain_compute(self) -> None:
        for m in (self.kv_cache_block_manager, self.image_cache_block_manager):
            device = getattr(m, "device", None)
            if device is not None and device.type == "cuda":
                import torch
                torch.cuda.current_stream(device).synchronize()
                return

    # ---- 2. receiver: queue the request; its PullCache runs when the scheduler admits it
    def migrate(self, src_node: "EPDNode", rcb: RequestControlBlock) -> None:
        rcb.current_instruction().src_node = src_node
        self.batch_scheduler.schedule_new(rcb)

    # ---- 3. receiver: allocate local blocks, pull, then tell the sender to free
    def _migrate_virtual_cache(self, src_cache, manager, src_node, which: str, pulled: list):
        dst = manager.allocate_virtual_cache()
        pulled.append((manager, dst))
        manager.realloc(dst, src_cache.n_cache_tokens)
        # a send/recv transfer (ranks on different hosts, or intranode_migrate_backend='nccl') has two
        # halves: ask the sender for its half first (the reference's pull_virtual_cache.remote,
        # epdnode.py:362-378,394-400); the IPC pull is one-sided
        needs = getattr(manager, "needs_sender", None)
        if needs is not None and needs(src_cache, dst):
            src_node.pull_virtual_cache(which, src_cache, dst)
        manager.migrate_blocks(src_cache, dst, is_send=False)
        return dst

    # ---- 3b. sender: its half of a send/recv transfer (no-op for IPC pulls, never called for them)
    def pull_virtual_cache(self, which: str, src_cache, dst_cache) -> None:
        manager = self.kv_cache_block_manager if which == "kv" else self.image_cache_block_manager
        manager.migrate_blocks(src_cache, dst_cache, is_send=True)

    def _can_pull(self, rcb: RequestControlBlock) -> bool:
        """Room for the caches of a request that wants to move in?  (The reference allocates
        blindly and dies on its 'not enough blocks' assert; here the request keeps waiting at its
        PullCache and is retried next step — its blocks stay pinned at the sender meanwhile.)"""
        for vc, manager, wanted in ((rcb.virtual_kv_cache, self.kv_cache_block_manager, self.node_type.has_kv_cache),
                                    (rcb.virtual_image_cache, self.image_cache_block_manager,
                                     self.node_type.has_image_cache)):
            if vc is not None and wanted:
                need = (vc.n_cache_tokens + manager.block_size - 1) // manager.block_size
                if need > len(manager.shared_cache.to_be_evicted):
                    return False
        return True

    def _execute_pull_cache(self, batch: BatchRequest) -> None:
        for rcb, inst in batch:
            if self.peer_alive is not None and not self.peer_alive(inst.src_node):
                # the sender's process is gone and its pool with it: nothing to pull, nobody to tell
                self._unqueue(rcb)
                rcb.virtual_kv_cache = rcb.virtual_image_cache = None      # (the sender's tables, not blocks of ours)
                self.terminate(rcb, f"{self.name}: the node holding this request's cache blocks died before the pull")
                continue
            if not self._can_pull(rcb):
                continue
            # upstream stamps "first pull = ep, second = pd" (epdnode.py:384-387), which files a
            # text-only or EP-node request's P->D pull under ep_transfer; the hop is known here
            m = rcb.metric
            stamps = m.ep_transfer if inst.hop == "ep" else m.pd_transfer
            stamps.append(time.perf_counter())
            old = copy.copy(rcb)
            pulled = []       # (manager, cache) allocated here so far: returned to the pool if the pull fails
            try:
                new_kv = new_img = None
                if old.virtual_kv_cache is not None and self.node_type.has_kv_cache:
                    new_kv = self._migrate_virtual_cache(old.virtual_kv_cache, self.kv_cache_block_manager,
                                                         inst.src_node, "kv", pulled)
                if old.virtual_image_cache is not None and self.node_type.has_image_cache:
                    new_img = self._migrate_virtual_cache(old.virtual_image_cache, self.image_cache_block_manager,
                                                          inst.src_node, "image", pulled)
                # the sender may only free once the copy has been issued AND completed
                for manager in (self.kv_cache_block_manager, self.image_cache_block_manager):
                    if manager is not None:
                        manager.synchronize()
            except RuntimeError as e:
                # A pull that fails (the peer of a send/recv transfer never shows up: MigrationTimeout; the peer's pool
                # cannot be mapped: HydraHipError) ends THIS REQUEST, not the rank — the reference retries the hand-over
                # twice and then terminates the request with a None token (epdnode.py:428-442).  The blocks taken here
                # go back; on the last attempt the sender is told to free its side.
                for manager, vc in pulled:
                    manager.realloc(vc, 0)
                stamps.pop()
                inst.attempts = getattr(inst, "attempts", 0) + 1
                if inst.attempts < MAX_MIGRATE_ATTEMPTS:
                    continue                          # still queued at its PullCache: tried again next step
                self._unqueue(rcb)
                inst.src_node.free_migrate_request(old)
                rcb.virtual_kv_cache = rcb.virtual_image_cache = None
                self.terminate(rcb, f"{self.name}: pulling the cache blocks failed {inst.attempts} times: {e!r}"[:400])
                continue
            rcb.virtual_kv_cache, rcb.virtual_image_cache = new_kv, new_img
            inst.src_node.free_migrate_request(old)
            rcb.step()
            stamps.append(time.perf_counter())

    def _unqueue(self, rcb: RequestControlBlock) -> None:
        """Take a request that step() has already re-queued out of the scheduler again."""
        s = self.batch_scheduler
        if rcb in s.running:
            s.running.remove(rcb)
        elif rcb in s.waiting:
            s.waiting.remove(rcb)

    def terminate(self, rcb: RequestControlBlock, reason: str) -> None:
        """End one request without ending the node (epdnode.py:440-442: free its blocks, push `(request_id, None)` to
        the stream): its caches here are freed, every output processor is told (OutputTokenProcessor.fail — a None
        token unless the processor knows better), and it is listed under `failed`."""
        self._free_cache(rcb)
        rcb.virtual_kv_cache = rcb.virtual_image_cache = None
        rcb.failed = reason
        rcb.metric.finished_time = time.perf_counter()
        for p_ in rcb.output_token_processors:
            p_.fail(RuntimeError(reason))
        rcb.release_instructions()
        self.failed.append(rcb)

    # ---- 4. sender: release the blocks of a request that has been pulled
    def free_migrate_request(self, rcb: RequestControlBlock) -> None:
        self._free_cache(rcb)
        self.batch_scheduler.migrating_release()

    def _free_cache(self, rcb: RequestControlBlock) -> None:
        if rcb.virtual_kv_cache is not None and self.kv_cache_block_manager is not None:
            self.kv_cache_block_manager.realloc(rcb.virtual_kv_cache, 0)
        if rcb.virtual_image_cache is not None and self.image_cache_block_manager is not None:
            self.image_cache_block_manager.realloc(rcb.virtual_image_cache, 0)


class LocalCluster:
    """Nodes + routing: new requests go round-robin to the E nodes (text-only ones to the P
    nodes), cluster.py:178-184; every E node may hand over to every P node and every P node to
    every D node (MigrateGraphBuilder.build_graph, migrate.py:96-119)."""

    def __init__(self, nodes: List[EPDNode]):
        self.nodes = nodes
        e = [n for n in nodes if n.node_type.enable_encode]
        p = [n for n in nodes if n.node_type.enable_prefill]
        d = [n for n in nodes if n.node_type.enable_decode]
        for n in nodes:
            n.connect(p if n.node_type.enable_encode else [], d if n.node_type.enable_prefill else [])
        self._entry_image, self._entry_text = e, p
        self._next_image = self._next_text = 0

    def add_request(self, rcb: RequestControlBlock) -> None:
        if isinstance(rcb.current_instruction(), ImageEmbed):
            node = self._entry_image[self._next_image % len(self._entry_image)]
            self._next_image += 1
        else:
            node = self._entry_text[self._next_text % len(self._entry_text)]
            self._next_text += 1
        node.add_request(rcb)

    def step(self) -> int:
        return sum(node.step() for node in self.nodes)

    def idle(self) -> bool:
        return all(n.idle() for n in self.nodes)

    def run_until_idle(self, max_steps: int = 1 << 20) -> int:
        steps = 0
        while not self.idle() and steps < max_steps:
            self.step()
            steps += 1
        return steps

    def finished(self) -> List[RequestControlBlock]:
        return [r for n in self.nodes for r in n.finished]
