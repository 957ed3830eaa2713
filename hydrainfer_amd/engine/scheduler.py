"""Continuous-batching scheduler — mirror of hydrainfer/engine/scheduler.py:16-200.

Every step: admit waiting requests up to `max_running_requests` (requests that arrive to pull a
migrated cache may overshoot by `max_overload_requests`, which breaks the E<->P pull deadlock the
reference describes at :110-114), grow each running request's block table for the tokens of its
current instruction, then pack one batch: image encodes up to `image_budgets`, fills up to
`token_budgets` tokens in priority order with the last prefill chunked to fit.  Requests that do
not fit stay in `running` for the next step; the caller re-queues executed ones through
`schedule_running`.

The reference derives the two budgets by timing the executor against a TPOT SLO
(engine/profiler.py:182-210); here they are plain config fields, and
`hydrainfer_amd.engine.profiler.BatchSchedulerProfiler` fills them in on a GPU."""
import time
from collections import deque
from dataclasses import dataclass
from typing import Deque, List, Optional

from hydrainfer_amd.engine.isa import Fill, ImageEmbed, ImageEmbedFill, PullCache, TextFill
from hydrainfer_amd.engine.rcb import BatchRequest, RequestControlBlock
from hydrainfer_amd.memory.token_cache_manger import BlockTableManager


@dataclass
class BatchSchedulerMetrics:
    n_running_requests: int
    n_requests_waiting_migrate: int


@dataclass
class BatchSchedulerConfig:
    priority: str = "prefill"          # 'prefill' | 'decode'
    max_running_requests: int = 15
    chunked_prefill: bool = True
    token_budgets: int = 2048
    image_budgets: int = 8
    debug: bool = False


@dataclass
class BatchSchedulerContext:
    kv_cache_block_manager: Optional[BlockTableManager]
    image_cache_block_manager: Optional[BlockTableManager]


class BatchScheduler:
    def __init__(self, config: BatchSchedulerConfig, context: BatchSchedulerContext):
        self.config = config
        self.context = context
        self.token_budgets = config.token_budgets
        self.image_budgets = config.image_budgets
        self.waiting: Deque[RequestControlBlock] = deque()
        self.running: List[RequestControlBlock] = []
        self.step_cnt = 0
        self.next_sid = 1
        self.max_overload_requests = config.max_running_requests
        self.running_cnt = 0
        self.migrating_cnt = 0
        self.stalled_steps = 0
        self.stall_since: Optional[float] = None
        self.stall_timeout_s = 10.0

    # -- requests handed to a downstream node but not pulled yet still hold their blocks here
    def migrating_acquire(self) -> None:
        assert self.migrating_cnt < self.config.max_running_requests + self.max_overload_requests, \
            "invalid acquire"
        self.migrating_cnt += 1

    def migrating_release(self) -> None:
        assert self.migrating_cnt > 0, "invalid release"
        self.migrating_cnt -= 1

    # -- queueing-time stamps (scheduler.py:64-86): first open phase gets the begin stamp,
    #    first phase with exactly one stamp gets the end stamp
    @staticmethod
    def _stamp_begin(rcb: RequestControlBlock) -> None:
        m, now = rcb.metric, time.perf_counter()
        if isinstance(rcb.current_instruction(), ImageEmbed):
            m.encode_queueing.append(now)
        elif not m.prefill_queueing:
            m.prefill_queueing.append(now)
        elif not m.decode_queueing:
            m.decode_queueing.append(now)

    @staticmethod
    def _stamp_end(rcb: RequestControlBlock) -> None:
        m = rcb.metric
        if len(m.decode_queueing) == 2:          # every phase closed long ago: the steady decode steps of a request
            return
        now = time.perf_counter()
        for phase in (m.encode_queueing, m.prefill_queueing, m.decode_queueing):
            if len(phase) == 1:
                phase.append(now)
                return

    def schedule_new(self, rcb: RequestControlBlock) -> None:
        rcb.sid = self.next_sid
        self.next_sid += 1
        if isinstance(rcb.current_instruction(), PullCache):
            self.waiting.appendleft(rcb)     # migrated-in requests jump the queue
        else:
            self.waiting.append(rcb)
        self._stamp_begin(rcb)

    def schedule_running(self, rcb: RequestControlBlock) -> None:
        self.running.append(rcb)
        self._stamp_end(rcb)

    def _admit(self) -> None:
        cap = self.config.max_running_requests - self.migrating_cnt
        while len(self.running) < cap and self.waiting:
            self.schedule_running(self.waiting.popleft())
        while (len(self.running) < cap + self.max_overload_requests and self.waiting
               and isinstance(self.waiting[0].current_instruction(), PullCache)):
            self.schedule_running(self.waiting.popleft())

    def _grow_caches(self) -> set:
        """Grows every running request's block table for its current instruction.  Returns the ids
        of requests whose growth does not fit the pool right now: they sit this step out and retry
        (the reference allocates blindly and dies on 'not enough blocks', token_cache_manger.py:101)."""
        kv, img = self.context.kv_cache_block_manager, self.context.image_cache_block_manager
        deferred = set()

        def fits(manager, vc, n_tokens) -> bool:
            have = len(vc.block_table) if vc is not None else 0
            need = (n_tokens + manager.block_size - 1) // manager.block_size - have
            return need <= len(manager.shared_cache.to_be_evicted)

        for rcb in self.running:
            inst = rcb.current_instruction()
            if isinstance(inst, Fill):
                vc = rcb.virtual_kv_cache
                if vc is not None and len(inst.cache_ids) == 1:
                    # a decode step (one token): 15 times of 16 its slot lies inside the last block already
                    want = inst.cache_ids[0] + 1
                    if want <= len(vc.block_table) * kv.block_size:
                        if want > vc.n_cache_tokens:
                            vc.n_cache_tokens = want
                        continue
                if rcb.virtual_kv_cache is None:
                    if not fits(kv, None, max(inst.cache_ids) + 1):
                        deferred.add(id(rcb))
                        continue
                    rcb.virtual_kv_cache = kv.allocate_virtual_cache(inst.hashes)
                    n_hit = rcb.virtual_kv_cache.n_cache_tokens
                    assert n_hit <= len(inst.token_ids)
                    if n_hit == len(inst.token_ids):
                        # The whole prompt is cached (its length is a multiple of the block size).
                        # The reference skips the fill altogether (scheduler.py:128-133) and so
                        # never samples the first token; here the last block is computed again.
                        n_hit -= kv.block_size
                        kv.realloc(rcb.virtual_kv_cache, n_hit)
                    if n_hit > 0:
                        # prefix-cache hit: split the matched tokens off and skip them
                        inst.chunk_prefill(chunk_size=n_hit)
                        rcb.step()
                inst = rcb.current_instruction()
                if isinstance(inst, Fill):
                    want = max(rcb.virtual_kv_cache.n_cache_tokens, max(inst.cache_ids) + 1)
                    if not fits(kv, rcb.virtual_kv_cache, want):
                        deferred.add(id(rcb))
                        continue
                    kv.realloc(rcb.virtual_kv_cache, want)
            elif isinstance(inst, ImageEmbed):
                want = max(inst.cache_ids) + 1
                if rcb.virtual_image_cache is None:
                    if not fits(img, None, want):
                        deferred.add(id(rcb))
                        continue
                    rcb.virtual_image_cache = img.allocate_virtual_cache()
                want = max(rcb.virtual_image_cache.n_cache_tokens, want)
                if not fits(img, rcb.virtual_image_cache, want):
                    deferred.add(id(rcb))
                    continue
                img.realloc(rcb.virtual_image_cache, want)
        return deferred

    def step(self) -> BatchRequest:
        self.step_cnt += 1
        self._admit()
        self.running_cnt = len(self.running)
        if not self.running:
            return BatchRequest()
        deferred = self._grow_caches()

        embeds, prefills, decodes = [], [], []
        this_step: List[RequestControlBlock] = []
        next_step: List[RequestControlBlock] = []
        for rcb in self.running:
            inst = rcb.current_instruction()
            if id(rcb) in deferred:
                next_step.append(rcb)      # no room in the pool this step
            elif isinstance(inst, Fill):
                (decodes if len(inst.token_ids) == 1 else prefills).append(rcb)
            elif isinstance(inst, ImageEmbed):
                embeds.append(rcb)
            else:
                this_step.append(rcb)      # Empty / Migrate / PullCache cost no budget

        this_step += embeds[: self.image_budgets]      # one image per request
        next_step += embeds[self.image_budgets:]

        n_tok, budget = 0, self.token_budgets
        fills = prefills + decodes if self.config.priority == "prefill" else decodes + prefills
        for rcb in fills:
            inst = rcb.current_instruction()
            n = len(inst.token_ids)
            if n_tok + n <= budget:
                this_step.append(rcb)
                n_tok += n
            elif (n_tok < budget and n > 1 and self.config.chunked_prefill
                  and isinstance(inst, (TextFill, ImageEmbedFill))):
                inst.chunk_prefill(budget - n_tok)
                this_step.append(rcb)
                n_tok = budget
            elif n_tok == 0:               # a fill larger than the whole budget must still run
                this_step.append(rcb)
                n_tok += n
            else:
                next_step.append(rcb)

        if deferred and not this_step:
            # Nothing can run this step: every running request waits for cache blocks.  That is
            # ordinary back-pressure while blocks are pinned by requests a downstream node has
            # not pulled yet (their FREE message returns them) — an empty step takes microseconds,
            # so counting steps would kill a healthy node within a fraction of a second.  It is a
            # deadlock only when nothing outside this scheduler can return blocks (no migrating
            # request) and the state has not changed for stall_timeout_s of wall-clock time.
            now = time.monotonic()
            self.stalled_steps += 1
            if self.stall_since is None:
                self.stall_since = now
            if self.migrating_cnt == 0 and now - self.stall_since > self.stall_timeout_s:
                raise RuntimeError("cache pool exhausted: every running request has been waiting for blocks "
                                   f"for {self.stall_timeout_s:.0f} s and no migrating request can free any")
            time.sleep(0.0002)             # back off instead of spinning the host and the control plane
        else:
            self.stalled_steps = 0
            self.stall_since = None
        self.running = next_step
        return BatchRequest(this_step)

    def get_metrics(self) -> BatchSchedulerMetrics:
        return BatchSchedulerMetrics(self.running_cnt, self.migrating_cnt)
