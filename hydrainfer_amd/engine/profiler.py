"""Scheduler budgets from a TPOT SLO — mirror of hydrainfer/engine/profiler.py:36-210.

The batch scheduler packs at most `image_budgets` encodes and `token_budgets` fill tokens per
step; the reference derives both by timing the executors on synthetic batches and searching for the
largest batch whose step stays under `tpot_slo - 0.01` s (default SLO 0.4 s, E/EP/P nodes 1 s,
config/engine/batch_scheduler_profiler.yaml:2).  Same search (including its +1/-1 stepping, which
can return one past the last size that met the SLO), same synthetic batches: 336x336 random image
per encode, 16-token prompts for fills."""
import copy
import time
from dataclasses import dataclass
from typing import Callable, Optional

import torch

from hydrainfer_amd.engine.isa import ImageEmbed, Instruction, InstructionListBuilder, TextFill
from hydrainfer_amd.engine.rcb import BatchRequest, RequestControlBlock, SamplingParameters


@dataclass
class BatchSchedulerProfilerConfig:
    tpot_slo: float = 0.4
    n_warmup_iter: int = 3
    n_profile_iter: int = 3


def binary_search_max_batch_size(left: int, right: int, criterion: Callable[[int], bool]) -> int:
    """profiler.py:118-133.  A criterion that raises counts as not met."""
    while left < right:
        mid = (left + right + 1) // 2
        try:
            ok = criterion(mid)
        except Exception:
            ok = False
        if ok:
            left = mid + 1      # (sic) latency is not monotonic in batch size; avoids a dead loop
        else:
            right = mid - 1
    return left


class BatchSchedulerProfiler:
    def __init__(self, config: BatchSchedulerProfilerConfig, executor, kv_cache_block_manager,
                 image_cache_block_manager, pixel_values: Optional[torch.Tensor] = None,
                 n_image_tokens: int = 576, device: Optional[torch.device] = None):
        self.config, self.executor = config, executor
        self.kv, self.img = kv_cache_block_manager, image_cache_block_manager
        self.pixel_values, self.n_image_tokens = pixel_values, n_image_tokens
        self.device = device

    def _prepare_rcb(self, inst: Instruction) -> RequestControlBlock:
        b = InstructionListBuilder()
        for _ in range(self.config.n_warmup_iter + self.config.n_profile_iter):
            b.append(copy.deepcopy(inst))
        rcb = RequestControlBlock()
        rcb.request_id = -1
        rcb.instructions = b.build_instruction_list()
        rcb.sampling_params = SamplingParameters(max_tokens=1 << 30)
        return rcb

    def _prepare_encode_batch(self, batch_size: int) -> BatchRequest:
        inst = ImageEmbed(self.pixel_values, list(range(self.n_image_tokens)), [(336, 336)], None)
        batch = BatchRequest()
        try:
            for _ in range(batch_size):
                rcb = self._prepare_rcb(inst)
                batch.append(rcb)
                rcb.virtual_image_cache = self.img.allocate_virtual_cache()
                self.img.realloc(rcb.virtual_image_cache, self.n_image_tokens)
        except Exception:
            self._free(batch)      # a size that does not fit the pool is simply "not met"
            raise
        # ImageEmbed drops its pixels once executed (executor.py:228); every copy needs its own
        for rcb in batch.rcbs:
            for inst_copy in rcb.instructions:
                if isinstance(inst_copy, ImageEmbed):
                    inst_copy.pixel_values = self.pixel_values
        return batch

    def _prepare_prefill_batch(self, batch_size: int) -> BatchRequest:
        n = 16
        inst = TextFill(list(range(n)), list(range(n)), list(range(n)), True, None, None)
        batch = BatchRequest()
        try:
            for _ in range(batch_size // n):
                rcb = self._prepare_rcb(inst)
                batch.append(rcb)
                rcb.virtual_kv_cache = self.kv.allocate_virtual_cache()
                self.kv.realloc(rcb.virtual_kv_cache, n)
        except Exception:
            self._free(batch)
            raise
        return batch

    def _free(self, batch: BatchRequest) -> None:
        for rcb in batch.rcbs:
            if rcb.virtual_image_cache is not None:
                self.img.realloc(rcb.virtual_image_cache, 0)
            if rcb.virtual_kv_cache is not None:
                self.kv.realloc(rcb.virtual_kv_cache, 0)

    def _sync(self) -> None:
        if self.device is not None and self.device.type == "cuda":
            torch.cuda.synchronize(self.device)

    def _criterion(self, size: int, prepare, execute) -> bool:
        batch = prepare(size)
        try:
            for _ in range(self.config.n_warmup_iter):
                execute(batch)
            self._sync()
            t0 = time.perf_counter()
            for _ in range(self.config.n_profile_iter):
                execute(batch)
                self._sync()
            avg = (time.perf_counter() - t0) / self.config.n_profile_iter
        finally:
            self._free(batch)
        return avg < self.config.tpot_slo - 0.01

    def profile_image_budgets(self) -> int:
        return binary_search_max_batch_size(
            1, 8, lambda n: self._criterion(n, self._prepare_encode_batch, self.executor.execute_image_embed))

    def profile_token_budgets(self) -> int:
        return binary_search_max_batch_size(
            1, 2048, lambda n: self._criterion(n, self._prepare_prefill_batch, self.executor.execute_fill))
