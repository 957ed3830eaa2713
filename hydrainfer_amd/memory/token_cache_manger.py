"""TokenCacheBlockManager — host-side mirror of hydrainfer/memory/token_cache_manger.py:51-179.

HBM layout: one pool
    (n_layers, n_tokens in {1 image, 2 k/v}, n_blocks, block_size, n_heads, head_size)
so that every per-layer K or V cache is a contiguous [n_blocks, block_size, H, D] slab and
a (layer, k/v, block) triple is one contiguous block_size*H*D run — the unit the migration
gather kernel copies.  The (layer, k/v) planes lie a few hundred bytes apart (memory/kv_pool.py:
K and V of one token on different HBM channels); everything else is the reference's format."""
from dataclasses import dataclass, field
from typing import Dict, List, Optional

import torch

import hydrainfer_amd.memory.kv_pool as kv_pool
from hydrainfer_amd._C.data_transfer.block_migration import get_ipc_mem_handle
from hydrainfer_amd.memory.block_allocator import BlockAllocator, BlockAllocatorMetrics
from hydrainfer_amd.memory.communication import (CommunicationBackendManager,
                                                 CommunicationBackendManagerConfig,
                                                 CommunicationBackendManagerContext)
from hydrainfer_amd.memory.shared_cache import SharedCache, SharedCacheConfig
from hydrainfer_amd.memory.token_cache import TokenCache, VirtualTokenCache

_DTYPES = {"fp16": torch.float16, "bf16": torch.bfloat16, "fp32": torch.float32}


@dataclass
class TokenCacheManagerMetrics:
    allocator_metrics: BlockAllocatorMetrics
    cache_hit_rate: float


@dataclass
class TokenCacheBlockManagerConfig:
    communication_backend_manager_config: CommunicationBackendManagerConfig = field(
        default_factory=CommunicationBackendManagerConfig)
    n_layers: int = 32
    n_tokens: int = 2
    n_blocks: int = 1024
    block_size: int = 16
    n_heads: int = 32
    head_size: int = 128
    dtype: str = "fp16"  # 'bf16' accepted (extension; reference str2dtype knows fp16/fp32 only)
    device: str = "cuda:0"


@dataclass
class TokenCacheBlockManagerContext:
    rank: int
    rank2host: Dict[int, str]


class _IncreasingAllocator:
    def __init__(self, first_value: int = 0):
        self.next = first_value

    def allocate(self) -> int:
        v = self.next
        self.next += 1
        return v


class BlockTableManager:
    """The host-only half of the block manager: free list, prefix-hash table, virtual caches and
    their block tables (token_cache_manger.py:97-153 of the reference).  The scheduler and the
    parameter builders need nothing else, so they run (and are tested) without a GPU."""

    def __init__(self, n_blocks: int, block_size: int, rank: int = 0,
                 memory_handle: Optional[List[int]] = None):
        self.n_blocks, self.block_size, self.rank = n_blocks, block_size, rank
        self.memory_handle: List[int] = memory_handle if memory_handle is not None else []
        self.block_allocator = BlockAllocator(self.n_blocks)
        self.vid_allocator = _IncreasingAllocator(first_value=1)
        self.shared_cache = SharedCache(SharedCacheConfig(n_blocks=self.n_blocks))
        self.total_block_queried = 0.0
        self.total_block_matched = 0.0

    def synchronize(self) -> None:
        pass

    def get_num_avaiable_blocks(self) -> int:
        return self.block_allocator.get_num_avaiable_blocks() + self.shared_cache.get_num_avaiable_blocks()

    def _allocate_new_blocks(self, n_blocks: int) -> List[int]:
        return allocate_new_blocks(self.block_allocator, self.shared_cache, n_blocks)

    def allocate_virtual_cache(self, hashes: Optional[List[int]] = None) -> VirtualTokenCache:
        if hashes is None:
            n_cached_tokens, matched = 0, []
        else:
            ids = self.shared_cache.match(hashes)
            matched = ids[: ids.index(-1) if -1 in ids else len(ids)]
            self.shared_cache.pin(matched)
            n_cached_tokens = len(matched) * self.block_size
            self.total_block_matched += len(matched)
            self.total_block_queried += len(hashes)
        return VirtualTokenCache(vid=self.vid_allocator.allocate(), n_cache_tokens=n_cached_tokens,
                                 block_table=matched, memory_handle=self.memory_handle,
                                 rank=self.rank, n_blocks_of_cache_manager=self.n_blocks)

    def v2p(self, virtual_cache: VirtualTokenCache, virtual_cache_ids: List[int]) -> List[int]:
        return v2p(virtual_cache.block_table, virtual_cache_ids, self.block_size)

    def set_blocks(self, virtual_cache: VirtualTokenCache, virtual_block_ids: List[int],
                   hashes: List[int]) -> None:
        assert len(virtual_block_ids) == len(hashes)
        physical = [virtual_cache.block_table[v] for v in virtual_block_ids]
        self.shared_cache.insert(hashes=hashes, block_ids=physical)

    def realloc(self, virtual_cache: VirtualTokenCache, n_tokens: int) -> None:
        realloc(self.block_allocator, self.shared_cache, virtual_cache, n_tokens, self.block_size)

    def get_metrics(self) -> TokenCacheManagerMetrics:
        rate = self.total_block_matched / self.total_block_queried if self.total_block_queried else 0.0
        return TokenCacheManagerMetrics(self.block_allocator.get_metrics(), rate)


class TokenCacheBlockManager(BlockTableManager):
    def __init__(self, config: TokenCacheBlockManagerConfig, context: TokenCacheBlockManagerContext):
        self.config = config
        self.context = context
        self.n_layers, self.n_tokens = config.n_layers, config.n_tokens
        self.n_heads, self.head_size = config.n_heads, config.head_size
        self.dtype = _DTYPES[config.dtype]
        itemsize = torch.empty((), dtype=self.dtype).element_size()
        n_blocks = ipc_safe_n_blocks(config.n_blocks, self.n_layers * self.n_tokens * config.block_size *
                                     self.n_heads * self.head_size * itemsize,
                                     extra_bytes=self.n_layers * self.n_tokens * kv_pool.KV_POOL_SKEW_BYTES)
        self.device = torch.device(config.device)

        # reference fills the pool with randn ("garbage but finite", token_cache_manger.py:65)
        # the same 6-D shape and the same per-layer views, with the (layer, k/v) planes a few hundred bytes apart so
        # that K and V of one token do not share an HBM channel (memory/kv_pool.py)
        self.cache_tensor = kv_pool.allocate_kv_pool(
            (self.n_layers, self.n_tokens, n_blocks, config.block_size, self.n_heads, self.head_size),
            self.dtype, self.device, fill="randn")
        super().__init__(n_blocks, config.block_size, context.rank, get_ipc_mem_handle(self.cache_tensor))
        self.migrate_stream = torch.cuda.Stream(device=self.device)
        self.migrate_manager = CommunicationBackendManager(
            config.communication_backend_manager_config,
            CommunicationBackendManagerContext(migrate_stream=self.migrate_stream,
                                               cache=self.cache_tensor, n_blocks=self.n_blocks,
                                               rank2host=context.rank2host))

    def get_layer_cache(self, layer_id: int) -> TokenCache:
        return TokenCache([self.cache_tensor[layer_id, t] for t in range(self.n_tokens)])

    def migrate_blocks(self, src_virtual_cache: VirtualTokenCache,
                       dst_virtual_cache: VirtualTokenCache, is_send: bool = False) -> None:
        self.migrate_manager.migrate_blocks(src_virtual_cache, dst_virtual_cache, is_send)

    def needs_sender(self, src_virtual_cache: VirtualTokenCache, dst_virtual_cache: VirtualTokenCache) -> bool:
        return self.migrate_manager.needs_sender(src_virtual_cache.rank, dst_virtual_cache.rank)

    def synchronize(self) -> None:
        self.migrate_stream.synchronize()

    @classmethod
    def compute_n_blocks(cls, config: TokenCacheBlockManagerConfig, memory: int) -> int:
        itemsize = torch.empty((), dtype=_DTYPES[config.dtype]).element_size()
        return memory // (config.n_layers * config.n_tokens * config.block_size * config.n_heads *
                          config.head_size * itemsize)


# --- pure host logic, usable (and tested) without a GPU --------------------------------
def ipc_safe_n_blocks(n_blocks: int, bytes_per_block: int, extra_bytes: int = 0) -> int:
    """Smallest block count >= n_blocks whose pool size (+ extra_bytes: what the allocation holds beside the blocks,
    memory/kv_pool.py's plane skew) is not in [7/8 * 2^k, 2^k).
    Observed on this MI355X / ROCm 7.2 pool (tools/ipc_probe2.py): hipIpcOpenMemHandle of an
    allocation of 14, 14.65, 15, 15.5, 30 or 31 GiB never returns, while 8.5, 12, 13, 16, 17, 20
    and 24 GiB map in 1 ms.  Pools that other processes map are therefore sized past the window."""
    size = n_blocks * bytes_per_block + extra_bytes
    if n_blocks * bytes_per_block <= 0:
        return n_blocks
    p2 = 1 << (size - 1).bit_length()          # next power of two >= size
    if size < p2 and size * 8 >= p2 * 7:
        return -(-(p2 - extra_bytes) // bytes_per_block)
    return n_blocks



def v2p(block_table: List[int], virtual_cache_ids: List[int], block_size: int) -> List[int]:
    """slot = table[id // bs] * bs + id % bs  (token_cache_manger.py:126-133)."""
    return [block_table[i // block_size] * block_size + i % block_size for i in virtual_cache_ids]


def allocate_new_blocks(block_allocator: BlockAllocator, shared_cache: SharedCache,
                        n_blocks: int) -> List[int]:
    """token_cache_manger.py:93-99: free list first, then evict unpinned shared blocks.
    Where the free list holds some but not all of the blocks the reference evicts `n_blocks` more
    (not the shortfall) and then fails its own length assert; here the free-list blocks are pinned
    first (they sit in the eviction set until then, shared_cache.py:26) and only the shortfall is
    evicted.  Identical results whenever the reference does not assert."""
    # every unpinned block (free-list ones included) sits in the eviction set: fail before taking any
    assert n_blocks <= len(shared_cache.to_be_evicted), "not enough blocks"
    block_ids = block_allocator.allocate(n_blocks)
    shared_cache.pin(block_ids)
    if len(block_ids) < n_blocks:
        evicted = shared_cache.allocate(n_blocks - len(block_ids))
        shared_cache.pin(evicted)
        block_ids += evicted
    assert len(block_ids) == n_blocks, "not enough blocks"
    return block_ids


def realloc(block_allocator: BlockAllocator, shared_cache: SharedCache,
            virtual_cache: VirtualTokenCache, n_tokens: int, block_size: int) -> None:
    """token_cache_manger.py:149-158."""
    n_need = (n_tokens + block_size - 1) // block_size
    if n_tokens > virtual_cache.n_cache_tokens:
        more = n_need - len(virtual_cache.block_table)
        if more > 0:          # (15 decode steps of 16 grow inside the last block: nothing to allocate)
            virtual_cache.block_table += allocate_new_blocks(block_allocator, shared_cache, more)
    else:
        shared_cache.unpin(virtual_cache.block_table[n_need:])
        virtual_cache.block_table = virtual_cache.block_table[:n_need]
    virtual_cache.n_cache_tokens = n_tokens
