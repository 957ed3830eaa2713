"""TokenCache / VirtualTokenCache — host-side mirror of hydrainfer/memory/token_cache.py:15-66.
The scatter runs on the HIP kernel only; there is no index_put fallback."""
from dataclasses import dataclass, field
from typing import List, Optional

from torch import Tensor

from hydrainfer_amd._C.kernel.cache_kernels import set_image_cache


class TokenCache:
    """caches: [key_cache, value_cache] or [image_embed_cache], each
    (n_blocks, block_size, n_heads, head_size)."""

    def __init__(self, caches: List[Tensor]):
        for cache in caches:
            assert cache.dim() == 4, f"cache dim should be 4 but got shape {cache.shape}"
            assert cache.shape == caches[0].shape
            assert cache.dtype == caches[0].dtype
            assert cache.device == caches[0].device
        self.caches = caches
        self.block_size = caches[0].shape[1]
        self.dtype = caches[0].dtype
        self.device = caches[0].device

    def get_caches(self) -> List[Tensor]:
        return self.caches

    def set_caches(self, slot_ids: Tensor, values: List[Tensor]) -> None:
        assert slot_ids.dim() == 1
        for value in values:
            assert value.dim() == 3
            assert slot_ids.shape[0] == value.shape[0]
            assert slot_ids.device == value.device
        for cache, value in zip(self.caches, values):
            set_image_cache(slot_ids, value, cache)


@dataclass
class VirtualTokenCache:
    vid: int
    n_blocks_of_cache_manager: int
    n_cache_tokens: int = 0
    block_table: List[int] = field(default_factory=list)
    memory_handle: Optional[List[int]] = None  # IPC handle bytes
    rank: int = -1
