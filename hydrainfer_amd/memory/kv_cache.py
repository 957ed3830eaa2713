"""KVCache — host-side mirror of hydrainfer/memory/kv_cache.py:14-56 (HIP only)."""
from torch import Tensor

from hydrainfer_amd._C.kernel.kv_cache_kernels import set_kv_cache as set_kv_cache_kernel
from hydrainfer_amd.memory.token_cache import TokenCache


class KVCache:
    def __init__(self, key_cache: Tensor, value_cache: Tensor):
        assert key_cache.dim() == 4, f"key cache dim should be 4 but got shape {key_cache.shape}"
        assert value_cache.shape == key_cache.shape
        self.key_cache = key_cache
        self.value_cache = value_cache
        self.dtype = key_cache.dtype
        self.device = key_cache.device
        self.block_size = key_cache.shape[1]

    def get_kv_cache(self):
        return (self.key_cache, self.value_cache)

    def set_kv_cache(self, slot_ids: Tensor, keys: Tensor, values: Tensor) -> None:
        assert slot_ids.shape[0] == keys.shape[0], f"{slot_ids.shape} {keys.shape}"
        assert slot_ids.shape[0] == values.shape[0], f"{slot_ids.shape} {values.shape}"
        set_kv_cache_kernel(slot_ids, keys, values, self.key_cache, self.value_cache)

    @classmethod
    def from_token_cache(cls, token_cache: TokenCache) -> "KVCache":
        tensors = token_cache.get_caches()
        assert len(tensors) == 2
        return cls(tensors[0], tensors[1])
