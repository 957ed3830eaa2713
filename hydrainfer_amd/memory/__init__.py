from .block_allocator import BlockAllocator, BlockAllocatorMetrics
from .shared_cache import SharedBlock, SharedCache, SharedCacheConfig, compute_block_hash, compute_hash, compute_image_hash


def __getattr__(name):
    # GPU-facing classes import the HIP shims lazily so pure host logic stays importable
    # in tooling that only needs the integer paths.
    import importlib
    for mod in ("token_cache", "kv_cache", "communication", "token_cache_manger"):
        m = importlib.import_module(f"hydrainfer_amd.memory.{mod}")
        if hasattr(m, name):
            return getattr(m, name)
    raise AttributeError(name)
