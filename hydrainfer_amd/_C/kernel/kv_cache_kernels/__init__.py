"""hydrainfer._C.kernel.kv_cache_kernels — drop-in surface
(reference stub: hydrainfer/_C/kernel/kv_cache_kernels/__init__.pyi:5-10;
CUDA original: csrc/kernel/kv_cache_kernels/kv_cache_kernels.cu:60-95)."""
from torch import Tensor

from hydrainfer_amd import _lib


def set_kv_cache(slot_ids: Tensor, keys: Tensor, values: Tensor, key_cache: Tensor,
                 value_cache: Tensor) -> None:
    _lib.require_gpu(slot_ids, keys, values, key_cache, value_cache)
    if slot_ids.dtype.itemsize != 4 or slot_ids.dtype.is_floating_point:
        raise _lib.HydraHipError("set_kv_cache: slot_ids must be int32")
    if keys.dim() != 3 or values.dim() != 3 or key_cache.dim() != 4 or value_cache.dim() != 4:
        raise _lib.HydraHipError("set_kv_cache: keys/values must be 3-D, caches 4-D")
    # keys and values must be contiguous in (n_kv_heads, head_dim) — kv_cache_kernels.cu:67-68
    for t in (keys, values):
        if t.stride(-1) != 1 or t.stride(-2) != t.size(-1):
            raise _lib.HydraHipError("set_kv_cache: last two dims of keys/values must be contiguous")
    for c in (key_cache, value_cache):
        if c.stride(-1) != 1 or c.stride(-2) != c.size(-1) or c.stride(-3) != c.size(-1) * c.size(-2):
            raise _lib.HydraHipError("set_kv_cache: cache rows inside a block must be contiguous")
    if not (keys.dtype == values.dtype == key_cache.dtype == value_cache.dtype):
        raise _lib.HydraHipError("set_kv_cache: dtype mismatch")
    if not slot_ids.is_contiguous():
        slot_ids = slot_ids.contiguous()
    n_tokens, n_kv_heads, head_dim = keys.shape
    _lib.check(_lib.lib().hx_set_kv_cache(
        slot_ids.data_ptr(), keys.data_ptr(), values.data_ptr(), key_cache.data_ptr(),
        value_cache.data_ptr(), n_tokens, n_kv_heads, head_dim, key_cache.size(1),
        keys.stride(0), values.stride(0), key_cache.stride(0), value_cache.stride(0),
        _lib.dtype_code(keys), _lib.current_stream()), "set_kv_cache")
