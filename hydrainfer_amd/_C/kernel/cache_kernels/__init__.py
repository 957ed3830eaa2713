"""hydrainfer._C.kernel.cache_kernels — drop-in surface
(reference stub: hydrainfer/_C/kernel/cache_kernels/__init__.pyi:4-7;
CUDA original: csrc/kernel/cache_kernels/cache_kernels.cu:55-83)."""
from torch import Tensor

from hydrainfer_amd import _lib


def set_image_cache(slot_ids: Tensor, image_tokens: Tensor, image_cache: Tensor) -> None:
    _lib.require_gpu(slot_ids, image_tokens, image_cache)
    if slot_ids.dtype.itemsize != 4 or slot_ids.dtype.is_floating_point:
        raise _lib.HydraHipError("set_image_cache: slot_ids must be int32")
    if image_tokens.dim() != 3 or image_cache.dim() != 4:
        raise _lib.HydraHipError("set_image_cache: image_tokens must be 3-D, image_cache 4-D")
    if image_tokens.stride(-1) != 1 or image_tokens.stride(-2) != image_tokens.size(-1):
        raise _lib.HydraHipError("set_image_cache: last two dims of image_tokens must be contiguous")
    c = image_cache
    if c.stride(-1) != 1 or c.stride(-2) != c.size(-1) or c.stride(-3) != c.size(-1) * c.size(-2):
        raise _lib.HydraHipError("set_image_cache: cache rows inside a block must be contiguous")
    if image_tokens.dtype != image_cache.dtype:
        raise _lib.HydraHipError("set_image_cache: dtype mismatch")
    if not slot_ids.is_contiguous():
        slot_ids = slot_ids.contiguous()
    n_tokens, n_heads, head_dim = image_tokens.shape
    _lib.check(_lib.lib().hx_set_image_cache(
        slot_ids.data_ptr(), image_tokens.data_ptr(), image_cache.data_ptr(), n_tokens, n_heads,
        head_dim, image_cache.size(1), image_tokens.stride(0), image_cache.stride(0),
        _lib.dtype_code(image_tokens), _lib.current_stream()), "set_image_cache")
