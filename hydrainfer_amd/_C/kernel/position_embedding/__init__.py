"""hydrainfer._C.kernel.position_embedding — drop-in surface
(reference stub: hydrainfer/_C/kernel/position_embedding/__init__.pyi:5-11;
CUDA original: csrc/kernel/position_embedding/rope.cu:82-117)."""
from torch import Tensor

from hydrainfer_amd import _lib


def apply_rotary_pos_emb(query: Tensor, key: Tensor, positions: Tensor, cos_sin: Tensor,
                         rotary_dim: int, interleaved: bool) -> None:
    _lib.require_gpu(query, key, positions, cos_sin)
    if query.dim() != 3 or key.dim() != 3:
        raise _lib.HydraHipError("apply_rotary_pos_emb: query/key must be [n_tokens, heads, head_dim]")
    # rope.cu:90-91: last two dims contiguous
    for t in (query, key):
        if t.stride(-1) != 1 or t.stride(-2) != t.size(-1):
            raise _lib.HydraHipError("apply_rotary_pos_emb: last two dims must be contiguous")
    if positions.dtype.itemsize != 4 or positions.dtype.is_floating_point:
        raise _lib.HydraHipError("apply_rotary_pos_emb: positions must be int32")
    if not (query.dtype == key.dtype == cos_sin.dtype):
        raise _lib.HydraHipError("apply_rotary_pos_emb: query/key/cos_sin dtype mismatch")
    if not cos_sin.is_contiguous() or cos_sin.numel() % rotary_dim != 0:
        raise _lib.HydraHipError("apply_rotary_pos_emb: cos_sin must be contiguous [max_pos, 2, rotary_dim/2]")
    if key.size(0) != query.size(0) or key.size(2) != query.size(2) or positions.numel() != query.size(0):
        raise _lib.HydraHipError("apply_rotary_pos_emb: shape mismatch")
    if not positions.is_contiguous():
        positions = positions.contiguous()
    _lib.check(_lib.lib().hx_apply_rotary_pos_emb(
        query.data_ptr(), key.data_ptr(), positions.data_ptr(), cos_sin.data_ptr(),
        query.size(0), query.size(1), key.size(1), query.size(2), int(rotary_dim),
        query.stride(0), key.stride(0), 1 if interleaved else 0, _lib.dtype_code(query),
        _lib.current_stream()), "apply_rotary_pos_emb")


def rope_set_kv_cache(query: Tensor, key: Tensor, value: Tensor, positions: Tensor, cos_sin: Tensor,
                      rotary_dim: int, slot_ids: Tensor, key_cache: Tensor, value_cache: Tensor) -> None:
    """Extension: apply_rotary_pos_emb(query, key, ..., interleaved=False) followed by
    set_kv_cache(slot_ids, key, value, key_cache, value_cache), as one launch."""
    _lib.require_gpu(query, key, value, positions, cos_sin, slot_ids, key_cache, value_cache)
    for t in (query, key, value):
        if t.dim() != 3 or t.stride(-1) != 1 or t.stride(-2) != t.size(-1):
            raise _lib.HydraHipError("rope_set_kv_cache: q/k/v must be [n, heads, head_dim], last two dims contiguous")
    for c in (key_cache, value_cache):
        if c.dim() != 4 or c.stride(-1) != 1 or c.stride(-2) != c.size(-1) or c.stride(-3) != c.size(-1) * c.size(-2):
            raise _lib.HydraHipError("rope_set_kv_cache: cache rows inside a block must be contiguous")
    if positions.dtype.itemsize != 4 or slot_ids.dtype.itemsize != 4:
        raise _lib.HydraHipError("rope_set_kv_cache: positions / slot_ids must be int32")
    if not (query.dtype == key.dtype == value.dtype == cos_sin.dtype == key_cache.dtype == value_cache.dtype):
        raise _lib.HydraHipError("rope_set_kv_cache: dtype mismatch")
    _lib.check(_lib.lib().hx_rope_set_kv_cache(
        query.data_ptr(), key.data_ptr(), value.data_ptr(), positions.contiguous().data_ptr(),
        cos_sin.data_ptr(), slot_ids.contiguous().data_ptr(), key_cache.data_ptr(), value_cache.data_ptr(),
        query.size(0), query.size(1), key.size(1), query.size(2), int(rotary_dim), query.stride(0),
        key.stride(0), value.stride(0), key_cache.size(1), key_cache.stride(0), value_cache.stride(0),
        _lib.dtype_code(query), _lib.current_stream()), "rope_set_kv_cache")
