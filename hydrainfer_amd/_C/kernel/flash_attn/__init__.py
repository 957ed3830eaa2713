"""hydrainfer._C.kernel.flash_attn — drop-in surface
(reference stub: hydrainfer/_C/kernel/flash_attn/__init__.pyi:23-40;
CUDA original: csrc/kernel/flash_attn/flash_api.cpp:216-355).

Positional-only in practice, like the pybind original ("pybind module can't be called
by key value form", __init__.pyi:22).  Argument validation mirrors the TORCH_CHECKs of
flash_api.cpp:236-283 and raises RuntimeError (HydraHipError) in every failure case."""
import ctypes
from typing import Optional

import torch
from torch import Tensor

from hydrainfer_amd import _lib


def _i32(t: Tensor, name: str) -> Tensor:
    if t.dtype != torch.int32:
        raise _lib.HydraHipError(f"{name} must have dtype int32")
    return t if t.is_contiguous() else t.contiguous()


def mha_varlen_fwd(out: Tensor, q: Tensor, k: Tensor, v: Tensor, cu_seqlens_q: Tensor,
                   cu_seqlens_k: Tensor, block_table_: Optional[Tensor],
                   cu_block_lens: Optional[Tensor], alibi_slopes: Optional[Tensor],
                   max_seqlen_q: int, max_seqlen_k: int, softmax_scale: float, softcap: float,
                   window_size_left: int, window_size_right: int, num_splits: int) -> None:
    _lib.require_gpu(out, q, k, v, cu_seqlens_q, cu_seqlens_k, block_table_, cu_block_lens)
    if q.dtype not in (torch.float16, torch.bfloat16):
        raise _lib.HydraHipError("FlashAttention only support fp16 and bf16 data type")
    if k.dtype != q.dtype or v.dtype != q.dtype or out.dtype != q.dtype:
        raise _lib.HydraHipError("query, key, value and out must have the same dtype")
    cu_seqlens_q = _i32(cu_seqlens_q, "cu_seqlens_q")
    cu_seqlens_k = _i32(cu_seqlens_k, "cu_seqlens_k")
    for t in (q, k, v, out):
        if t.stride(-1) != 1:
            raise _lib.HydraHipError("Input tensor must have contiguous last dimension")
    if q.dim() != 3 or out.shape != q.shape:
        raise _lib.HydraHipError("q/out must be [n_tokens, n_heads, head_dim]")
    if q.stride(1) != q.size(2) or out.stride(1) != out.size(2):
        raise _lib.HydraHipError("q/out heads must be contiguous")
    if alibi_slopes is not None:
        raise _lib.HydraHipError("alibi_slopes is not supported by the MI355X implementation")
    if softcap < 0:
        raise _lib.HydraHipError("softcap must be >= 0")
    # flash_api.cpp:99-107: causal iff window (-1, 0); full iff (-1, -1); anything else is local
    # (sliding window) attention, served by the general kernel
    causal = window_size_left < 0 and window_size_right == 0
    local = not causal and (window_size_left >= 0 or window_size_right >= 0)
    # a window that covers every key is no window (flash-attn's own normalisation)
    if local and window_size_left >= max_seqlen_k and window_size_right >= max_seqlen_k:
        local = False

    paged = block_table_ is not None
    a = _lib.hx_attn_args()
    batch = cu_seqlens_q.numel() - 1
    if batch <= 0:
        raise _lib.HydraHipError("batch size must be positive")
    if cu_seqlens_k.numel() != batch + 1:
        raise _lib.HydraHipError("cu_seqlens_k must have shape [batch + 1]")
    n_heads, head_dim = q.size(1), q.size(2)
    if paged:
        if cu_block_lens is None:
            raise _lib.HydraHipError("cu_block_lens is required with block_table")
        block_table_ = _i32(block_table_, "block_table")
        cu_block_lens = _i32(cu_block_lens, "cu_block_lens")
        if k.dim() != 4 or v.shape != k.shape or k.size(3) != head_dim:
            raise _lib.HydraHipError("paged k/v must be [n_blocks, block_size, n_kv_heads, head_dim]")
        if k.size(1) % 16 != 0:
            raise _lib.HydraHipError("Paged KV cache block size must be divisible by 16")
        a.block_table = block_table_.data_ptr()
        a.cu_block_lens = cu_block_lens.data_ptr()
        a.block_size = k.size(1)
        a.k_block_stride, a.k_row_stride, a.k_head_stride = k.stride(0), k.stride(1), k.stride(2)
        a.v_block_stride, a.v_row_stride, a.v_head_stride = v.stride(0), v.stride(1), v.stride(2)
    else:
        if k.dim() != 3 or v.shape != k.shape or k.size(2) != head_dim:
            raise _lib.HydraHipError("dense k/v must be [total_k, n_kv_heads, head_dim]")
        a.block_table = None
        a.cu_block_lens = None
        a.block_size = 0
        a.k_block_stride, a.k_row_stride, a.k_head_stride = 0, k.stride(0), k.stride(1)
        a.v_block_stride, a.v_row_stride, a.v_head_stride = 0, v.stride(0), v.stride(1)
    n_kv_heads = k.size(-2)
    if head_dim % 8 != 0:
        raise _lib.HydraHipError("FlashAttention forward only supports head dimension divisible by 8")
    if head_dim > 256:
        raise _lib.HydraHipError("FlashAttention forward only supports head dimension at most 256")
    if n_heads % n_kv_heads != 0:
        raise _lib.HydraHipError("Number of heads in key/value must divide number of heads in query")

    a.out, a.q, a.k, a.v = out.data_ptr(), q.data_ptr(), k.data_ptr(), v.data_ptr()
    a.cu_seqlens_q, a.cu_seqlens_k = cu_seqlens_q.data_ptr(), cu_seqlens_k.data_ptr()
    a.batch, a.n_heads, a.n_kv_heads, a.head_dim = batch, n_heads, n_kv_heads, head_dim
    a.max_seqlen_q, a.max_seqlen_k, a.total_q = int(max_seqlen_q), int(max_seqlen_k), q.size(0)
    a.q_row_stride, a.o_row_stride = q.stride(0), out.stride(0)
    a.softmax_scale = float(softmax_scale)
    a.causal = 1 if causal else 0
    a.dtype = _lib.dtype_code(q)
    a.num_splits = int(num_splits)
    a.workspace, a.workspace_bytes = None, 0
    a.softcap = float(softcap)
    a.window_left = int(window_size_left) if local else -1
    a.window_right = int(window_size_right) if local else -1
    a.flags = _lib.HX_ATTN_LOCAL_WINDOW if local else 0

    l = _lib.lib()
    need = l.hx_mha_varlen_fwd_workspace_bytes(ctypes.byref(a))
    ws = None
    if need > 0:
        ws = torch.empty(need, dtype=torch.uint8, device=q.device)
        a.workspace, a.workspace_bytes = ws.data_ptr(), need
    with torch.cuda.device(q.device):
        _lib.check(l.hx_mha_varlen_fwd(ctypes.byref(a), _lib.current_stream()), "mha_varlen_fwd")


def decode_attention_fused(out: Tensor, q: Tensor, k_new: Tensor, v_new: Tensor, key_cache: Tensor,
                           value_cache: Tensor, positions: Tensor, cos_sin: Tensor,
                           new_cache_slots: Tensor, cu_seqlens_q: Tensor, cu_seqlens_k: Tensor,
                           block_table: Tensor, cu_block_lens: Tensor, max_seqlen_k: int,
                           softmax_scale: float, num_splits: int = 0,
                           qkv_partial: Optional[Tensor] = None, qkv_splits: int = 0,
                           rank_desc: Optional[Tensor] = None) -> None:
    """Extension: apply_rotary_pos_emb(q, k_new) + set_kv_cache(new_cache_slots, k_new, v_new) +
    mha_varlen_fwd(paged, causal) for an all-decode batch, as ONE launch.  q / k_new / v_new are
    the un-rotated projections [batch, heads, head_dim]; cu_seqlens_k already counts the new
    token.  Bit-identical to the three separate ops (q and k_new are NOT modified in place).
    rank_desc (optional, int32 [1 + batch] on the device: decode_rank / the runner's step head / the engine's host-built
    step): a big ragged batch is then laid over the CUs in length-ranked snake order — same results bit for bit."""
    _lib.require_gpu(out, q, k_new, v_new, key_cache, value_cache, positions, cos_sin, new_cache_slots,
                     cu_seqlens_q, cu_seqlens_k, block_table, cu_block_lens)
    if q.dtype not in (torch.float16, torch.bfloat16):
        raise _lib.HydraHipError("FlashAttention only support fp16 and bf16 data type")
    for t in (k_new, v_new, key_cache, value_cache, cos_sin, out):
        if t.dtype != q.dtype:
            raise _lib.HydraHipError("decode_attention_fused: dtype mismatch")
    for t, name in ((positions, "positions"), (new_cache_slots, "new_cache_slots"), (cu_seqlens_q, "cu_seqlens_q"),
                    (cu_seqlens_k, "cu_seqlens_k"), (block_table, "block_table"), (cu_block_lens, "cu_block_lens")):
        if t.dtype != torch.int32 or not t.is_contiguous():
            raise _lib.HydraHipError(f"{name} must be contiguous int32")
    if q.dim() != 3 or k_new.dim() != 3 or v_new.dim() != 3 or key_cache.dim() != 4:
        raise _lib.HydraHipError("decode_attention_fused: q/k_new/v_new [B, heads, D], caches 4-D")
    for t in (q, k_new, v_new, out):
        if t.stride(-1) != 1 or t.stride(-2) != t.size(-1):
            raise _lib.HydraHipError("decode_attention_fused: last two dims must be contiguous")
    batch, n_heads, head_dim = q.shape
    if cu_seqlens_q.numel() != batch + 1:
        raise _lib.HydraHipError("decode_attention_fused needs exactly one query token per sequence")
    k, v = key_cache, value_cache
    a = _lib.hx_attn_args()
    a.out, a.q, a.k, a.v = out.data_ptr(), q.data_ptr(), k.data_ptr(), v.data_ptr()
    a.cu_seqlens_q, a.cu_seqlens_k = cu_seqlens_q.data_ptr(), cu_seqlens_k.data_ptr()
    a.block_table, a.cu_block_lens = block_table.data_ptr(), cu_block_lens.data_ptr()
    a.batch, a.n_heads, a.n_kv_heads, a.head_dim = batch, n_heads, k.size(2), head_dim
    a.block_size, a.max_seqlen_q, a.max_seqlen_k, a.total_q = k.size(1), 1, int(max_seqlen_k), batch
    a.q_row_stride, a.o_row_stride = q.stride(0), out.stride(0)
    a.k_block_stride, a.k_row_stride, a.k_head_stride = k.stride(0), k.stride(1), k.stride(2)
    a.v_block_stride, a.v_row_stride, a.v_head_stride = v.stride(0), v.stride(1), v.stride(2)
    a.softmax_scale, a.causal, a.dtype, a.num_splits = float(softmax_scale), 1, _lib.dtype_code(q), int(num_splits)
    a.workspace, a.workspace_bytes = None, 0
    a.softcap, a.window_left, a.window_right, a.flags = 0.0, -1, -1, 0
    fz = _lib.hx_fused_decode_args()
    fz.k_new, fz.v_new = k_new.data_ptr(), v_new.data_ptr()
    fz.k_new_row_stride, fz.v_new_row_stride = k_new.stride(0), v_new.stride(0)
    fz.positions, fz.cos_sin, fz.new_cache_slots = positions.data_ptr(), cos_sin.data_ptr(), new_cache_slots.data_ptr()
    fz.rotary_dim, fz.interleaved = head_dim, 0
    if qkv_partial is not None:
        # q / k_new / v_new are then only shape carriers: the kernel reads the fp32 slabs
        n_kv = k.size(2)
        if qkv_partial.dtype != torch.float32 or qkv_partial.numel() < qkv_splits * batch * (n_heads + 2 * n_kv) * head_dim:
            raise _lib.HydraHipError("decode_attention_fused: qkv_partial must be float32 [splits, batch, (H+2HK)*D]")
        fz.qkv_partial, fz.qkv_splits = qkv_partial.data_ptr(), int(qkv_splits)
    else:
        fz.qkv_partial, fz.qkv_splits = None, 0
    fz.rank_desc = None
    if rank_desc is not None:
        if rank_desc.dtype != torch.int32 or not rank_desc.is_contiguous() or not rank_desc.is_cuda or rank_desc.numel() < batch + 1:
            raise _lib.HydraHipError("decode_attention_fused: rank_desc must be a contiguous int32 device tensor [1 + batch]")
        fz.rank_desc = rank_desc.data_ptr()
    l = _lib.lib()
    need = l.hx_mha_varlen_fwd_workspace_bytes(ctypes.byref(a))
    if need > 0:
        ws = torch.empty(need, dtype=torch.uint8, device=q.device)
        a.workspace, a.workspace_bytes = ws.data_ptr(), need
    with torch.cuda.device(q.device):
        _lib.check(l.hx_decode_attention_fused(ctypes.byref(a), ctypes.byref(fz), _lib.current_stream()),
                   "decode_attention_fused")


def decode_rank(cu_seqlens_k: Tensor, out: Optional[Tensor] = None) -> Tensor:
    """Extension: the RANK DESCRIPTOR of a decode batch (hx_decode_rank; include/hydra_hip.h, hx_fused_decode_args) —
    int32 [1 + batch]: [0] = 1 when some sequence holds more than 1.125 x the mean + 16 keys, [1 + r] = the sequence with
    the r-th most keys (ties: the lower number first).  A batch of more than 256 sequences is declared even."""
    _lib.require_gpu(cu_seqlens_k, out)
    if cu_seqlens_k.dtype != torch.int32 or not cu_seqlens_k.is_contiguous() or cu_seqlens_k.numel() < 2:
        raise _lib.HydraHipError("decode_rank: cu_seqlens_k must be contiguous int32 [batch + 1]")
    batch = cu_seqlens_k.numel() - 1
    if out is None:
        out = torch.empty(batch + 1, dtype=torch.int32, device=cu_seqlens_k.device)
    elif out.dtype != torch.int32 or not out.is_contiguous() or out.numel() < batch + 1:
        raise _lib.HydraHipError("decode_rank: out must be contiguous int32 [1 + batch]")
    with torch.cuda.device(cu_seqlens_k.device):
        _lib.check(_lib.lib().hx_decode_rank(cu_seqlens_k.data_ptr(), batch, out.data_ptr(), _lib.current_stream()), "decode_rank")
    return out
