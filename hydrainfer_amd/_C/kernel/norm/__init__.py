"""hydrainfer._C.kernel.norm — drop-in surface
(reference stub: hydrainfer/_C/kernel/norm/__init__.pyi:4-9;
CUDA original: csrc/kernel/norm/rms_norm.cu:43-63).  bf16 is accepted (extension: the
reference dispatch, csrc/kernel/dispatch.h:12-28, is fp32/fp16 only)."""
import ctypes
from dataclasses import dataclass
from typing import Optional

import torch
from torch import Tensor

from hydrainfer_amd import _lib


def rms_norm(out: Tensor, input: Tensor, weight: Tensor, epsilon: float) -> None:
    _lib.require_gpu(out, input, weight)
    if input.dim() != 2 or out.shape != input.shape:
        raise _lib.HydraHipError("rms_norm: input/out must be 2-D with equal shapes")
    if not input.is_contiguous() or not out.is_contiguous() or not weight.is_contiguous():
        raise _lib.HydraHipError("rms_norm: tensors must be contiguous")
    if weight.numel() != input.size(1) or not (out.dtype == input.dtype == weight.dtype):
        raise _lib.HydraHipError("rms_norm: weight shape / dtype mismatch")
    _lib.check(_lib.lib().hx_rms_norm(
        out.data_ptr(), input.data_ptr(), weight.data_ptr(), float(epsilon), input.size(0),
        input.size(1), _lib.dtype_code(input), _lib.current_stream()), "rms_norm")


def add_rms_norm(out: Tensor, residual: Tensor, x: Tensor, weight: Tensor, epsilon: float) -> None:
    """Extension: residual += x (in place), out = rms_norm(residual)."""
    _lib.require_gpu(out, residual, x, weight)
    if x.dim() != 2 or out.shape != x.shape or residual.shape != x.shape:
        raise _lib.HydraHipError("add_rms_norm: shapes must match and be 2-D")
    for t in (out, residual, x, weight):
        if not t.is_contiguous():
            raise _lib.HydraHipError("add_rms_norm: tensors must be contiguous")
    if weight.numel() != x.size(1) or not (out.dtype == x.dtype == weight.dtype == residual.dtype):
        raise _lib.HydraHipError("add_rms_norm: weight shape / dtype mismatch")
    _lib.check(_lib.lib().hx_add_rms_norm(
        out.data_ptr(), residual.data_ptr(), x.data_ptr(), weight.data_ptr(), float(epsilon),
        x.size(0), x.size(1), _lib.dtype_code(x), _lib.current_stream()), "add_rms_norm")


def add_layer_norm(out: Tensor, residual: Tensor, x: Optional[Tensor], weight: Tensor, bias: Tensor,
                   epsilon: float) -> None:
    """Extension (vision tower): residual += x (in place, one T rounding); out = layer_norm(residual) * weight + bias —
    `h = h + y; x = nn.LayerNorm(h)` of the CLIP encoder layer in one pass.  x=None: plain layer norm of `residual`."""
    _lib.require_gpu(out, residual, weight, bias)
    hidden = residual.shape[-1]
    tensors = [out, residual, weight, bias] + ([x] if x is not None else [])
    if out.shape != residual.shape or (x is not None and x.shape != residual.shape):
        raise _lib.HydraHipError("add_layer_norm: shapes must match")
    for t in tensors:
        if not t.is_contiguous() or t.dtype != residual.dtype:
            raise _lib.HydraHipError("add_layer_norm: tensors must be contiguous and of one dtype")
    if weight.numel() != hidden or bias.numel() != hidden:
        raise _lib.HydraHipError("add_layer_norm: weight / bias shape mismatch")
    if x is not None:
        _lib.require_gpu(x)
    _lib.check(_lib.lib().hx_add_layer_norm(
        out.data_ptr(), residual.data_ptr(), x.data_ptr() if x is not None else None, weight.data_ptr(), bias.data_ptr(),
        float(epsilon), residual.numel() // hidden, hidden, _lib.dtype_code(residual), _lib.current_stream()), "add_layer_norm")


def add_rms_norm_slabs(out: Tensor, residual: Tensor, partial: Tensor, n_splits: int, weight: Tensor,
                       epsilon: float, fragment_major: bool = False) -> None:
    """Extension: x = (T) sum of the n_splits fp32 slabs in `partial` ([n_splits, rows, hidden]);
    residual += x (in place); out = rms_norm(residual).  Bit-identical to reduce + add_rms_norm.
    fragment_major: `out` (at least gemm.fragment_major_elems(rows, hidden) elements) receives the same
    values in the MFMA-B-fragment order the activations-in-registers GEMM reads (hydra_hip.h)."""
    _lib.require_gpu(out, residual, partial, weight)
    rows, hidden = residual.shape
    if partial.dtype != torch.float32 or partial.numel() < n_splits * rows * hidden:
        raise _lib.HydraHipError("add_rms_norm_slabs: partial must be float32 [n_splits, rows, hidden]")
    for t in (out, residual, weight):
        if not t.is_contiguous() or t.dtype != residual.dtype:
            raise _lib.HydraHipError("add_rms_norm_slabs: tensors must be contiguous and of one dtype")
    if fragment_major and (hidden % 32 or out.numel() < (rows + 15) // 16 * 16 * hidden):
        raise _lib.HydraHipError("add_rms_norm_slabs: fragment-major output needs hidden % 32 == 0 and ceil16(rows) * hidden elements")
    _lib.check(_lib.lib().hx_add_rms_norm_slabs_ex(out.data_ptr(), residual.data_ptr(), partial.data_ptr(),
                                                   int(n_splits), weight.data_ptr(), float(epsilon), rows,
                                                   hidden, _lib.dtype_code(residual), 1 if fragment_major else 0,
                                                   _lib.current_stream()),
               "add_rms_norm_slabs")


def embed_rms_norm(ids: Tensor, table: Tensor, weight: Tensor, epsilon: float):
    """Extension: (h, x) with h = table[ids], x = rms_norm(h) * weight — one launch, bit-identical to
    torch.nn.functional.embedding + rms_norm for ids inside the vocabulary.  ids int32 / int64 [rows]; fp16 / bf16;
    hidden % 8 == 0, <= 8192.  DIVERGENCE from torch: an id outside [0, vocab) is CLAMPED by the kernel where
    torch.nn.functional.embedding (the reference path) raises — callers validate host-provided ids themselves
    (engine/graph_decode.py does, before the launch); ids produced by hx_argmax_rows are in range by construction."""
    _lib.require_gpu(ids, table, weight)
    if ids.dim() != 1 or ids.dtype not in (torch.int32, torch.int64) or not ids.is_contiguous():
        raise _lib.HydraHipError("embed_rms_norm: ids must be contiguous int32 / int64 [rows]")
    if table.dim() != 2 or not table.is_contiguous() or weight.dtype != table.dtype or weight.numel() != table.shape[1]:
        raise _lib.HydraHipError("embed_rms_norm: table [vocab, hidden] contiguous, weight [hidden] of the same dtype")
    rows, (vocab, hidden) = ids.numel(), table.shape
    h = torch.empty((rows, hidden), dtype=table.dtype, device=table.device)
    x = torch.empty_like(h)
    _lib.check(_lib.lib().hx_embed_rms_norm(h.data_ptr(), x.data_ptr(), ids.data_ptr(), 1 if ids.dtype == torch.int64 else 0,
                                            table.data_ptr(), weight.data_ptr(), float(epsilon), rows, hidden, vocab,
                                            _lib.dtype_code(table), _lib.current_stream()), "embed_rms_norm")
    return h, x


@dataclass
class StepHead:
    """What a decode step does before its first GEMM besides the embedding gather + first norm, folded into that
    launch (hx_decode_step_head): the metadata advance of a device-resident decode loop (model/runner.py) and / or the
    look-ahead feed of the engine's decoder (engine/graph_decode.py)."""
    # hx_decode_advance arguments (all or none)
    positions: Optional[Tensor] = None
    kv_lens: Optional[Tensor] = None
    cu_seqlens_k: Optional[Tensor] = None
    new_cache_slots: Optional[Tensor] = None
    block_table: Optional[Tensor] = None
    cu_block_lens: Optional[Tensor] = None
    batch: int = 0
    block_size: int = 16
    stride: int = 1
    rank_desc: Optional[Tensor] = None      # int32 [1 + batch]: receives the advanced batch's rank descriptor (attn_decode.hip)
    # hx_decode_feed_ids arguments (both or neither)
    feed_src: Optional[Tensor] = None       # int32 [rows]
    feed_prev: Optional[Tensor] = None      # int64: the previous launch's samples


def decode_step_head(ids: Tensor, table: Tensor, weight: Tensor, epsilon: float, zero: Optional[Tensor] = None,
                     head: Optional[StepHead] = None):
    """Extension: embed_rms_norm(ids, table, weight) + memset_zero(zero) + the StepHead's advance / feed, ONE launch
    (3-4 launches of ~4.6 us each in round 3's decode-step timeline).  Returns (h, x) like embed_rms_norm."""
    _lib.require_gpu(ids, table, weight, zero)
    if ids.dim() != 1 or ids.dtype not in (torch.int32, torch.int64) or not ids.is_contiguous():
        raise _lib.HydraHipError("decode_step_head: ids must be contiguous int32 / int64 [rows]")
    if table.dim() != 2 or not table.is_contiguous() or weight.dtype != table.dtype or weight.numel() != table.shape[1]:
        raise _lib.HydraHipError("decode_step_head: table [vocab, hidden] contiguous, weight [hidden] of the same dtype")
    rows, (vocab, hidden) = ids.numel(), table.shape
    h = torch.empty((rows, hidden), dtype=table.dtype, device=table.device)
    x = torch.empty_like(h)
    a = _lib.hx_step_head_args()
    a.h_out, a.x_out, a.ids, a.table, a.weight = h.data_ptr(), x.data_ptr(), ids.data_ptr(), table.data_ptr(), weight.data_ptr()
    a.rows, a.hidden, a.vocab, a.epsilon = rows, hidden, vocab, float(epsilon)
    a.ids_are_int64, a.dtype = (1 if ids.dtype == torch.int64 else 0), _lib.dtype_code(table)
    if zero is not None:
        if not zero.is_contiguous() or (zero.numel() * zero.element_size()) % 4:
            raise _lib.HydraHipError("decode_step_head: the area to zero must be contiguous, a multiple of 4 bytes")
        a.zero_ptr, a.zero_bytes = zero.data_ptr(), zero.numel() * zero.element_size()
    if head is not None:
        if head.batch > 0:
            adv = (head.positions, head.kv_lens, head.cu_seqlens_k, head.new_cache_slots, head.block_table, head.cu_block_lens)
            if any(t is None or t.dtype != torch.int32 or not t.is_contiguous() or not t.is_cuda for t in adv):
                raise _lib.HydraHipError("decode_step_head: advance arguments must be contiguous int32 device tensors")
            if (head.positions.numel() < head.batch or head.kv_lens.numel() < head.batch or head.new_cache_slots.numel() < head.batch
                    or head.cu_seqlens_k.numel() < head.batch + 1 or head.cu_block_lens.numel() < head.batch + 1):
                raise _lib.HydraHipError("decode_step_head: advance arguments shorter than the batch")
            (a.positions, a.kv_lens, a.cu_seqlens_k, a.new_cache_slots, a.block_table, a.cu_block_lens) = (t.data_ptr() for t in adv)
            a.batch, a.block_size, a.stride = int(head.batch), int(head.block_size), int(head.stride)
            if head.rank_desc is not None:
                rd = head.rank_desc
                if rd.dtype != torch.int32 or not rd.is_contiguous() or not rd.is_cuda or rd.numel() < head.batch + 1:
                    raise _lib.HydraHipError("decode_step_head: rank_desc must be a contiguous int32 device tensor [1 + batch]")
                a.rank_desc = rd.data_ptr()
        if (head.feed_src is None) != (head.feed_prev is None):
            raise _lib.HydraHipError("decode_step_head: feed_src and feed_prev come together")
        if head.feed_src is not None:
            if (head.feed_src.dtype != torch.int32 or head.feed_prev.dtype != torch.int64 or head.feed_src.numel() < rows
                    or not head.feed_src.is_contiguous() or not head.feed_prev.is_contiguous()):
                raise _lib.HydraHipError("decode_step_head: feed_src int32 [rows], feed_prev int64, contiguous")
            a.feed_src, a.feed_prev = head.feed_src.data_ptr(), head.feed_prev.data_ptr()
    _lib.check(_lib.lib().hx_decode_step_head(ctypes.byref(a), _lib.current_stream()), "decode_step_head")
    return h, x


def embed_rms_norm_supported(ids: Tensor, table: Tensor) -> bool:
    return (ids.is_cuda and ids.dim() == 1 and ids.dtype in (torch.int32, torch.int64) and ids.is_contiguous()
            and table.dtype in (torch.float16, torch.bfloat16) and table.is_contiguous()
            and table.shape[1] % 8 == 0 and table.shape[1] <= 8192)


def argmax_rows(logits: Tensor, out: Optional[Tensor] = None) -> Tensor:
    """Extension: torch.argmax(logits, dim=-1) for fp16 / bf16 [rows, n] (greedy sampling), one small launch;
    `out`: an int64 [rows] tensor to write into (a decode loop's next-input buffer)."""
    _lib.require_gpu(logits)
    if logits.dim() != 2 or logits.stride(1) != 1 or logits.dtype not in (torch.float16, torch.bfloat16):
        raise _lib.HydraHipError("argmax_rows: logits must be fp16 / bf16 [rows, n] with contiguous rows")
    if out is None:
        out = torch.empty(logits.shape[0], dtype=torch.int64, device=logits.device)
    elif out.dtype != torch.int64 or out.shape != (logits.shape[0],) or not out.is_contiguous() or out.device != logits.device:
        raise _lib.HydraHipError("argmax_rows: out must be a contiguous int64 [rows] tensor on the logits' device")
    _lib.check(_lib.lib().hx_argmax_rows(out.data_ptr(), logits.data_ptr(), logits.shape[0], logits.shape[1],
                                         logits.stride(0), _lib.dtype_code(logits), _lib.current_stream()), "argmax_rows")
    return out
