"""hydrainfer._C.kernel.norm — drop-in surface
(reference stub: hydrainfer/_C/kernel/norm/__init__.pyi:4-9;
CUDA original: csrc/kernel/norm/rms_norm.cu:43-63).  bf16 is accepted (extension: the
reference dispatch, csrc/kernel/dispatch.h:12-28, is fp32/fp16 only)."""
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
