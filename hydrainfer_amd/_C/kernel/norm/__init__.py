"""hydrainfer._C.kernel.norm — drop-in surface
(reference stub: hydrainfer/_C/kernel/norm/__init__.pyi:4-9;
CUDA original: csrc/kernel/norm/rms_norm.cu:43-63).  bf16 is accepted (extension: the
reference dispatch, csrc/kernel/dispatch.h:12-28, is fp32/fp16 only)."""
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
