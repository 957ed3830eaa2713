"""hydrainfer._C.kernel.activation — drop-in surface
(reference stub: hydrainfer/_C/kernel/activation/__init__.pyi:3-4;
CUDA original: csrc/kernel/activation/activation.cu:35-56)."""
import torch
from torch import Tensor

from hydrainfer_amd import _lib


def silu(input: Tensor) -> Tensor:
    _lib.require_gpu(input)
    if input.dim() != 2 or input.stride(1) != 1:
        raise _lib.HydraHipError("silu: input must be 2-D with a contiguous last dimension")
    out = torch.empty(input.shape, dtype=input.dtype, device=input.device)
    _lib.check(_lib.lib().hx_silu(
        out.data_ptr(), input.data_ptr(), input.size(0), input.size(1), input.stride(0),
        _lib.dtype_code(input), _lib.current_stream()), "silu")
    return out


def quick_gelu(input: Tensor) -> Tensor:
    """Extension: x * sigmoid(1.702 x) in one pass with the three T roundings of the reference's three torch ops
    (hydrainfer/layer/activation.py:17-22, the CLIP MLP's activation)."""
    _lib.require_gpu(input)
    x = input.reshape(-1, input.shape[-1])
    if x.stride(1) != 1:
        raise _lib.HydraHipError("quick_gelu: input must have a contiguous last dimension")
    out = torch.empty(x.shape, dtype=x.dtype, device=x.device)
    _lib.check(_lib.lib().hx_quick_gelu(out.data_ptr(), x.data_ptr(), x.size(0), x.size(1), x.stride(0),
                                        _lib.dtype_code(x), _lib.current_stream()), "quick_gelu")
    return out.view(input.shape)


def silu_and_mul(gate: Tensor, up: Tensor) -> Tensor:
    """Extension: (T)silu(gate) * up in one pass (model_forward.py:36 fused)."""
    _lib.require_gpu(gate, up)
    if gate.dim() != 2 or gate.shape != up.shape or gate.stride(1) != 1 or up.stride(1) != 1:
        raise _lib.HydraHipError("silu_and_mul: gate/up must be 2-D, same shape, contiguous last dim")
    if gate.dtype != up.dtype:
        raise _lib.HydraHipError("silu_and_mul: dtype mismatch")
    out = torch.empty(gate.shape, dtype=gate.dtype, device=gate.device)
    _lib.check(_lib.lib().hx_silu_and_mul(
        out.data_ptr(), gate.data_ptr(), up.data_ptr(), gate.size(0), gate.size(1),
        gate.stride(0), up.stride(0), _lib.dtype_code(gate), _lib.current_stream()), "silu_and_mul")
    return out


def silu_and_mul_slabs(partial: Tensor, n_splits: int, rows: int, inter: int, dtype: torch.dtype,
                       fragment_major: bool = False) -> Tensor:
    """Extension: gate|up = (T) sum of the fp32 slabs [n_splits, rows, 2*inter]; returns
    (T)silu(gate) * up.  Bit-identical to reduce + silu_and_mul.  fragment_major: the result is a flat
    tensor of ceil16(rows) * inter elements in the order the activations-in-registers GEMM reads."""
    _lib.require_gpu(partial)
    if partial.dtype != torch.float32 or partial.numel() < n_splits * rows * 2 * inter:
        raise _lib.HydraHipError("silu_and_mul_slabs: partial must be float32 [n_splits, rows, 2*inter]")
    if fragment_major:
        if inter % 32:
            raise _lib.HydraHipError("silu_and_mul_slabs: fragment-major output needs inter % 32 == 0")
        out = torch.empty((rows + 15) // 16 * 16 * inter, dtype=dtype, device=partial.device)
    else:
        out = torch.empty((rows, inter), dtype=dtype, device=partial.device)
    code = {torch.float16: _lib.HX_F16, torch.bfloat16: _lib.HX_BF16}.get(dtype)
    if code is None:
        raise _lib.HydraHipError("silu_and_mul_slabs: fp16 / bf16 only")
    _lib.check(_lib.lib().hx_silu_and_mul_slabs_ex(out.data_ptr(), partial.data_ptr(), int(n_splits), rows, inter,
                                                   code, 1 if fragment_major else 0, _lib.current_stream()),
               "silu_and_mul_slabs")
    return out
