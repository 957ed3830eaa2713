"""Extension op (no counterpart in hydrainfer._C): decode-batch linear layer on the
weight-streaming HIP kernel (csrc/gemm_skinny.hip)."""
from typing import Optional

import torch
from torch import Tensor

from hydrainfer_amd import _lib


def supported(x: Tensor, weight: Tensor) -> bool:
    M, K = x.shape
    N = weight.shape[0]
    return (x.dtype in (torch.float16, torch.bfloat16) and 1 <= M <= 64 and N % 16 == 0 and K % 256 == 0
            and x.stride(1) == 1 and weight.stride(1) == 1 and x.stride(0) % 8 == 0 and weight.stride(0) % 8 == 0)


def linear_decode(x: Tensor, weight: Tensor, out: Optional[Tensor] = None) -> Tensor:
    """out[M, N] = x[M, K] @ weight[N, K]^T, M <= 64."""
    _lib.require_gpu(x, weight)
    if x.dim() != 2 or weight.dim() != 2 or x.shape[1] != weight.shape[1] or x.dtype != weight.dtype:
        raise _lib.HydraHipError("linear_decode: x [M, K], weight [N, K], same dtype")
    if not supported(x, weight):
        raise _lib.HydraHipError("linear_decode: needs M <= 64, N % 16 == 0, K % 256 == 0, fp16/bf16, contiguous rows")
    M, K = x.shape
    N = weight.shape[0]
    if out is None:
        out = torch.empty((M, N), dtype=x.dtype, device=x.device)
    l = _lib.lib()
    nbytes = l.hx_linear_decode_workspace_bytes(M, N, K)
    ws = torch.empty(nbytes, dtype=torch.uint8, device=x.device)
    _lib.check(l.hx_linear_decode(out.data_ptr(), x.data_ptr(), weight.data_ptr(), M, N, K, x.stride(0),
                                  weight.stride(0), out.stride(0), ws.data_ptr(), nbytes,
                                  _lib.dtype_code(x), _lib.current_stream()), "linear_decode")
    return out


def workspace_floats(M: int, N: int, K: int) -> int:
    return _lib.lib().hx_linear_decode_workspace_bytes(M, N, K) // 4


def linear_decode_partial(x: Tensor, weight: Tensor, partial: Tensor) -> int:
    """GEMM only: fp32 split-K slabs [n_splits, M, N] are written into `partial` (a float32
    buffer of at least workspace_floats(M, N, K) elements) for a fused consumer.  Returns n_splits."""
    _lib.require_gpu(x, weight, partial)
    if not supported(x, weight) or x.dtype != weight.dtype:
        raise _lib.HydraHipError("linear_decode_partial: needs M <= 64, N % 16 == 0, K % 256 == 0, fp16/bf16")
    if partial.dtype != torch.float32 or not partial.is_contiguous():
        raise _lib.HydraHipError("linear_decode_partial: partial must be contiguous float32")
    M, K = x.shape
    N = weight.shape[0]
    rc = _lib.lib().hx_linear_decode_partial(partial.data_ptr(), x.data_ptr(), weight.data_ptr(), M, N, K,
                                             x.stride(0), weight.stride(0), partial.numel() * 4,
                                             _lib.dtype_code(x), _lib.current_stream())
    if rc < 0:
        _lib.check(rc, "linear_decode_partial")
    return rc
