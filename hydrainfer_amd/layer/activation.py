"""Activations — host-side mirror of hydrainfer/layer/activation.py:24-68 (HIP kernels only)."""
from torch import Tensor, nn

from hydrainfer_amd._C.kernel.activation import silu as silu_kernel
from hydrainfer_amd._C.kernel.activation import silu_and_mul as silu_and_mul_kernel


def silu(h: Tensor) -> Tensor:
    return silu_kernel(h)


class Silu(nn.Module):
    def forward(self, h: Tensor) -> Tensor:
        return silu(h)


class SiluAndMul(nn.Module):
    """x (n_tokens, 2*hidden) -> silu(x[:, :hidden]) * x[:, hidden:]  (activation.py:52-68)."""

    def forward(self, x: Tensor) -> Tensor:
        hidden = x.shape[1] // 2
        return silu_and_mul_kernel(x[:, :hidden], x[:, hidden:])
