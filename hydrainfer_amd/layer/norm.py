"""RMSNorm — host-side mirror of hydrainfer/layer/norm.py:12-33 (HIP kernel only)."""
import torch
from torch import Tensor, nn

from hydrainfer_amd._C.kernel.norm import rms_norm as rms_norm_kernel


def rmsnorm(h: Tensor, w: Tensor, eps: float) -> Tensor:
    o = torch.empty_like(h)
    rms_norm_kernel(o, h, w, eps)
    return o


class RMSNorm(nn.Module):
    def __init__(self, hidden_size: int, eps: float) -> None:
        super().__init__()
        self.weight = nn.Parameter(torch.ones(hidden_size))
        self.variance_epsilon = eps

    def forward(self, hidden_states: Tensor) -> Tensor:
        return rmsnorm(hidden_states, self.weight, self.variance_epsilon)
