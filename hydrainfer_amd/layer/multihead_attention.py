"""Dense multi-head attention (vision tower) — host-side mirror of
hydrainfer/layer/multihead_attention.py:21-176 (FlashAttentionMutliHeadAttentionHandler2 path:
mha_varlen_fwd dense, non-causal, window (-1,-1), :114-160)."""
import math
from dataclasses import dataclass
from typing import Optional

import torch
from torch import Tensor, nn

from hydrainfer_amd._C.kernel.flash_attn import mha_varlen_fwd


@dataclass
class MultiHeadAttentionConfig:
    n_heads: int
    head_dim: int


@dataclass
class MultiHeadAttentionParameters:
    return_scores: bool = False


@dataclass
class MultiHeadAttentionOutput:
    o: Tensor
    attention_scores: Optional[Tensor]


class MultiHeadAttention(nn.Module):
    def __init__(self, config: MultiHeadAttentionConfig):
        super().__init__()
        self.n_heads = config.n_heads
        self.head_dim = config.head_dim
        self._cu = {}      # (batch, seq_len, device) -> cumulative lengths: built once, not by a launch per call

    def forward(self, query: Tensor, key: Tensor, value: Tensor,
                params: MultiHeadAttentionParameters) -> MultiHeadAttentionOutput:
        if params.return_scores:
            raise NotImplementedError("attention scores are not materialised by the fused HIP kernel")
        batch_size, seq_len, hidden_size = query.shape
        q = query.reshape(batch_size * seq_len, self.n_heads, self.head_dim)
        k = key.reshape(batch_size * seq_len, self.n_heads, self.head_dim)
        v = value.reshape(batch_size * seq_len, self.n_heads, self.head_dim)
        o = torch.empty((batch_size * seq_len, self.n_heads, self.head_dim), dtype=query.dtype,
                        device=query.device)
        key_cu = (batch_size, seq_len, query.device)
        cu = self._cu.get(key_cu)
        if cu is None:
            cu = torch.arange(0, (batch_size + 1) * seq_len, seq_len, dtype=torch.int32, device=query.device)
            if not (query.is_cuda and torch.cuda.is_current_stream_capturing()):      # (a tensor born inside a capture holds
                if len(self._cu) > 64:                                                #  nothing until the graph has run)
                    self._cu.clear()
                self._cu[key_cu] = cu
        mha_varlen_fwd(o, q, k, v, cu, cu, None, None, None, seq_len, seq_len,
                       1.0 / math.sqrt(self.head_dim), 0, -1, -1, 0)
        return MultiHeadAttentionOutput(o=o.view(batch_size, seq_len, hidden_size), attention_scores=None)
