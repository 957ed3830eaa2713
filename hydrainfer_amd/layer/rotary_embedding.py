"""Rotary embedding — host-side mirror of hydrainfer/layer/rotary_embedding.py:13-146
(FusedKernelRotaryEmbeddingHandler path; cos/sin cache layout [max_pos, 2, rot/2], :111-116)."""
import torch
from torch import Tensor, nn

from hydrainfer_amd._C.kernel.position_embedding import apply_rotary_pos_emb


def compute_default_inv_freq(rotary_dim: int, theta: float) -> Tensor:
    assert rotary_dim % 2 == 0, "rotary_dim must be even"
    return 1.0 / torch.pow(theta, torch.arange(0, rotary_dim, 2, dtype=torch.float) / rotary_dim)


class RotaryEmbedding(nn.Module):
    def __init__(self, rotary_dim: int, max_position_embeddings: int, inv_freq: Tensor,
                 interleaved: bool):
        super().__init__()
        self.rotary_dim = rotary_dim
        self.max_position_embeddings = max_position_embeddings
        self.interleaved = interleaved
        t = torch.arange(max_position_embeddings, dtype=torch.float)
        freqs = torch.einsum("i,j->ij", t, inv_freq.to(torch.float))
        cos_sin = torch.cat([freqs.cos()[:, None, :], freqs.sin()[:, None, :]], dim=1)
        self.register_buffer("cos_sin_cache", cos_sin, persistent=False)

    def forward(self, query: Tensor, key: Tensor, position_ids: Tensor):
        # in place on query and key
        apply_rotary_pos_emb(query, key, position_ids, self.cos_sin_cache, self.rotary_dim,
                             self.interleaved)
        return query, key
