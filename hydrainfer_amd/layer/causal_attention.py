"""Causal grouped-query paged attention — host-side mirror of
hydrainfer/layer/causal_attention.py (AttentionParameters :31-107, builder :110-210,
module :377-406).  The reference's handler chain (FlashInfer -> flash_attn -> torch) is
replaced by the single HIP path; a CPU tensor or a missing library raises."""
import math
from dataclasses import dataclass
from typing import List, Optional

import torch
from torch import Tensor, nn

from hydrainfer_amd._C.kernel.flash_attn import mha_varlen_fwd
from hydrainfer_amd.memory.kv_cache import KVCache


@dataclass
class AttentionParameters:
    kv_cache: KVCache
    q_cu_seq_lens: Tensor = None           # int32 [n_seq + 1]
    kv_cu_seq_lens: Tensor = None          # int32 [n_seq + 1]
    paged_kv_last_page_len: Tensor = None  # int32 [n_seq]
    new_cache_slots: Tensor = None         # int32 [n_tokens]
    block_tables: Tensor = None            # int32 [sum blocks]  (flat)
    cu_blocks_lens: Tensor = None          # int32 [n_seq + 1]
    num_sequences: int = None
    all_sequences_decode: bool = False
    q_max_seq_len: int = 128
    kv_max_seq_len: int = 128
    # MI355X extension (not in the reference): the rank descriptor of an all-decode batch, int32 [1 + n_seq] — [0] = 1
    # when the batch is ragged, [1 + r] = the sequence with the r-th most keys.  The fused decode attention lays a big
    # ragged batch over the CUs in that order (csrc/attn_decode.hip, RANKED); None = the static grid.
    decode_rank: Tensor = None

    def to(self, device: torch.device) -> None:
        for f in ("q_cu_seq_lens", "kv_cu_seq_lens", "paged_kv_last_page_len", "new_cache_slots",
                  "block_tables", "cu_blocks_lens", "decode_rank"):
            t = getattr(self, f)
            if t is not None:
                setattr(self, f, t.to(device))


RANKED_THR = 1.125      # csrc/hx_common.h HX_RANKED_THR: ragged = some sequence longer than this x the mean + 16 keys


def decode_rank_descriptor(kv_lens: List[int]) -> List[int]:
    """The rank descriptor of a decode batch, worked out on the host (the engine builds its steps there): the same words
    as hx_decode_rank writes — [ragged?] + the sequences by decreasing length (ties: the lower number first)."""
    n = len(kv_lens)
    if n == 0 or n > 256:
        return [0] + list(range(n))
    mean = sum(kv_lens) / n
    ragged = any(float(l) > mean * RANKED_THR + 16.0 for l in kv_lens)
    return [1 if ragged else 0] + sorted(range(n), key=lambda b: (-kv_lens[b], b))


class AttentionParametersBuilder:
    """add_request(...) per sequence, add_kv_cache(...) per layer, then
    build_attention_parameters(): every layer shares the same control tensors."""

    def __init__(self, num_qo_heads: int, num_kv_heads: int, head_dim: int, block_size: int,
                 device: torch.device):
        self.num_qo_heads = num_qo_heads
        self.num_kv_heads = num_kv_heads
        self.head_dim = head_dim
        self.block_size = block_size
        self.device = device
        self.kv_caches: List[KVCache] = []
        self.q_cu_seq_lens: List[int] = [0]
        self.kv_cu_seq_lens: List[int] = [0]
        self.paged_kv_last_page_len: List[int] = []
        self.new_cache_slots: List[int] = []
        self.block_tables: List[int] = []
        self.cu_blocks_lens: List[int] = [0]
        self.num_sequences = 0
        self.all_sequences_decode = True
        self.q_max_seq_len = 0
        self.kv_max_seq_len = 0

    def add_request(self, q_seq_len: int, kv_seq_len: int, new_cache_slots: List[int],
                    block_table: List[int]) -> None:
        self.q_cu_seq_lens.append(self.q_cu_seq_lens[-1] + q_seq_len)
        self.kv_cu_seq_lens.append(self.kv_cu_seq_lens[-1] + kv_seq_len)
        self.paged_kv_last_page_len.append((kv_seq_len + self.block_size - 1) % self.block_size + 1)
        self.new_cache_slots += new_cache_slots
        self.block_tables += block_table
        self.cu_blocks_lens.append(self.cu_blocks_lens[-1] + len(block_table))
        self.num_sequences += 1
        self.all_sequences_decode = self.all_sequences_decode and q_seq_len == 1
        self.q_max_seq_len = max(self.q_max_seq_len, q_seq_len)
        self.kv_max_seq_len = max(self.kv_max_seq_len, kv_seq_len)

    def add_kv_cache(self, kv_cache: KVCache) -> None:
        self.kv_caches.append(kv_cache)

    def _tensors(self):
        # one pinned staging buffer + one H2D copy for all six arrays (the reference issues
        # six torch.tensor(list, device=...) copies per step, causal_attention.py:163-168)
        kv = self.kv_cu_seq_lens
        rank = decode_rank_descriptor([kv[i + 1] - kv[i] for i in range(self.num_sequences)]) \
            if self.all_sequences_decode and self.num_sequences > 0 else []
        lists = [self.q_cu_seq_lens, self.kv_cu_seq_lens, self.paged_kv_last_page_len,
                 self.new_cache_slots, self.block_tables, self.cu_blocks_lens, rank]
        flat = torch.tensor([x for l in lists for x in l], dtype=torch.int32)
        if self.device.type == "cuda":
            flat = flat.pin_memory().to(self.device, non_blocking=True)
        outs, off = [], 0
        for l in lists:
            outs.append(flat[off: off + len(l)])
            off += len(l)
        return outs

    def build_attention_parameters(self) -> List[AttentionParameters]:
        q_cu, kv_cu, last_page, slots, tables, cu_blocks, rank = self._tensors()
        return [AttentionParameters(
            kv_cache=kv_cache, q_cu_seq_lens=q_cu, kv_cu_seq_lens=kv_cu,
            paged_kv_last_page_len=last_page, new_cache_slots=slots, block_tables=tables,
            cu_blocks_lens=cu_blocks, num_sequences=self.num_sequences,
            all_sequences_decode=self.all_sequences_decode, q_max_seq_len=self.q_max_seq_len,
            kv_max_seq_len=self.kv_max_seq_len, decode_rank=rank if rank.numel() else None) for kv_cache in self.kv_caches]


@dataclass
class CausalGroupedQueryPageAttentionConfig:
    n_qo_heads: int
    n_kv_heads: int
    head_dim: int


@dataclass
class CausalGroupedQueryPageAttentionOutput:
    o: Tensor


class CausalGroupedQueryPageAttention(nn.Module):
    def __init__(self, config: CausalGroupedQueryPageAttentionConfig):
        super().__init__()
        assert config.n_qo_heads % config.n_kv_heads == 0
        self.n_qo_heads = config.n_qo_heads
        self.n_kv_heads = config.n_kv_heads
        self.head_dim = config.head_dim

    def forward(self, query: Tensor, key: Tensor, value: Tensor,
                attention_params: AttentionParameters) -> CausalGroupedQueryPageAttentionOutput:
        n_tokens = query.shape[0]
        query = query.view(n_tokens, self.n_qo_heads, self.head_dim)
        key = key.view(n_tokens, self.n_kv_heads, self.head_dim)
        value = value.view(n_tokens, self.n_kv_heads, self.head_dim)
        # append the new tokens, then attend over the cache (causal_attention.py:401-406)
        kv_cache = attention_params.kv_cache
        kv_cache.set_kv_cache(attention_params.new_cache_slots, key, value)
        key_cache, value_cache = kv_cache.get_kv_cache()
        output = torch.empty((n_tokens, self.n_qo_heads, self.head_dim), dtype=query.dtype,
                             device=query.device)
        mha_varlen_fwd(output, query, key_cache, value_cache, attention_params.q_cu_seq_lens,
                       attention_params.kv_cu_seq_lens, attention_params.block_tables,
                       attention_params.cu_blocks_lens, None, attention_params.q_max_seq_len,
                       attention_params.kv_max_seq_len, 1.0 / math.sqrt(self.head_dim), 0, -1, 0, 0)
        return CausalGroupedQueryPageAttentionOutput(o=output.view(n_tokens, self.n_qo_heads * self.head_dim))
