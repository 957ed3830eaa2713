"""oracle/model.py — CPU restatement of the reference's Llama/LLaVA language-model forward
(the "eager-PyTorch CPU path" of BASELINE config 0), built on oracle/ops.py.

TEST INFRASTRUCTURE / reported CPU baseline only (see oracle/ops.py header).

Follows, op for op, the reference's CPU execution:
  hydrainfer/model/llama.py:21-104 (module structure, greedy argmax :99-104),
  hydrainfer/model/model_forward.py:66-105 (attention + decoder layer),
  hydrainfer/layer/norm.py:18-23 (torch rmsnorm), rotary_embedding.py:46-83,
  hydrainfer/layer/activation.py:24-25 (F.silu on CPU), memory/kv_cache.py:44-50,
  hydrainfer/layer/causal_attention.py:307-374.
Weights are consumed under the reference's (HF Llama) parameter names."""
from dataclasses import dataclass
from typing import Dict, List, Optional

import torch
import torch.nn.functional as F
from torch import Tensor

from oracle import ops


@dataclass
class OracleAttnMeta:
    """The six integer tensors of AttentionParameters (causal_attention.py:31-66), on CPU."""
    q_cu_seq_lens: Tensor
    kv_cu_seq_lens: Tensor
    new_cache_slots: Tensor
    block_tables: Tensor
    cu_blocks_lens: Tensor


class OracleLlama:
    def __init__(self, shape, sd: Dict[str, Tensor], dtype: torch.dtype,
                 n_layers: Optional[int] = None):
        self.shape, self.sd, self.dtype = shape, sd, dtype
        self.n_layers = shape.num_hidden_layers if n_layers is None else n_layers
        self.cos_sin = ops.build_cos_sin_cache(shape.head_dim, shape.max_position_embeddings,
                                               shape.rope_theta, dtype)

    def layer(self, l: int, h: Tensor, position_ids: Tensor, meta: OracleAttnMeta,
              key_cache: Tensor, value_cache: Tensor, select: Optional[Tensor]) -> Tensor:
        sh, sd = self.shape, self.sd
        p = f"model.layers.{l}."
        H, HK, D = sh.num_attention_heads, sh.num_key_value_heads, sh.head_dim
        x = ops.rms_norm_torch(h, sd[p + "input_layernorm.weight"], sh.rms_norm_eps)
        q = F.linear(x, sd[p + "self_attn.q_proj.weight"]).view(-1, H, D)
        k = F.linear(x, sd[p + "self_attn.k_proj.weight"]).view(-1, HK, D)
        v = F.linear(x, sd[p + "self_attn.v_proj.weight"]).view(-1, HK, D)
        q, k = ops.apply_rotary_pos_emb(q, k, position_ids, self.cos_sin, D, False)
        ops.set_kv_cache(meta.new_cache_slots, k, v, key_cache, value_cache)
        o = ops.paged_attention(q, key_cache, value_cache, meta.q_cu_seq_lens, meta.kv_cu_seq_lens,
                                meta.block_tables, meta.cu_blocks_lens)
        h = h + F.linear(o.reshape(-1, H * D), sd[p + "self_attn.o_proj.weight"])
        if select is not None and l == sh.num_hidden_layers - 1:
            h = h[select]
        x = ops.rms_norm_torch(h, sd[p + "post_attention_layernorm.weight"], sh.rms_norm_eps)
        m = F.linear(ops.silu(F.linear(x, sd[p + "mlp.gate_proj.weight"])) *
                     F.linear(x, sd[p + "mlp.up_proj.weight"]), sd[p + "mlp.down_proj.weight"])
        return h + m

    def forward_logits(self, input_ids_or_embeds: Tensor, position_ids: Tensor,
                       meta: OracleAttnMeta, caches: List, select: Optional[Tensor] = None) -> Tensor:
        """caches[l] = (key_cache, value_cache) CPU tensors [n_blocks, bs, HK, D]."""
        if input_ids_or_embeds.dtype in (torch.int32, torch.int64):
            h = F.embedding(input_ids_or_embeds.long(), self.sd["model.embed_tokens.weight"])
        else:
            h = input_ids_or_embeds
        h = self.forward_hidden(h, position_ids, meta, caches, select)
        return F.linear(h, self.sd["lm_head.weight"])

    def forward_hidden(self, input_ids_or_embeds: Tensor, position_ids: Tensor, meta: OracleAttnMeta, caches: List,
                       select: Optional[Tensor] = None) -> Tensor:
        """The lm_head's input: the decoder layers + the final norm (llama.py:88-98)."""
        if input_ids_or_embeds.dtype in (torch.int32, torch.int64):
            h = F.embedding(input_ids_or_embeds.long(), self.sd["model.embed_tokens.weight"])
        else:
            h = input_ids_or_embeds
        for l in range(self.n_layers):
            h = self.layer(l, h, position_ids, meta, caches[l][0], caches[l][1], select)
        return ops.rms_norm_torch(h, self.sd["model.norm.weight"], self.shape.rms_norm_eps)

    def forward(self, *a, **k) -> Tensor:
        return torch.argmax(self.forward_logits(*a, **k), dim=-1)
