"""Per-step model inputs — mirror of hydrainfer/engine/parameters_builder.py:11-97.

One builder per fill batch: add(rcb, inst) for every request, then
build_language_model_parameters().  Two deliberate differences from the reference:
  * the integer arrays (token ids, positions, sampled rows, and the six attention arrays) reach
    the device in two pinned H2D copies instead of 6 + 4 `torch.tensor(list, device=...)` calls;
  * a sequence's kv length is `max(cache_ids) + 1`, not `virtual_kv_cache.n_cache_tokens`
    (:69).  The two are equal except on the head chunk of a budget-chunked prefill, where the
    reference has already grown the cache to the whole prompt (scheduler.py:137 runs before the
    chunking at :172) and so lets the chunk attend to cache slots that hold no token yet."""
from dataclasses import dataclass
from typing import List, Optional

import torch
from torch import Tensor

from hydrainfer_amd.engine.isa import Fill, ImageEmbedFill
from hydrainfer_amd.engine.rcb import BatchRequest, RequestControlBlock
from hydrainfer_amd.layer.causal_attention import AttentionParameters, AttentionParametersBuilder
from hydrainfer_amd.memory.kv_cache import KVCache


@dataclass
class FillInputs:
    input_ids: Tensor                 # int32 [n_tokens]
    position_ids: Tensor              # int32 [n_tokens]
    image_features: Optional[Tensor]  # [n_image_tokens, hidden] or None
    attention_params: List[AttentionParameters]
    all_sequences_decode: bool
    selected_token_ids: List[int]
    selected_token_ids_tensor: Optional[Tensor]   # int64 on the device
    image_row_index: Optional[Tensor] = None      # int64 on the device: rows that are image tokens


class LanguageModelParametersBuilder:
    def __init__(self, image_block_manager, kv_cache_block_manager, n_layers: int, n_qo_heads: int,
                 n_kv_heads: int, head_dim: int, image_token_id: int, dtype: torch.dtype,
                 device: torch.device):
        self.image_block_manager = image_block_manager
        self.kv_cache_block_manager = kv_cache_block_manager
        self.n_layers, self.n_qo_heads, self.head_dim = n_layers, n_qo_heads, head_dim
        self.image_token_id = image_token_id
        self.dtype, self.device = dtype, device
        self.image_slot_ids: List[int] = []
        self.token_ids: List[int] = []
        self.position_ids: List[int] = []
        self.selected_token_ids: List[int] = []
        self.attention_params_builder = AttentionParametersBuilder(
            num_qo_heads=n_qo_heads, num_kv_heads=n_kv_heads, head_dim=head_dim,
            block_size=kv_cache_block_manager.block_size, device=device)

    def add(self, rcb: RequestControlBlock, inst: Fill) -> None:
        assert isinstance(inst, Fill)
        if isinstance(inst, ImageEmbedFill):
            self.image_slot_ids += self.image_block_manager.v2p(rcb.virtual_image_cache,
                                                                inst.image_token_cache_ids)
        self.token_ids += inst.token_ids
        self.position_ids += inst.position_ids
        if inst.sample:
            self.selected_token_ids.append(len(self.token_ids) - 1)
        vc = rcb.virtual_kv_cache
        self.attention_params_builder.add_request(
            q_seq_len=len(inst.token_ids), kv_seq_len=max(inst.cache_ids) + 1,
            new_cache_slots=self.kv_cache_block_manager.v2p(vc, inst.cache_ids),
            block_table=vc.block_table)

    def add_batch(self, batch: BatchRequest) -> None:
        for rcb, inst in batch:
            self.add(rcb, inst)

    def _to_device(self, values: List[int], dtype=torch.int32) -> Tensor:
        t = torch.tensor(values, dtype=dtype)
        if self.device.type == "cuda":
            t = t.pin_memory().to(self.device, non_blocking=True)
        return t

    def build_language_model_parameters(self) -> FillInputs:
        n, n_sel, n_img = len(self.token_ids), len(self.selected_token_ids), len(self.image_slot_ids)
        n_image_rows = sum(t == self.image_token_id for t in self.token_ids)
        assert n_image_rows == n_img, "image token rows and cached image embeddings differ"
        image_rows = [i for i, t in enumerate(self.token_ids) if t == self.image_token_id] if n_img else []
        flat = self._to_device(self.token_ids + self.position_ids + self.selected_token_ids +
                               self.image_slot_ids + image_rows)
        input_ids, position_ids = flat[:n], flat[n:2 * n]
        selected = flat[2 * n:2 * n + n_sel].to(torch.int64) if n_sel else None
        image_features = None
        if n_img:
            cache = self.image_block_manager.get_layer_cache(layer_id=0).get_caches()[0]
            rows = cache.view(-1, self.n_qo_heads * self.head_dim)
            image_features = rows.index_select(0, flat[2 * n + n_sel:2 * n + n_sel + n_img]).to(self.dtype)
        for layer_id in range(self.n_layers):
            self.attention_params_builder.add_kv_cache(
                KVCache.from_token_cache(self.kv_cache_block_manager.get_layer_cache(layer_id)))
        attn = self.attention_params_builder.build_attention_parameters()
        row_index = flat[2 * n + n_sel + n_img:].to(torch.int64) if n_img else None
        return FillInputs(input_ids, position_ids, image_features, attn, attn[0].all_sequences_decode,
                          self.selected_token_ids, selected, row_index)
