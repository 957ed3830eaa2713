"""LLaVA-1.5 language side — mirrors hydrainfer/model/llava.py:110-140 (LlavaLanguageModel):
token embedding, image-token rows overwritten by projected image features, Llama decoder,
greedy token ids out."""
from typing import Optional

import torch
from torch import Tensor

from hydrainfer_amd.model.llama import LanguageModelParameters, LlamaForCausalLM


class LlavaLanguageModel:
    def __init__(self, language_model: LlamaForCausalLM, image_token_id: int = 32000):
        self.language_model = language_model
        self.image_token_id = image_token_id

    def embed(self, input_ids: Tensor, image_features: Optional[Tensor],
              image_row_index: Optional[Tensor] = None) -> Tensor:
        input_embeds = self.language_model.embed(input_ids)
        if image_features is not None:
            feats = image_features.reshape(-1, input_embeds.shape[-1]).to(input_embeds.dtype)
            if image_row_index is not None:       # rows known on the host: no device->host sync
                input_embeds.index_copy_(0, image_row_index, feats)
            else:
                mask = input_ids == self.image_token_id                   # llava.py:133-135
                input_embeds[mask] = feats
        return input_embeds

    def forward_logits(self, input_ids: Tensor, image_features: Optional[Tensor], position_ids: Tensor,
                       model_params: LanguageModelParameters) -> Tensor:
        embeds = self.embed(input_ids, image_features, getattr(model_params, "image_row_index", None))
        return self.language_model.forward_logits(embeds, position_ids, model_params)

    def forward(self, input_ids: Tensor, image_features: Optional[Tensor], position_ids: Tensor,
                model_params: LanguageModelParameters) -> Tensor:
        return torch.argmax(self.forward_logits(input_ids, image_features, position_ids, model_params), dim=-1)

    __call__ = forward
