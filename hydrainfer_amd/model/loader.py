"""LLaVA-1.5 checkpoint directory (HF layout: config.json + *.safetensors) -> the two models of
the serving path.  Role of hydrainfer/model/model_loader.py:11-42 (scan every .safetensors file,
copy what belongs to the model, assert that nothing is missing) and of the `from_safetensor`
constructors in hydrainfer/model/llava.py:53-124 — with the language weights fused on the fly into
the [q|k|v] and [gate|up] tensors the decode kernels stream."""
import json
import os
from typing import Dict, Tuple

import torch
from torch import Tensor

from hydrainfer_amd.model.clip import ClipShape, LlavaVisionModel
from hydrainfer_amd.model.llama import LlamaForCausalLM, LlamaShape
from hydrainfer_amd.model.llava import LlavaLanguageModel


def read_shapes(model_path: str) -> Tuple[LlamaShape, ClipShape, int]:
    """(language shape, vision shape, image_token_index) from config.json (HF LlavaConfig)."""
    cfg = json.load(open(os.path.join(model_path, "config.json")))
    t, v = cfg.get("text_config", {}), cfg.get("vision_config", {})
    hidden = t.get("hidden_size", 4096)
    heads = t.get("num_attention_heads", 32)
    lshape = LlamaShape(
        hidden_size=hidden, intermediate_size=t.get("intermediate_size", 11008),
        num_hidden_layers=t.get("num_hidden_layers", 32), num_attention_heads=heads,
        num_key_value_heads=t.get("num_key_value_heads", heads), head_dim=t.get("head_dim", hidden // heads),
        vocab_size=t.get("vocab_size", cfg.get("vocab_size", 32064)), rms_norm_eps=t.get("rms_norm_eps", 1e-5),
        rope_theta=t.get("rope_theta", 10000.0), max_position_embeddings=t.get("max_position_embeddings", 4096))
    cshape = ClipShape(
        hidden_size=v.get("hidden_size", 1024), intermediate_size=v.get("intermediate_size", 4096),
        num_hidden_layers=v.get("num_hidden_layers", 24), num_attention_heads=v.get("num_attention_heads", 16),
        image_size=v.get("image_size", 336), patch_size=v.get("patch_size", 14),
        num_channels=v.get("num_channels", 3), layer_norm_eps=v.get("layer_norm_eps", 1e-5),
        vision_feature_layer=cfg.get("vision_feature_layer", -2), projector_hidden_size=hidden)
    return lshape, cshape, cfg.get("image_token_index", 32000)


def scan_safetensors(model_path: str, prefixes: Tuple[str, ...]) -> Dict[str, Tensor]:
    """All tensors whose name starts with one of `prefixes`, from every shard in the directory."""
    import safetensors.torch
    out: Dict[str, Tensor] = {}
    for entry in sorted(os.scandir(model_path), key=lambda e: e.name):
        if entry.is_file() and entry.name.endswith(".safetensors"):
            for name, weight in safetensors.torch.load_file(entry.path).items():
                if name.startswith(prefixes):
                    out[name] = weight
    return out


def load_llava(model_path: str, dtype: torch.dtype, device, vision: bool = True, language: bool = True):
    """Returns (LlavaLanguageModel or None, LlavaVisionModel or None) — an E node needs only the
    second, a D node only the first (NodeType.has_vision_model / has_language_model)."""
    lshape, cshape, image_token_id = read_shapes(model_path)
    lm = vm = None
    if language:
        sd = scan_safetensors(model_path, ("language_model.",))
        try:
            model = LlamaForCausalLM.from_reference_state_dict(lshape, sd, dtype, device, prefix="language_model.")
        except KeyError as e:
            raise RuntimeError(f"{model_path}: language-model tensor {e} is missing from the checkpoint") from None
        n_expected = 3 + 9 * lshape.num_hidden_layers
        assert len(sd) >= n_expected, f"expected {n_expected} language tensors, found {len(sd)}"
        lm = LlavaLanguageModel(model, image_token_id=image_token_id)
    if vision:
        sd = scan_safetensors(model_path, ("vision_tower.", "multi_modal_projector."))
        state = {k: v.to(dtype).to(device) for k, v in sd.items()}
        vm = LlavaVisionModel(cshape, dtype, device, state)
        missing = [k for k in vm.required_tensor_names() if k not in state]
        if missing:
            raise RuntimeError(f"{model_path}: {len(missing)} vision tensors missing, e.g. {missing[:3]}")
    return lm, vm
