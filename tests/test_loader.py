"""Checkpoint loader: a synthetic tiny LLaVA checkpoint (HF layout, two safetensors shards) loads
into the same fused tensors as direct construction; missing tensors fail loudly."""
import json
import os

import pytest
import torch

from tests.golden import cases as C


def _write_checkpoint(path, drop=None):
    import safetensors.torch
    from hydrainfer_amd.model.clip import ClipShape, random_state_dict
    t, v = C.TINY_LLAMA, C.TINY_CLIP
    cfg = {"image_token_index": C.TINY_IMAGE_TOKEN_ID, "vision_feature_layer": v["vision_feature_layer"],
           "text_config": {k: t[k] for k in ("hidden_size", "intermediate_size", "num_hidden_layers",
                                             "num_attention_heads", "num_key_value_heads", "vocab_size")},
           "vision_config": {k: v[k] for k in ("hidden_size", "intermediate_size", "num_hidden_layers",
                                               "num_attention_heads", "image_size", "patch_size")}}
    cfg["text_config"]["head_dim"] = t["head_dim"]
    json.dump(cfg, open(os.path.join(path, "config.json"), "w"))
    lm = {"language_model." + k: w.contiguous() for k, w in C.tiny_llama_state_dict(torch.float32).items()}
    vis = random_state_dict(ClipShape(**v), seed=3, std=0.05)
    vis["vision_tower.vision_model.post_layernorm.weight"] = torch.ones(v["hidden_size"])   # present in real files, unused
    tensors = {**lm, **vis}
    if drop:
        tensors.pop(drop)
    names = sorted(tensors)
    half = len(names) // 2
    safetensors.torch.save_file({k: tensors[k] for k in names[:half]}, os.path.join(path, "model-00001-of-00002.safetensors"))
    safetensors.torch.save_file({k: tensors[k] for k in names[half:]}, os.path.join(path, "model-00002-of-00002.safetensors"))
    return vis


def test_loader_round_trip(tmp_path):
    from hydrainfer_amd.model.llama import LlamaForCausalLM, LlamaShape
    from hydrainfer_amd.model.loader import load_llava, read_shapes
    vis = _write_checkpoint(str(tmp_path))
    lshape, cshape, image_token_id = read_shapes(str(tmp_path))
    assert lshape == LlamaShape(**C.TINY_LLAMA) and image_token_id == C.TINY_IMAGE_TOKEN_ID
    assert cshape.projector_hidden_size == lshape.hidden_size and cshape.vision_feature_layer == -2
    lm, vm = load_llava(str(tmp_path), torch.float16, "cpu")
    direct = LlamaForCausalLM.from_reference_state_dict(lshape, C.tiny_llama_state_dict(torch.float16), torch.float16, "cpu")
    assert sorted(lm.language_model.state) == sorted(direct.state)
    for k, w in direct.state.items():
        assert torch.equal(lm.language_model.state[k], w), k
    for k in vm.required_tensor_names():
        assert torch.equal(vm.state[k], vis[k].to(torch.float16)), k
    only_vision = load_llava(str(tmp_path), torch.float16, "cpu", language=False)
    assert only_vision[0] is None and only_vision[1] is not None


@pytest.mark.parametrize("drop", ["language_model.model.layers.1.mlp.up_proj.weight",
                                  "vision_tower.vision_model.encoder.layers.0.mlp.fc1.bias"])
def test_loader_reports_missing_tensors(tmp_path, drop):
    from hydrainfer_amd.model.loader import load_llava
    _write_checkpoint(str(tmp_path), drop=drop)
    with pytest.raises(RuntimeError, match="missing"):
        load_llava(str(tmp_path), torch.float16, "cpu")


@pytest.mark.gpu
def test_offline_engine_from_checkpoint(tmp_path):
    """Checkpoint directory -> OfflineInferenceEngine -> tokens; the same tokens as the oracle-backed
    engine on CPU wherever the oracle's greedy choice is clear (first token of every request here)."""
    from hydrainfer_amd.engine.offline import OfflineInferenceEngine, OfflineRequest
    from PIL import Image
    import numpy as np
    _write_checkpoint(str(tmp_path))
    eng = OfflineInferenceEngine.from_checkpoint(str(tmp_path), torch.float16, "cuda:0", max_running_requests=4,
                                                 token_budgets=64, max_context=256, warm_up=False)
    g = torch.Generator().manual_seed(0)
    rng = np.random.RandomState(1)
    reqs = []
    for i in range(6):
        text = torch.randint(0, C.TINY_IMAGE_TOKEN_ID, (5 + 7 * i,), generator=g).tolist()
        img = Image.fromarray(rng.randint(0, 256, (60 + 10 * i, 80, 3), dtype=np.uint8)) if i % 3 != 2 else None
        reqs.append(OfflineRequest(([C.TINY_IMAGE_TOKEN_ID] if img is not None else []) + text, img, max_tokens=3 + i))
    outs = eng.generate(reqs)
    for r, o in zip(reqs, outs):
        assert len(o.output_token_ids) == r.max_tokens and o.ttft > 0 and len(o.tpot) == r.max_tokens - 1
        assert all(0 <= t < C.TINY_LLAMA["vocab_size"] for t in o.output_token_ids)
    again = eng.generate(reqs)           # pools are clean again and prefix hits do not change tokens
    assert [o.output_token_ids for o in again] == [o.output_token_ids for o in outs]
    kv = eng.node.kv_cache_block_manager
    assert kv.get_metrics().cache_hit_rate > 0
