"""oracle/vision.py — CPU restatement of the reference's CLIP vision tower + LLaVA projector
(hydrainfer/model/clip.py:10-135, hydrainfer/model/llava.py:30-41,99-107) with the torch
attention handler (hydrainfer/layer/multihead_attention.py:46-70 via oracle.ops).
TEST INFRASTRUCTURE ONLY.  Pinned by tests/golden/g9_tiny_clip.npz (reference CLIPVisionModel +
LlavaMultiModalProjector run on CPU by tests/golden/generate_goldens.py)."""
from typing import Dict

import torch
import torch.nn.functional as F
from torch import Tensor

from oracle import ops


def vision_forward(shape, sd: Dict[str, Tensor], pixel_values: Tensor) -> Tensor:
    dtype = sd["multi_modal_projector.linear_1.weight"].dtype
    pre = "vision_tower.vision_model."
    H = shape.num_attention_heads
    D = shape.hidden_size // H
    n = pixel_values.shape[0]
    patches = F.conv2d(pixel_values.to(dtype), sd[pre + "embeddings.patch_embedding.weight"],
                       stride=shape.patch_size).flatten(2).transpose(1, 2)
    h = torch.cat([sd[pre + "embeddings.class_embedding"].expand(n, 1, -1), patches], dim=1)
    h = h + sd[pre + "embeddings.position_embedding.weight"][None]
    h = F.layer_norm(h, (shape.hidden_size,), sd[pre + "pre_layrnorm.weight"], sd[pre + "pre_layrnorm.bias"],
                     shape.layer_norm_eps)
    L = shape.num_hidden_layers
    n_run = (shape.vision_feature_layer + L) % L + 1
    S = h.shape[1]
    cu = torch.arange(0, (n + 1) * S, S, dtype=torch.int32)
    for l in range(n_run):
        p = pre + f"encoder.layers.{l}."
        x = F.layer_norm(h, (shape.hidden_size,), sd[p + "layer_norm1.weight"], sd[p + "layer_norm1.bias"],
                         shape.layer_norm_eps)
        q = F.linear(x, sd[p + "self_attn.q_proj.weight"], sd[p + "self_attn.q_proj.bias"])
        k = F.linear(x, sd[p + "self_attn.k_proj.weight"], sd[p + "self_attn.k_proj.bias"])
        v = F.linear(x, sd[p + "self_attn.v_proj.weight"], sd[p + "self_attn.v_proj.bias"])
        o = ops.varlen_attention(q.reshape(n * S, H, D), k.reshape(n * S, H, D), v.reshape(n * S, H, D),
                                 cu, cu, causal=False).reshape(n, S, H * D)
        h = h + F.linear(o, sd[p + "self_attn.out_proj.weight"], sd[p + "self_attn.out_proj.bias"])
        x = F.layer_norm(h, (shape.hidden_size,), sd[p + "layer_norm2.weight"], sd[p + "layer_norm2.bias"],
                         shape.layer_norm_eps)
        x = F.linear(x, sd[p + "mlp.fc1.weight"], sd[p + "mlp.fc1.bias"])
        x = x * torch.sigmoid(1.702 * x)
        h = h + F.linear(x, sd[p + "mlp.fc2.weight"], sd[p + "mlp.fc2.bias"])
    x = F.linear(h[:, 1:], sd["multi_modal_projector.linear_1.weight"], sd["multi_modal_projector.linear_1.bias"])
    return F.linear(F.gelu(x), sd["multi_modal_projector.linear_2.weight"],
                    sd["multi_modal_projector.linear_2.bias"])
