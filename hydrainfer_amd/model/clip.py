"""CLIP vision tower + LLaVA projector — the caller of the dense attention path.

Mirrors hydrainfer/model/clip.py:10-135 (pre-LN encoder layers, quick-GELU MLP, class +
position embeddings, layers 0..vision_feature_layer) and hydrainfer/model/llava.py:30-41,99-107
(2-layer GELU projector, CLS token dropped).  Attention runs on the HIP dense kernel
(mha_varlen_fwd, non-causal, hydrainfer/layer/multihead_attention.py:114-160); the linears are library
GEMMs (q, k, v as ONE product over the three weights laid side by side); on the GPU the residual add + LayerNorm pairs
and the quick-GELU run as one hand-written launch each (hx_add_layer_norm, hx_quick_gelu): 8 launches per encoder
layer instead of 16 at a size (577 x 1024) where every launch is ~5 us whatever it does."""
from dataclasses import dataclass
from typing import Dict

import torch
import torch.nn.functional as F
from torch import Tensor

from hydrainfer_amd._C.kernel.activation import quick_gelu
from hydrainfer_amd._C.kernel.norm import add_layer_norm
from hydrainfer_amd.layer.multihead_attention import (MultiHeadAttention, MultiHeadAttentionConfig,
                                                      MultiHeadAttentionParameters)


@dataclass
class ClipShape:
    hidden_size: int = 1024
    intermediate_size: int = 4096
    num_hidden_layers: int = 24
    num_attention_heads: int = 16
    image_size: int = 336
    patch_size: int = 14
    num_channels: int = 3
    layer_norm_eps: float = 1e-5
    vision_feature_layer: int = -2
    projector_hidden_size: int = 4096   # language-model hidden size

    @property
    def num_positions(self) -> int:
        return (self.image_size // self.patch_size) ** 2 + 1


CLIP_VIT_L_14_336 = ClipShape()


class LlavaVisionModel:
    """state: reference-named tensors ('vision_tower.vision_model....', 'multi_modal_projector....')."""

    def __init__(self, shape: ClipShape, dtype: torch.dtype, device, state: Dict[str, Tensor]):
        self.shape, self.dtype, self.device, self.state = shape, dtype, torch.device(device), state
        self.attn = MultiHeadAttention(MultiHeadAttentionConfig(
            shape.num_attention_heads, shape.hidden_size // shape.num_attention_heads))
        L = shape.num_hidden_layers
        self.n_run = (shape.vision_feature_layer + L) % L + 1   # clip.py:106-108
        self._qkv: Dict[int, tuple] = {}

    def required_tensor_names(self):
        """The reference-named tensors forward() reads (layers past vision_feature_layer are never run)."""
        names = [k for k in random_state_dict(self.shape, std=0.0)
                 if ".encoder.layers." not in k or int(k.split(".encoder.layers.")[1].split(".")[0]) < self.n_run]
        return names

    @classmethod
    def random_init(cls, shape: ClipShape, dtype, device, seed: int = 0, std: float = 0.02):
        return cls(shape, dtype, device,
                   {k: v.to(dtype).to(device) for k, v in random_state_dict(shape, seed, std).items()})

    def _fused_qkv(self, l: int):
        """q, k and v projections as one [3h, h] weight / [3h] bias: built on first use; the reference-named tensors in
        `state` become views of it (no second copy)."""
        hit = self._qkv.get(l)
        if hit is None:
            s = self.state
            p = f"vision_tower.vision_model.encoder.layers.{l}.self_attn."
            names = ("q_proj", "k_proj", "v_proj")
            w = torch.cat([s[p + n + ".weight"] for n in names], dim=0)
            b = torch.cat([s[p + n + ".bias"] for n in names], dim=0)
            h = self.shape.hidden_size
            for i, n in enumerate(names):
                s[p + n + ".weight"], s[p + n + ".bias"] = w[i * h:(i + 1) * h], b[i * h:(i + 1) * h]
            hit = self._qkv[l] = (w, b)
        return hit

    def _layer(self, l: int, h: Tensor, x: Tensor):
        """h: the residual stream; x = layer_norm1(h) (computed by the previous layer's last launch).  Returns the new
        (h, x) — x is layer_norm1 of the NEXT layer, or None after the last layer that runs."""
        s, sh = self.state, self.shape
        p = f"vision_tower.vision_model.encoder.layers.{l}."
        hid = sh.hidden_size
        wqkv, bqkv = self._fused_qkv(l)
        qkv = F.linear(x, wqkv, bqkv)
        q, k, v = qkv[..., :hid], qkv[..., hid:2 * hid], qkv[..., 2 * hid:]       # views: the kernel takes the row stride
        o = self.attn(q, k, v, MultiHeadAttentionParameters()).o
        y = F.linear(o, s[p + "self_attn.out_proj.weight"], s[p + "self_attn.out_proj.bias"])
        x = torch.empty_like(h)
        add_layer_norm(x, h, y, s[p + "layer_norm2.weight"], s[p + "layer_norm2.bias"], sh.layer_norm_eps)       # h += y
        x = quick_gelu(F.linear(x, s[p + "mlp.fc1.weight"], s[p + "mlp.fc1.bias"]))    # QuickGELU, activation.py:17-22
        m = F.linear(x, s[p + "mlp.fc2.weight"], s[p + "mlp.fc2.bias"])
        if l + 1 == self.n_run:
            return h.add_(m), None
        pn = f"vision_tower.vision_model.encoder.layers.{l + 1}."
        x = torch.empty_like(h)
        add_layer_norm(x, h, m, s[pn + "layer_norm1.weight"], s[pn + "layer_norm1.bias"], sh.layer_norm_eps)     # h += m
        return h, x

    def forward(self, pixel_values: Tensor) -> Tensor:
        """pixel_values (n_images, C, H, W) -> image_features (n_images, n_patches, lm_hidden)."""
        s, sh = self.state, self.shape
        pre = "vision_tower.vision_model."
        n = pixel_values.shape[0]
        # The patch embedding is a stride == kernel convolution (clip.py:33-40 of the reference), i.e.
        # one GEMM over non-overlapping patches.  MIOpen runs the bf16 convolution on a naive kernel
        # (1.8 ms for 8 images, profiles/r1_serving7b_summary.md); the GEMM takes microseconds.
        w = s[pre + "embeddings.patch_embedding.weight"]
        ps, g = sh.patch_size, pixel_values.shape[-1] // sh.patch_size
        x = pixel_values.to(self.dtype).reshape(n, w.shape[1], g, ps, g, ps).permute(0, 2, 4, 1, 3, 5)
        patches = torch.matmul(x.reshape(n, g * g, -1), w.reshape(w.shape[0], -1).t())
        cls_tok = s[pre + "embeddings.class_embedding"].expand(n, 1, -1)
        h = torch.cat([cls_tok, patches], dim=1) + s[pre + "embeddings.position_embedding.weight"][None]
        h = F.layer_norm(h, (sh.hidden_size,), s[pre + "pre_layrnorm.weight"], s[pre + "pre_layrnorm.bias"],
                         sh.layer_norm_eps)
        p0 = pre + "encoder.layers.0."
        x = F.layer_norm(h, (sh.hidden_size,), s[p0 + "layer_norm1.weight"], s[p0 + "layer_norm1.bias"], sh.layer_norm_eps)
        for l in range(self.n_run):
            h, x = self._layer(l, h, x)
        feat = h[:, 1:]                                        # drop CLS (llava.py:104)
        x = F.linear(feat, s["multi_modal_projector.linear_1.weight"], s["multi_modal_projector.linear_1.bias"])
        x = F.gelu(x)
        return F.linear(x, s["multi_modal_projector.linear_2.weight"], s["multi_modal_projector.linear_2.bias"])

    __call__ = forward


def random_state_dict(shape: ClipShape, seed: int = 0, std: float = 0.02) -> Dict[str, Tensor]:
    """fp32 CPU state dict under the reference's parameter names; LayerNorm weight 1 / bias 0
    (SURVEY.md §8d synthetic weights)."""
    g = torch.Generator().manual_seed(seed)
    h, i = shape.hidden_size, shape.intermediate_size

    def w(*size):
        return torch.randn(size, generator=g) * std
    pre = "vision_tower.vision_model."
    sd = {pre + "embeddings.class_embedding": w(h),
          pre + "embeddings.patch_embedding.weight": w(h, shape.num_channels, shape.patch_size, shape.patch_size),
          pre + "embeddings.position_embedding.weight": w(shape.num_positions, h),
          pre + "pre_layrnorm.weight": torch.ones(h), pre + "pre_layrnorm.bias": torch.zeros(h)}
    for l in range(shape.num_hidden_layers):
        p = pre + f"encoder.layers.{l}."
        for name in ("q_proj", "k_proj", "v_proj", "out_proj"):
            sd[p + f"self_attn.{name}.weight"] = w(h, h)
            sd[p + f"self_attn.{name}.bias"] = w(h)
        sd[p + "mlp.fc1.weight"], sd[p + "mlp.fc1.bias"] = w(i, h), w(i)
        sd[p + "mlp.fc2.weight"], sd[p + "mlp.fc2.bias"] = w(h, i), w(h)
        for ln in ("layer_norm1", "layer_norm2"):
            sd[p + ln + ".weight"], sd[p + ln + ".bias"] = torch.ones(h), torch.zeros(h)
    ph = shape.projector_hidden_size
    sd["multi_modal_projector.linear_1.weight"], sd["multi_modal_projector.linear_1.bias"] = w(ph, h), w(ph)
    sd["multi_modal_projector.linear_2.weight"], sd["multi_modal_projector.linear_2.bias"] = w(ph, ph), w(ph)
    return sd
