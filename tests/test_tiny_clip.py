"""G9: tiny CLIP vision tower + LLaVA projector — oracle (CPU) and HIP-backed model (GPU)
against features produced by the reference's CLIPVisionModel + LlavaMultiModalProjector."""
import numpy as np
import pytest
import torch

from tests.golden import cases as C
from tests.util import assert_close_t, load_golden


def _setup(dname):
    from hydrainfer_amd.model.clip import ClipShape, random_state_dict
    dt = C.DTYPES[dname]
    shape = ClipShape(**C.TINY_CLIP)
    sd = {k: v.to(dt) for k, v in random_state_dict(shape, seed=3, std=0.05).items()}
    return shape, sd, dt


@pytest.mark.parametrize("dname", ["fp16", "bf16"])
def test_oracle_vision_matches_reference(dname):
    from oracle.vision import vision_forward
    g = load_golden("g9_tiny_clip")
    shape, sd, dt = _setup(dname)
    pixels = C.tiny_clip_pixels()
    assert C.checksum(pixels) == str(g["clip_pixels_chk"])
    feat = vision_forward(shape, sd, pixels)
    ref = C.from_np(g[f"clip_{dname}_features"], dt)
    tol = 2e-2 if dname == "bf16" else 2e-3
    assert_close_t(feat, ref, tol, tol, what=dname)


@pytest.mark.gpu
@pytest.mark.parametrize("dname", ["fp16", "bf16"])
def test_hip_vision_matches_reference_and_fills_image_cache(dname):
    from hydrainfer_amd.memory.token_cache import TokenCache
    from hydrainfer_amd.model.clip import LlavaVisionModel
    g = load_golden("g9_tiny_clip")
    shape, sd, dt = _setup(dname)
    dev = torch.device("cuda:0")
    model = LlavaVisionModel(shape, dt, dev, {k: v.to(dev) for k, v in sd.items()})
    feat = model(C.tiny_clip_pixels().to(dev))
    ref = C.from_np(g[f"clip_{dname}_features"], dt)
    # stated tolerance vs the reference CPU path: fp16 5e-3, bf16 4e-2 (abs + rel)
    tol = 4e-2 if dname == "bf16" else 5e-3
    assert_close_t(feat, ref, tol, tol, what=dname)
    # executor.py:228-231: image features -> image cache (block = all patches of an image)
    n_img, n_tok, hid = feat.shape
    H, D = 2, hid // 2
    cache = torch.zeros((4, n_tok, H, D), dtype=dt, device=dev)
    slots = torch.cat([torch.arange(n_tok) + 3 * n_tok, torch.arange(n_tok) + 1 * n_tok]).to(torch.int32).to(dev)
    TokenCache([cache]).set_caches(slots, [feat.reshape(n_img * n_tok, H, D)])
    assert torch.equal(cache[3].reshape(n_tok, hid), feat[0])
    assert torch.equal(cache[1].reshape(n_tok, hid), feat[1])
    assert float(cache[0].float().abs().sum()) == 0
