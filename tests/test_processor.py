"""ClipImageProcessor against transformers' CLIPImageProcessor (the processor the reference uses)."""
import numpy as np
import pytest
import torch
from PIL import Image


@pytest.mark.parametrize("hw", [(336, 336), (480, 640), (700, 300), (100, 137), (337, 1000)])
@pytest.mark.parametrize("mode", ["RGB", "L", "RGBA"])
def test_clip_preprocessing_matches_transformers(hw, mode):
    transformers = pytest.importorskip("transformers")
    from hydrainfer_amd.model.processor import ClipImageProcessor
    rng = np.random.RandomState(hw[0] + hw[1])
    channels = {"RGB": 3, "L": 1, "RGBA": 4}[mode]
    arr = rng.randint(0, 256, hw + ((channels,) if channels > 1 else ()), dtype=np.uint8)
    img = Image.fromarray(arr, mode=mode)
    ref = transformers.CLIPImageProcessor(size={"shortest_edge": 336}, crop_size={"height": 336, "width": 336},
                                          do_convert_rgb=True)(img, return_tensors="pt")["pixel_values"]
    got = ClipImageProcessor().process(img)
    assert got.shape == ref.shape == (1, 3, 336, 336) and got.dtype == torch.float32
    assert (got - ref).abs().max().item() <= 2e-6
