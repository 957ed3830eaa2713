"""CLIP image preprocessing for LLaVA-1.5 (what the reference gets from the HF processor,
hydrainfer/model/model_factory.py -> AutoProcessor; the request path calls
`processor.process(image)`, engine/request_processor.py:104): convert to RGB, resize the shorter
side to 336 with bicubic resampling, centre-crop 336 x 336, scale to [0, 1], normalise with the CLIP
mean / std.  Pinned bit-for-bit against transformers' CLIPImageProcessor on PIL inputs
(tests/test_processor.py)."""
from typing import Tuple

import numpy as np
import torch
from PIL import Image

CLIP_MEAN = (0.48145466, 0.4578275, 0.40821073)
CLIP_STD = (0.26862954, 0.26130258, 0.27577711)


class ClipImageProcessor:
    def __init__(self, size: int = 336, mean: Tuple[float, ...] = CLIP_MEAN, std: Tuple[float, ...] = CLIP_STD):
        self.size = size
        self.mean = np.asarray(mean, dtype=np.float32)
        self.std = np.asarray(std, dtype=np.float32)

    def process(self, image: Image.Image) -> torch.Tensor:
        """PIL image -> float32 pixel_values (1, 3, size, size)."""
        if image.mode != "RGB":
            image = image.convert("RGB")
        w, h = image.size
        short, long = (w, h) if w <= h else (h, w)
        new_short, new_long = self.size, int(self.size * long / short)
        new_w, new_h = (new_short, new_long) if w <= h else (new_long, new_short)
        image = image.resize((new_w, new_h), resample=Image.Resampling.BICUBIC)
        left, top = (new_w - self.size) // 2, (new_h - self.size) // 2
        image = image.crop((left, top, left + self.size, top + self.size))
        x = np.asarray(image, dtype=np.float32) * np.float32(1.0 / 255.0)
        x = (x - self.mean) / self.std
        return torch.from_numpy(np.ascontiguousarray(x.transpose(2, 0, 1)))[None]
