"""Prefix (shared) block cache — host-side mirror of hydrainfer/memory/shared_cache.py:20-97.
Block hashes are chained xxh64 over the int64 little-endian bytes of a block's token ids,
prefixed by the previous block's hash as 8 LE bytes (shared_cache.py:73-88)."""
from dataclasses import dataclass
from typing import Dict, List

import numpy as np
import xxhash


@dataclass
class SharedBlock:
    ref_count: int


@dataclass
class SharedCacheConfig:
    n_blocks: int


class SharedCache:
    def __init__(self, config: SharedCacheConfig):
        self.hash_to_block_id: Dict[int, int] = {}
        self.block_id_to_hash: List[int] = list(range(config.n_blocks))
        self.blocks: List[SharedBlock] = [SharedBlock(0) for _ in range(config.n_blocks)]
        self.to_be_evicted = set(range(config.n_blocks))

    def match(self, hashes: List[int]) -> List[int]:
        return [self.hash_to_block_id.get(h, -1) for h in hashes]

    def pin(self, block_ids: List[int]) -> None:
        for b in block_ids:
            self.blocks[b].ref_count += 1
            assert self.blocks[b].ref_count > 0
            self.to_be_evicted.discard(b)

    def unpin(self, block_ids: List[int]) -> None:
        for b in block_ids:
            self.blocks[b].ref_count -= 1
            assert self.blocks[b].ref_count >= 0
            if self.blocks[b].ref_count == 0:
                self.to_be_evicted.add(b)

    def insert(self, hashes: List[int], block_ids: List[int]) -> None:
        for h, b in zip(hashes, block_ids):
            self.hash_to_block_id[h] = b
            self.block_id_to_hash[b] = h

    def evict(self, n_blocks: int) -> List[int]:
        evicted: List[int] = []
        for _ in range(min(n_blocks, len(self.to_be_evicted))):
            b = self.to_be_evicted.pop()
            h = self.block_id_to_hash[b]
            evicted.append(b)
            self.hash_to_block_id.pop(h, None)
            self.block_id_to_hash[b] = -1
        return evicted

    def allocate(self, n_blocks: int) -> List[int]:
        return self.evict(n_blocks)

    def get_num_avaiable_blocks(self) -> int:
        return len(self.to_be_evicted)

    def is_write_safe(self, block_id: int) -> bool:
        return self.blocks[block_id].ref_count == 1


def compute_block_hash(token_ids: List[int], prefix: int = -1) -> int:
    h = xxhash.xxh64()
    if prefix != -1:
        h.update(prefix.to_bytes(8, "little"))
    h.update(np.array(token_ids).tobytes())
    return h.intdigest()


def compute_hash(token_ids: List[int], block_size: int, prefix: int) -> List[int]:
    hashes = []
    h = prefix
    for i in range(len(token_ids) // block_size):
        h = compute_block_hash(token_ids[i * block_size: (i + 1) * block_size], prefix=h)
        hashes.append(h)
    return hashes


def compute_image_hash(image) -> int:
    """xxh64 over the RGB bytes of the image (shared_cache.py:91-97 of the reference); accepts a
    PIL image or an H x W x 3 uint8 array."""
    if hasattr(image, "mode"):                      # PIL.Image
        if image.mode != "RGB":
            image = image.convert("RGB")
        image = np.array(image)
    h = xxhash.xxh64()
    h.update(np.ascontiguousarray(image).tobytes())
    return h.intdigest()
