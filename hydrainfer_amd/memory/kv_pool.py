"""Where a KV pool's bytes lie in HBM.

The reference's pool is ONE contiguous 6-D tensor (n_layers, 2, n_blocks, block_size, n_heads, head_size)
(hydrainfer/memory/token_cache_manger.py:65); a layer's caches are the views pool[l, 0] / pool[l, 1].  Here the pool has
the same shape and the same views, but its (layer, k/v) PLANES are not back to back: every plane starts
KV_POOL_SKEW_BYTES past the end of the one before it.

Why (round 5, tools/probes/attn_placement.py, tools/probes/sweep64.sh): the decode-attention kernel reads K and V of one
(block, token, head) from the same wave at the same time.  In a contiguous pool the two lie n_blocks * block_bytes apart
— with pools sized to powers of two (ipc_safe_n_blocks) 256 or 512 MiB — so every such pair lands on the same HBM
channel.  An odd multiple of 256 bytes between the planes puts them on neighbouring channels: the fused decode-attention
launch of 64 sequences 141.7 -> 136 us, the 64-row decode step 7.92 -> 7.65 ms (-3.3 %), LLaVA-1.5-13B's batch-32 step
-1 %, the 7B batch-32 step -0.3 % (medians of six fresh processes each; 256, 768, 1280, 2304 and 4352 bytes are within
noise of one another, 512 and 1024 change nothing).

Nothing else about the format moves: a block is still [block_size, n_heads, head_size] contiguous, block ids index a
plane, pool[l, t] is a contiguous 4-D tensor.  What does change is that the 6-D tensor as a whole is not contiguous, so
the migration entries take the plane stride (hx_*_blocks_planes) and a pool's IPC handle carries it
(_C/data_transfer/block_migration.get_ipc_mem_handle)."""
import os
from typing import Tuple

import torch
from torch import Tensor

KV_POOL_SKEW_BYTES = int(os.environ.get("HX_KV_POOL_SKEW", "768"))      # (the variable exists for tools/probes/sweep64.sh)


def plane_elems(n_blocks: int, block_size: int, n_heads: int, head_size: int, itemsize: int,
                skew_bytes: int = None) -> int:
    skew = KV_POOL_SKEW_BYTES if skew_bytes is None else skew_bytes
    if skew % 16 != 0 or skew < 0:
        raise ValueError("the plane skew must be a non-negative multiple of 16 bytes (the kernels' vector width)")
    return n_blocks * block_size * n_heads * head_size + skew // itemsize


def pool_bytes(n_layers: int, n_tokens: int, n_blocks: int, block_size: int, n_heads: int, head_size: int,
               itemsize: int, skew_bytes: int = None) -> int:
    """Bytes of the allocation behind allocate_kv_pool's tensor."""
    return n_layers * n_tokens * plane_elems(n_blocks, block_size, n_heads, head_size, itemsize, skew_bytes) * itemsize


def allocate_kv_pool(shape: Tuple[int, int, int, int, int, int], dtype: torch.dtype, device, fill: str = "randn",
                     skew_bytes: int = None, generator=None) -> Tensor:
    """The 6-D pool (n_layers, n_tokens, n_blocks, block_size, n_heads, head_size) with skewed planes.  fill: "randn"
    (the reference's "garbage but finite", token_cache_manger.py:65, written plane by plane so the fp32 temporaries
    stay bounded), "zeros" or "empty"."""
    L, T, n_blocks, bs, H, D = shape
    itemsize = torch.empty((), dtype=dtype).element_size()
    plane = plane_elems(n_blocks, bs, H, D, itemsize, skew_bytes)
    flat = torch.zeros(L * T * plane, dtype=dtype, device=device) if fill == "zeros" \
        else torch.empty(L * T * plane, dtype=dtype, device=device)
    pool = flat.as_strided((L, T, n_blocks, bs, H, D), (T * plane, plane, bs * H * D, H * D, D, 1))
    if fill == "randn":
        for l in range(L):      # one layer at a time: bounded fp32 temporaries
            if generator is not None:
                pool[l].copy_(torch.randn(pool[l].shape, generator=generator, device=device, dtype=torch.float32).to(dtype))
            else:
                for t in range(T):      # plane by plane: each is contiguous (the strided fill of a whole layer is 5 x slower,
                    pool[l, t].normal_()    # and an engine that starts serving behind it pays that in its first TTFTs)
        if plane > n_blocks * bs * H * D:      # the spare bytes are never read; finite all the same (dumps, checksums)
            flat.as_strided((L * T, plane - n_blocks * bs * H * D), (plane, 1), n_blocks * bs * H * D).zero_()
    elif fill not in ("zeros", "empty"):
        raise ValueError(fill)
    return pool


def plane_bytes_of(pool: Tensor) -> int:
    """Plane stride in bytes of a 6-D pool — a contiguous tensor (the reference's) or one of allocate_kv_pool's.
    Raises if the tensor is neither (blocks of a plane not back to back, layers not T planes apart)."""
    if pool.dim() != 6:
        raise ValueError("a KV pool is 6-D: (n_layers, n_tokens, n_blocks, block_size, n_heads, head_size)")
    L, T, n_blocks, bs, H, D = pool.shape
    st = pool.stride()
    inner_ok = st[5] == 1 and st[4] == D and st[3] == H * D and st[2] == bs * H * D
    if not inner_ok or st[1] < n_blocks * bs * H * D or (L > 1 and st[0] != T * st[1]):
        raise ValueError(f"not a KV pool layout: strides {tuple(st)} for shape {tuple(pool.shape)}")
    pb = st[1] * pool.element_size()
    if pb % 16 != 0:
        raise ValueError("plane stride must be a multiple of 16 bytes")
    return pb
