"""CPU: the KV pool's layout in memory (hydrainfer_amd/memory/kv_pool.py) — the reference's 6-D shape
(hydrainfer/memory/token_cache_manger.py:65) with the (layer, k/v) planes a fixed number of bytes apart."""
import pytest
import torch

from hydrainfer_amd.memory import kv_pool
from hydrainfer_amd.memory.token_cache_manger import ipc_safe_n_blocks


def test_pool_has_the_reference_shape_and_contiguous_layer_views():
    shape = (3, 2, 5, 4, 2, 8)
    p = kv_pool.allocate_kv_pool(shape, torch.bfloat16, "cpu", fill="randn")
    assert tuple(p.shape) == shape and not p.is_contiguous()
    plane = 5 * 4 * 2 * 8 + kv_pool.KV_POOL_SKEW_BYTES // 2
    assert p.stride() == (2 * plane, plane, 64, 16, 8, 1)
    assert kv_pool.plane_bytes_of(p) == plane * 2
    for l in range(3):
        for t in range(2):
            v = p[l, t]
            assert v.is_contiguous() and v.data_ptr() == p.data_ptr() + (l * 2 + t) * plane * 2
    assert bool(torch.isfinite(p.float()).all())
    # K and V of the same (block, token, head) differ in address bit 8: an ODD multiple of 256 bytes apart beyond the
    # n_blocks * block_bytes of the contiguous pool
    assert (kv_pool.KV_POOL_SKEW_BYTES // 256) % 2 == 1 and kv_pool.KV_POOL_SKEW_BYTES % 256 == 0


def test_indexing_a_pool_is_indexing_the_reference_tensor():
    g = torch.Generator().manual_seed(0)
    ref = torch.randn((2, 2, 6, 4, 2, 8), generator=g)
    p = kv_pool.allocate_kv_pool(tuple(ref.shape), torch.float32, "cpu", fill="empty")
    p.copy_(ref)
    assert torch.equal(p, ref) and torch.equal(p[:, :, [4, 1]], ref[:, :, [4, 1]])
    p[:, :, [0, 3]] = ref[:, :, [5, 2]]
    assert torch.equal(p[1, 0, 3], ref[1, 0, 2])
    assert torch.equal(p.clone(), p) and torch.equal(p.cpu().contiguous()[0, 1], p[0, 1])


def test_plane_bytes_of_accepts_the_two_layouts_only():
    c = torch.zeros((3, 2, 5, 4, 2, 8), dtype=torch.float16)
    assert kv_pool.plane_bytes_of(c) == 5 * 4 * 2 * 8 * 2                       # the reference's contiguous pool
    assert kv_pool.plane_bytes_of(kv_pool.allocate_kv_pool(tuple(c.shape), c.dtype, "cpu", fill="zeros", skew_bytes=0)) == 5 * 4 * 2 * 8 * 2
    with pytest.raises(ValueError):
        kv_pool.plane_bytes_of(c.transpose(3, 4))
    with pytest.raises(ValueError):
        kv_pool.plane_bytes_of(c[:, :, ::2])
    with pytest.raises(ValueError):
        kv_pool.plane_bytes_of(c[0])
    with pytest.raises(ValueError):
        kv_pool.allocate_kv_pool(tuple(c.shape), c.dtype, "cpu", skew_bytes=40)   # not a multiple of 16


def test_ipc_window_counts_the_spare_bytes():
    """ipc_safe_n_blocks sizes the ALLOCATION out of [7/8 * 2^k, 2^k): the spare bytes between planes belong to it."""
    bpb = 8 << 20                                   # LLaVA-1.5-7B: 32 layers x 2 x 128 KiB per block
    assert ipc_safe_n_blocks(1920, bpb) == 2048     # 15 GiB -> 16 GiB
    extra = 64 * 768
    n = ipc_safe_n_blocks(1920, bpb, extra_bytes=extra)
    size = n * bpb + extra
    p2 = 1 << (size - 1).bit_length()
    assert n >= 1920 and not (size < p2 and size * 8 >= p2 * 7)
    # just under the window without the spare bytes, inside it with them
    n0 = (7 * (1 << 30) // 8) // 4096
    assert ipc_safe_n_blocks(n0 - 1, 4096) == n0 - 1
    n1 = ipc_safe_n_blocks(n0 - 1, 4096, extra_bytes=8192)
    s1 = n1 * 4096 + 8192
    assert not (s1 < (1 << 30) and s1 * 8 >= (1 << 30) * 7)
