"""oracle/ops.py — CPU restatement of the reference's algorithm for the attention +
paged-KV hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under hydrainfer_amd/ imports this package; only
tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may, and only as the
checker / reported baseline, never as the product path.

Parity status: PINNED.  Every function here is checked bit-for-bit (integer / copy
ops) or to fp32 round-off (floating ops) against outputs of the reference's own Python
handlers, produced by tests/golden/generate_goldens.py importing /root/reference in the
build container and committed as tests/golden/*.npz (see tests/test_oracle_golden.py).
The reference's CUDA kernels cannot be built here (nvcc + un-vendored cutlass,
SURVEY.md §8c); where the CUDA kernel and the torch fallback round differently
(rms_norm, P in attention) both variants are restated and named.

All citations are relative to the dongxianzhe/hydrainfer tree.
"""
import math
from typing import List, Optional, Tuple

import torch
from torch import Tensor


# ---------------------------------------------------------------------------
# paged cache scatter
# ---------------------------------------------------------------------------
def set_kv_cache(slot_ids: Tensor, keys: Tensor, values: Tensor, key_cache: Tensor,
                 value_cache: Tensor) -> None:
    """hydrainfer/memory/kv_cache.py:44-50 (per-token loop) ==
    csrc/kernel/kv_cache_kernels/kv_cache_kernels.cu:16-58: cache[slot // bs, slot % bs] = row.
    Vectorised; later duplicates of a slot win, as in the sequential loop."""
    block_size = key_cache.shape[1]
    slots = slot_ids.to(torch.int64)
    for i in range(slots.shape[0]):
        s = int(slots[i])
        b, o = s // block_size, s % block_size
        key_cache[b, o] = keys[i]
        value_cache[b, o] = values[i]


def set_image_cache(slot_ids: Tensor, image_tokens: Tensor, image_cache: Tensor) -> None:
    """hydrainfer/memory/token_cache.py:53-56: slot_view[slot_ids] = value."""
    view = image_cache.view(-1, image_cache.shape[-2], image_cache.shape[-1])
    view[slot_ids.to(torch.int64)] = image_tokens


# ---------------------------------------------------------------------------
# rms_norm
# ---------------------------------------------------------------------------
def rms_norm_torch(h: Tensor, w: Tensor, eps: float) -> Tensor:
    """The reference's CPU path, hydrainfer/layer/norm.py:18-23, verbatim semantics:
    rms in fp32; h / rms and * w promote to fp32; one final cast."""
    dtype = h.dtype
    rms = torch.sqrt(torch.mean(h.to(torch.float) ** 2, dim=-1, keepdim=True) + eps)
    return ((h / rms) * w).to(dtype)


def rms_norm_kernel(h: Tensor, w: Tensor, eps: float) -> Tensor:
    """Rounding points of the CUDA kernel, csrc/kernel/norm/rms_norm.cu:27-40:
    fp32 sum of squares, s = rsqrt(sum/n + eps), out = (T)(x*s) * w with the last multiply
    in T arithmetic."""
    dtype = h.dtype
    x = h.to(torch.float)
    s = torch.rsqrt(torch.mean(x * x, dim=-1, keepdim=True) + eps)
    n = (x * s).to(dtype)
    return (n.to(torch.float) * w.to(torch.float)).to(dtype) if dtype != torch.float else n * w


# ---------------------------------------------------------------------------
# rotary embedding
# ---------------------------------------------------------------------------
def build_cos_sin_cache(rotary_dim: int, max_position_embeddings: int, theta: float,
                        dtype: torch.dtype) -> Tensor:
    """hydrainfer/layer/rotary_embedding.py:13-17,111-116: [max_pos, 2, rotary_dim/2],
    computed in fp32 then cast (model.to(dtype), hydrainfer/model/llava.py:125-126)."""
    inv_freq = 1.0 / torch.pow(theta, torch.arange(0, rotary_dim, 2, dtype=torch.float) / rotary_dim)
    t = torch.arange(max_position_embeddings, dtype=torch.float)
    freqs = torch.einsum("i,j->ij", t, inv_freq)
    return torch.cat([freqs.cos()[:, None, :], freqs.sin()[:, None, :]], dim=1).to(dtype)


def apply_rotary_pos_emb(query: Tensor, key: Tensor, positions: Tensor, cos_sin: Tensor,
                         rotary_dim: int, interleaved: bool) -> Tuple[Tensor, Tensor]:
    """csrc/kernel/position_embedding/rope.cu:12-79: x' = x*c - y*s ; y' = x*s + y*c with
    every operation in T arithmetic (torch CPU rounds each half/bf16 op, which is exact T
    arithmetic).  Equal, operation for operation, to TorchRotaryEmbeddingHandler
    (hydrainfer/layer/rotary_embedding.py:46-83) when its cos/sin cache is in T.
    Returns new tensors (the kernel works in place)."""
    half = rotary_dim // 2
    cs = cos_sin.view(cos_sin.shape[0], 2, half)[positions.to(torch.int64)]  # [n, 2, half]
    c = cs[:, 0, None, :]
    s = cs[:, 1, None, :]

    def rot(t: Tensor) -> Tensor:
        out = t.clone()
        r = t[..., :rotary_dim]
        if interleaved:
            x, y = r[..., 0::2], r[..., 1::2]
        else:
            x, y = r[..., :half], r[..., half:]
        xo = x * c - y * s
        yo = x * s + y * c
        if interleaved:
            out[..., 0:rotary_dim:2] = xo
            out[..., 1:rotary_dim:2] = yo
        else:
            out[..., :half] = xo
            out[..., half:rotary_dim] = yo
        return out

    return rot(query), rot(key)


# ---------------------------------------------------------------------------
# activation
# ---------------------------------------------------------------------------
def silu(x: Tensor) -> Tensor:
    """hydrainfer/layer/activation.py:24-29 CPU branch == F.silu; the CUDA kernel
    (csrc/kernel/activation/activation.cu:16-19) is (T)(x / (1 + exp(-x))) in fp32."""
    return torch.nn.functional.silu(x)


def silu_kernel(x: Tensor) -> Tensor:
    """Formula of the CUDA kernel, csrc/kernel/activation/activation.cu:16-19, in fp32."""
    xf = x.to(torch.float)
    return (xf / (1.0 + torch.exp(-xf))).to(x.dtype)


def silu_and_mul(gate: Tensor, up: Tensor) -> Tensor:
    """hydrainfer/model/model_forward.py:36: activation(gate) * up, product in T."""
    return silu(gate) * up


# ---------------------------------------------------------------------------
# attention
# ---------------------------------------------------------------------------
def paged_attention(query: Tensor, key_cache: Tensor, value_cache: Tensor, q_cu_seq_lens: Tensor,
                    kv_cu_seq_lens: Tensor, block_tables: Tensor, cu_blocks_lens: Tensor,
                    causal: bool = True, sm_scale: Optional[float] = None, softcap: float = 0.0,
                    window: Optional[tuple] = None) -> Tensor:
    """TorchCausalGroupedQueryPageAttentionHandler.forward,
    hydrainfer/layer/causal_attention.py:307-374: per sequence gather pages, fp32 math,
    GQA by repeat_interleave, bottom-right aligned causal mask `(x - y) > (k_len - q_len)`,
    softmax, PV, cast to the query dtype.  query [n_tokens, H, D] -> [n_tokens, H, D]."""
    n_tokens, n_heads, head_dim = query.shape
    n_kv_heads = key_cache.shape[2]
    if sm_scale is None:
        sm_scale = 1.0 / math.sqrt(head_dim)
    outs = []
    n_seq = q_cu_seq_lens.numel() - 1
    for i in range(n_seq):
        bt = block_tables[int(cu_blocks_lens[i]): int(cu_blocks_lens[i + 1])].to(torch.int64)
        kv_len = int(kv_cu_seq_lens[i + 1]) - int(kv_cu_seq_lens[i])
        k = key_cache[bt].reshape(-1, n_kv_heads, head_dim)[:kv_len].to(torch.float)
        v = value_cache[bt].reshape(-1, n_kv_heads, head_dim)[:kv_len].to(torch.float)
        q = query[int(q_cu_seq_lens[i]): int(q_cu_seq_lens[i + 1])].to(torch.float)
        outs.append(_attend(q, k, v, sm_scale, causal, softcap=softcap, window=window))
    return torch.cat(outs, dim=0).to(query.dtype)


def varlen_attention(query: Tensor, key: Tensor, value: Tensor, cu_seqlens_q: Tensor,
                     cu_seqlens_k: Tensor, causal: bool, sm_scale: Optional[float] = None,
                     softcap: float = 0.0, window: Optional[tuple] = None) -> Tensor:
    """Dense varlen attention: TorchMultiHeadAttentionHandler.forward,
    hydrainfer/layer/multihead_attention.py:46-70 (non-causal, fp32 softmax(QK^T/sqrt(d))V)
    generalised to ragged cu_seqlens as mha_varlen_fwd's dense path takes them
    (multihead_attention.py:131-160)."""
    head_dim = query.shape[-1]
    if sm_scale is None:
        sm_scale = 1.0 / math.sqrt(head_dim)
    outs = []
    for i in range(cu_seqlens_q.numel() - 1):
        q = query[int(cu_seqlens_q[i]): int(cu_seqlens_q[i + 1])].to(torch.float)
        k = key[int(cu_seqlens_k[i]): int(cu_seqlens_k[i + 1])].to(torch.float)
        v = value[int(cu_seqlens_k[i]): int(cu_seqlens_k[i + 1])].to(torch.float)
        # multihead_attention.py:59-60 scales the query before the product
        outs.append(_attend(q, k, v, sm_scale, causal, scale_query_first=True, softcap=softcap, window=window))
    return torch.cat(outs, dim=0).to(query.dtype)


def _attend(q: Tensor, k: Tensor, v: Tensor, sm_scale: float, causal: bool,
            scale_query_first: bool = False, softcap: float = 0.0, window: Optional[tuple] = None) -> Tensor:
    """softcap / window are NOT in the reference's torch handlers (parity unpinned for them): they
    restate the CUDA kernel's semantics — scores = softcap * tanh(q.k * scale / softcap) before
    masking (flash_api.cpp:93-97, utils.h:383-388); local window (left, right), -1 = unbounded on
    that side: query row y of lq sees keys x with y + lk - lq - left <= x <= y + lk - lq + right
    (mask.h:173-193, flash_api.cpp:103-107).  Rows that see no key give zeros."""
    # q [Lq, H, D]; k, v [Lk, HK, D] fp32
    group = q.shape[1] // k.shape[1]
    k = k.repeat_interleave(group, dim=1)
    v = v.repeat_interleave(group, dim=1)
    if scale_query_first:
        scores = torch.einsum("qhd,khd->hqk", q * sm_scale, k)
    else:
        scores = torch.einsum("qhd,khd->hqk", q, k) * sm_scale
    if softcap > 0:
        scores = softcap * torch.tanh(scores / softcap)
    lq, lk = q.shape[0], k.shape[0]
    x = torch.arange(lk)[None, None, :]
    y = torch.arange(lq)[None, :, None]
    if causal:
        scores = scores.masked_fill((x - y) > (lk - lq), float("-inf"))
    if window is not None and (window[0] >= 0 or window[1] >= 0):
        left = window[0] if window[0] >= 0 else lk
        right = window[1] if window[1] >= 0 else lk
        d = x - y - (lk - lq)
        scores = scores.masked_fill((d > right) | (d < -left), float("-inf"))
    p = torch.softmax(scores, dim=-1)
    p = torch.nan_to_num(p, nan=0.0)          # a row whose window holds no key
    return torch.einsum("hqk,khd->qhd", p, v)


# ---------------------------------------------------------------------------
# block migration (index semantics only)
# ---------------------------------------------------------------------------
def migrate_blocks(src_block_table: List[int], dst_block_table: List[int], src_cache: Tensor,
                   dst_cache: Tensor) -> None:
    """csrc/data_transfer/block_migration.cpp:222-244:
    dst[l, t, dst_tbl[i]] = src[l, t, src_tbl[i]] for all layers l, token kinds t; the two
    pools may have different n_blocks."""
    assert len(src_block_table) == len(dst_block_table)
    for s, d in zip(src_block_table, dst_block_table):
        dst_cache[:, :, d] = src_cache[:, :, s]
