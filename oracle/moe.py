"""oracle/moe.py — CPU restatement of the reference's MoE routing / permutation semantics.

TEST INFRASTRUCTURE ONLY (see oracle/ops.py).

Parity status.  The reference has no Python implementation of these ops (CUDA kernels only, which
need nvcc and the un-vendored cutlass submodule), so the pin is the reference's own TEST ORACLE:
tests/golden/generate_goldens.py::gen_moe runs the reference's tests/kernel/test_moe.py test
functions — their grids, their torch references (:18-21 topk_softmax_ref, :55-88 permute /
unpermute index refs, :118-141 mask refs) and their assertions — on CPU with this module standing
in for `hydrainfer._C.kernel.moe`; every call that passed is frozen in tests/golden/g12_moe.npz and
tests/test_oracle_golden.py holds this module to it.  PINNED (to the reference's test oracle):
topk_softmax, permute / unpermute with index map, permute / unpermute with mask map.
PARITY UNPINNED: grouped_topk_sigmoid — the reference has no test or torch reference for it; it is
transcribed from the kernel's control flow (csrc/kernel/moe/grouped_topk_sigmoid_kernel.cu:64-180,
including its tie-breaks) — and sum_out (align_block_kernel.cu:172-188,242-272: the running sum in scalar_t for
topk in {2, 3, 4, 8}, torch::sum_out otherwise)."""
from typing import Tuple

import torch
from torch import Tensor


def topk_softmax(gating_logits: Tensor, topk: int) -> Tuple[Tensor, Tensor]:
    """tests/kernel/test_moe.py:18-21; ties resolved to the lower index
    (topk_softmax_kernel.cu:152-157) — done here with a stable sort."""
    p = torch.softmax(gating_logits, dim=-1)
    idx = torch.argsort(p, dim=-1, descending=True, stable=True)[:, :topk]
    return torch.gather(p, 1, idx), idx.to(torch.int32)


def grouped_topk_sigmoid(logits: Tensor, bias: Tensor, n_groups: int, topk_group: int,
                         topk: int) -> Tuple[Tensor, Tensor]:
    n_tokens, n_experts = logits.shape
    per = n_experts // n_groups
    scores = 1.0 / (1.0 + torch.exp(-logits))
    weights = torch.empty((n_tokens, topk))
    indices = torch.empty((n_tokens, topk), dtype=torch.int32)
    FMAX = torch.finfo(torch.float32).max
    for t in range(n_tokens):
        choice = (scores[t] + bias).clone()
        for _ in range(n_groups - topk_group):
            best_sum, best_g = None, None
            for g in range(n_groups):
                top2 = torch.topk(choice[g * per:(g + 1) * per], min(2, per)).values
                s = float(top2[0]) + (float(top2[1]) if per > 1 else -FMAX)
                if choice[g * per] == FMAX:
                    s = float("inf")
                if best_sum is None or s < best_sum or (s == best_sum and g > best_g):
                    best_sum, best_g = s, g
            choice[best_g * per:(best_g + 1) * per] = FMAX
        for k in range(topk):
            c = choice.clone()
            c[c == FMAX] = -FMAX
            m = c.max()
            col = int(torch.nonzero(c == m)[0])   # lowest index among equals
            weights[t, k] = scores[t, col]
            indices[t, k] = col
            choice[col] = -FMAX
    return weights, indices


def permute_index(tokens: Tensor, topk_indices: Tensor) -> Tuple[Tensor, Tensor, Tensor]:
    """tests/kernel/test_moe.py:55-70.  Also returns the [topk, n_tokens] row_id_map the kernel
    produces (permutation_index_kernel.cu:56-77)."""
    n_tokens, topk = topk_indices.shape
    sorted_idx = topk_indices.reshape(-1).argsort(stable=True)
    token_idx = sorted_idx.div(topk, rounding_mode="floor")
    row_id_map = torch.empty((topk, n_tokens), dtype=torch.int32)
    p = torch.arange(n_tokens * topk, dtype=torch.int32)
    row_id_map[sorted_idx % topk, token_idx] = p
    return tokens[token_idx], sorted_idx.to(torch.int32), row_id_map


def unpermute_index(permuted: Tensor, sorted_idx: Tensor, probs: Tensor, n_tokens: int, topk: int) -> Tensor:
    """tests/kernel/test_moe.py:72-88."""
    tokens = torch.zeros_like(permuted)
    tokens[sorted_idx.long()] = permuted
    tokens = tokens.reshape(n_tokens, topk, -1) * probs[:, :, None]
    return tokens.sum(dim=1)


def permute_mask(tokens: Tensor, routing_map: Tensor) -> Tuple[Tensor, Tensor, Tensor]:
    """tests/kernel/test_moe.py:118-126 + the [n_experts, n_tokens] map of
    permutation_mask_kernel.cu:43-130 (-1 where not routed)."""
    n_tokens, n_experts = routing_map.shape
    token_idx = torch.arange(n_tokens, dtype=torch.int32)[None, :].expand(n_experts, n_tokens)
    sorted_idx = token_idx.masked_select(routing_map.t())
    row_id_map = torch.full((n_experts, n_tokens), -1, dtype=torch.int32)
    row_id_map[routing_map.t()] = torch.arange(sorted_idx.numel(), dtype=torch.int32)
    return tokens[sorted_idx.long()], sorted_idx, row_id_map


def unpermute_mask(permuted: Tensor, permuted_probs: Tensor, sorted_idx: Tensor, n_tokens: int) -> Tensor:
    """tests/kernel/test_moe.py:128-141."""
    dim = permuted.shape[1]
    tokens = torch.zeros((n_tokens, dim), dtype=permuted.dtype)
    tokens.scatter_add_(0, sorted_idx[:, None].expand(-1, dim).to(torch.int64),
                        permuted * permuted_probs[:, None])
    return tokens


def unpermute_rows(permuted: Tensor, row_id_map: Tensor, probs: Tensor) -> Tensor:
    """Kernel-side unpermute for both map kinds (permutation_index_kernel.cu:146-160,
    permutation_mask_kernel.cu): out[t] = sum_r probs[t, r] * permuted[row_id_map[r, t]] over
    entries >= 0; product and running sum in the tokens' dtype like the kernel's frag_sum."""
    n_rows, n_tokens = row_id_map.shape
    dt = permuted.dtype
    out = torch.zeros((n_tokens, permuted.shape[1]), dtype=dt)
    for r in range(n_rows):
        idx = row_id_map[r].long()
        ok = idx >= 0
        w = probs[:, r].to(dt)[:, None] if probs is not None else 1
        contrib = (permuted[idx.clamp_min(0)] * w).to(dt)
        out = torch.where(ok[:, None], (out + contrib).to(dt), out)
    return out


def sum_out(x: Tensor) -> Tensor:
    """align_block_kernel.cu:242-272.  x [n_tokens, topk, dim] -> [n_tokens, dim].  topk in {2, 3, 4, 8} run
    topk_sum_kernel (:172-188): `scalar_t sum = 0; sum += input[k]` — the running sum lives in scalar_t, so EVERY
    partial sum is rounded to the tensor's dtype; any other topk falls to torch::sum_out (fp32 accumulation, one
    rounding).  PARITY UNPINNED (no reference-side test or torch form exists for this op)."""
    n_tokens, topk, dim = x.shape
    if topk in (2, 3, 4, 8):
        acc = torch.zeros((n_tokens, dim), dtype=x.dtype)
        for k in range(topk):
            acc = (acc.float() + x[:, k].float()).to(x.dtype)      # one T rounding per add (exact for fp32)
        return acc
    return torch.sum(x, dim=1)
