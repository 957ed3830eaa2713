"""hydrainfer._C.kernel.moe — drop-in surface
(reference stub: hydrainfer/_C/kernel/moe/__init__.pyi:4-145; pybind list
csrc/kernel/moe/moe_kernels_pybind.cpp:9-15)."""
from typing import Tuple

import torch
from torch import Tensor

from hydrainfer_amd import _lib


def _f32(t: Tensor, name: str) -> Tensor:
    if t.dtype != torch.float32:
        raise _lib.HydraHipError(f"{name} must be float32")
    if not t.is_contiguous():
        raise _lib.HydraHipError(f"{name} must be contiguous")
    return t


def topk_softmax(gating_logits: Tensor, topk_weights: Tensor, topk_indices: Tensor) -> None:
    _lib.require_gpu(gating_logits, topk_weights, topk_indices)
    _f32(gating_logits, "gating_logits"); _f32(topk_weights, "topk_weights")
    if topk_indices.dtype != torch.int32 or not topk_indices.is_contiguous():
        raise _lib.HydraHipError("topk_indices must be contiguous int32")
    n_tokens, n_experts = gating_logits.shape
    topk = topk_weights.shape[1]
    if topk_weights.shape != (n_tokens, topk) or topk_indices.shape != (n_tokens, topk):
        raise _lib.HydraHipError("topk_softmax: output shape mismatch")
    _lib.check(_lib.lib().hx_topk_softmax(gating_logits.data_ptr(), topk_weights.data_ptr(),
                                          topk_indices.data_ptr(), n_tokens, n_experts, topk,
                                          _lib.current_stream()), "topk_softmax")


def grouped_topk_sigmoid(gating_logits: Tensor, correction_bias: Tensor, n_expert_groups: int,
                         topk_group: int, topk: int, scaling_factor: float, topk_weights: Tensor,
                         topk_indices: Tensor) -> None:
    _lib.require_gpu(gating_logits, correction_bias, topk_weights, topk_indices)
    _f32(gating_logits, "gating_logits"); _f32(correction_bias, "correction_bias")
    _f32(topk_weights, "topk_weights")
    if topk_indices.dtype != torch.int32 or not topk_indices.is_contiguous():
        raise _lib.HydraHipError("topk_indices must be contiguous int32")
    n_tokens, n_experts = gating_logits.shape
    _lib.check(_lib.lib().hx_grouped_topk_sigmoid(
        gating_logits.data_ptr(), correction_bias.data_ptr(), topk_weights.data_ptr(),
        topk_indices.data_ptr(), n_tokens, n_experts, int(n_expert_groups), int(topk_group),
        int(topk), float(scaling_factor), _lib.current_stream()), "grouped_topk_sigmoid")


def _permute(tokens: Tensor, row_id_map: Tensor, n_rows: int, n_out_rows: int) -> Tensor:
    n_tokens, dim = tokens.shape
    permuted = torch.empty((n_out_rows, dim), dtype=tokens.dtype, device=tokens.device)
    _lib.check(_lib.lib().hx_moe_permute(tokens.data_ptr(), permuted.data_ptr(), row_id_map.data_ptr(),
                                         n_tokens, n_rows, dim, _lib.dtype_code(tokens),
                                         _lib.current_stream()), "moe permute")
    return permuted


def _unpermute(permuted: Tensor, row_id_map: Tensor, probs: Tensor) -> Tensor:
    n_rows, n_tokens = row_id_map.shape
    dim = permuted.shape[1]
    if probs.dtype != permuted.dtype:
        probs = probs.to(permuted.dtype)
    probs = probs.contiguous()
    out = torch.empty((n_tokens, dim), dtype=permuted.dtype, device=permuted.device)
    _lib.check(_lib.lib().hx_moe_unpermute(permuted.data_ptr(), out.data_ptr(), row_id_map.data_ptr(),
                                           probs.data_ptr(), n_tokens, n_rows, dim,
                                           _lib.dtype_code(permuted), _lib.current_stream()),
               "moe unpermute")
    return out


def permute_with_index_map(tokens: Tensor, topk_ids: Tensor) -> Tuple[Tensor, Tensor]:
    """Returns (permuted_tokens [n_tokens*topk, dim], row_id_map [topk, n_tokens])."""
    _lib.require_gpu(tokens, topk_ids)
    if tokens.dim() != 2 or topk_ids.dim() != 2 or not tokens.is_contiguous():
        raise _lib.HydraHipError("permute_with_index_map: tokens [n, dim] contiguous, topk_ids [n, topk]")
    topk_ids = topk_ids.to(torch.int32).contiguous()
    n_tokens, topk = topk_ids.shape
    l = _lib.lib()
    ws_bytes = l.hx_moe_sort_workspace_bytes(n_tokens, topk)
    ws = torch.empty(max(ws_bytes, 1), dtype=torch.uint8, device=tokens.device)
    row_id_map = torch.empty((topk, n_tokens), dtype=torch.int32, device=tokens.device)
    _lib.check(l.hx_moe_row_id_map_from_indices(topk_ids.data_ptr(), row_id_map.data_ptr(), n_tokens,
                                                topk, ws.data_ptr(), ws_bytes, _lib.current_stream()),
               "moe row_id_map (index)")
    return _permute(tokens, row_id_map, topk, n_tokens * topk), row_id_map


def unpermute_with_index_map(permuted_tokens: Tensor, row_id_map: Tensor, probs: Tensor) -> Tensor:
    _lib.require_gpu(permuted_tokens, row_id_map, probs)
    return _unpermute(permuted_tokens.contiguous(), row_id_map.contiguous(), probs)


def permute_with_mask_map(tokens: Tensor, routing_map: Tensor, topk: int) -> Tuple[Tensor, Tensor]:
    """Returns (permuted_tokens [n_tokens*topk, dim], row_id_map [n_experts, n_tokens])."""
    _lib.require_gpu(tokens, routing_map)
    if routing_map.dtype != torch.bool:
        raise _lib.HydraHipError("routing_map must be a bool tensor")
    if tokens.dim() != 2 or not tokens.is_contiguous():
        raise _lib.HydraHipError("tokens must be contiguous [n_tokens, dim]")
    routing_map = routing_map.contiguous()
    n_tokens, n_experts = routing_map.shape
    row_id_map = torch.empty((n_experts, n_tokens), dtype=torch.int32, device=tokens.device)
    ws = torch.empty(n_experts, dtype=torch.int32, device=tokens.device)
    _lib.check(_lib.lib().hx_moe_row_id_map_from_mask(routing_map.data_ptr(), row_id_map.data_ptr(),
                                                      n_tokens, n_experts, ws.data_ptr(), n_experts * 4,
                                                      _lib.current_stream()), "moe row_id_map (mask)")
    return _permute(tokens, row_id_map, n_experts, n_tokens * int(topk)), row_id_map


def unpermute_with_mask_map(permuted_tokens: Tensor, row_id_map: Tensor, probs: Tensor) -> Tensor:
    _lib.require_gpu(permuted_tokens, row_id_map, probs)
    return _unpermute(permuted_tokens.contiguous(), row_id_map.contiguous(), probs)


def sum_out(input: Tensor, output: Tensor) -> None:
    _lib.require_gpu(input, output)
    if input.dim() != 3 or not input.is_contiguous() or not output.is_contiguous():
        raise _lib.HydraHipError("sum_out: input [n_tokens, topk, dim] and output must be contiguous")
    n_tokens, topk, dim = input.shape
    if output.shape != (n_tokens, dim) or output.dtype != input.dtype:
        raise _lib.HydraHipError("sum_out: output shape / dtype mismatch")
    _lib.check(_lib.lib().hx_moe_sum_out(input.data_ptr(), output.data_ptr(), n_tokens, topk, dim,
                                         _lib.dtype_code(input), _lib.current_stream()), "sum_out")
