"""ctypes binding of libhydra_hip.so (the C ABI declared in include/hydra_hip.h).

This is the only place the shared library is opened.  Loading fails loudly: there is
no CPU or eager-PyTorch fallback anywhere in the product path (the reference instead
wraps every `hydrainfer._C` import in try/except and silently drops to torch, e.g.
hydrainfer/layer/causal_attention.py:13-17).
"""
import ctypes
import os
from ctypes import c_char_p, c_float, c_int, c_int32, c_int64, c_void_p, POINTER

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
# HX_LIB_PATH: A/B tooling only (tools/: the same program against a library built from another revision); the product,
# the tests and bench.py load the in-tree library
LIB_PATH = os.environ.get("HX_LIB_PATH") or os.path.join(_HERE, "lib", "libhydra_hip.so")

HX_F32, HX_F16, HX_BF16 = 0, 1, 2
HX_IPC_HANDLE_BYTES = 64
HX_ABI_VERSION = 3            # include/hydra_hip.h
HX_ATTN_LOCAL_WINDOW = 1

_DTYPE = {torch.float32: HX_F32, torch.float16: HX_F16, torch.bfloat16: HX_BF16}


class HydraHipError(RuntimeError):
    """Raised for every non-zero hx_status (the reference raises RuntimeError via
    TORCH_CHECK, or aborts the process via glog CHECK; we always raise)."""


class hx_attn_args(ctypes.Structure):
    _fields_ = [
        ("out", c_void_p), ("q", c_void_p), ("k", c_void_p), ("v", c_void_p),
        ("cu_seqlens_q", c_void_p), ("cu_seqlens_k", c_void_p),
        ("block_table", c_void_p), ("cu_block_lens", c_void_p),
        ("batch", c_int32), ("n_heads", c_int32), ("n_kv_heads", c_int32),
        ("head_dim", c_int32), ("block_size", c_int32), ("max_seqlen_q", c_int32),
        ("max_seqlen_k", c_int32), ("total_q", c_int32),
        ("q_row_stride", c_int64), ("o_row_stride", c_int64),
        ("k_block_stride", c_int64), ("k_row_stride", c_int64), ("k_head_stride", c_int64),
        ("v_block_stride", c_int64), ("v_row_stride", c_int64), ("v_head_stride", c_int64),
        ("softmax_scale", c_float), ("causal", c_int32), ("dtype", c_int32),
        ("num_splits", c_int32), ("workspace", c_void_p), ("workspace_bytes", c_int64),
        ("softcap", c_float), ("window_left", c_int32), ("window_right", c_int32), ("flags", c_int32),
    ]


class hx_fused_decode_args(ctypes.Structure):
    _fields_ = [
        ("k_new", c_void_p), ("v_new", c_void_p), ("k_new_row_stride", c_int64),
        ("v_new_row_stride", c_int64), ("positions", c_void_p), ("cos_sin", c_void_p),
        ("new_cache_slots", c_void_p), ("rotary_dim", c_int32), ("interleaved", c_int32),
        ("qkv_partial", c_void_p), ("qkv_splits", c_int32), ("rank_desc", c_void_p),
    ]


class hx_decode_weight(ctypes.Structure):
    _fields_ = [("packed", c_void_p), ("N", c_int64), ("K", c_int64), ("dtype", c_int32), ("layout", c_int32),
                ("flags", c_int32), ("max_rows", c_int32)]


HX_DW_LDS_SLICE, HX_DW_XREG, HX_DW_GATE_UP = 0, 1, 1
HX_DW_FORCE_LDS_SLICE = 2


class hx_step_head_args(ctypes.Structure):
    _fields_ = [
        ("h_out", c_void_p), ("x_out", c_void_p), ("ids", c_void_p), ("feed_src", c_void_p), ("feed_prev", c_void_p),
        ("fed_out", c_void_p), ("table", c_void_p), ("weight", c_void_p), ("zero_ptr", c_void_p), ("zero_bytes", c_int64),
        ("positions", c_void_p), ("kv_lens", c_void_p), ("cu_seqlens_k", c_void_p), ("new_cache_slots", c_void_p),
        ("block_table", c_void_p), ("cu_block_lens", c_void_p),
        ("rows", c_int64), ("hidden", c_int64), ("vocab", c_int64), ("epsilon", c_float), ("ids_are_int64", c_int32),
        ("dtype", c_int32), ("batch", c_int32), ("block_size", c_int32), ("stride", c_int32), ("rank_desc", c_void_p),
    ]


HX_XREG_SYNC_WORDS = 512

_SIGNATURES = {
    "hx_abi_version": (c_int, []),
    "hx_strerror": (c_char_p, [c_int]),
    "hx_last_hip_error": (c_int, []),
    "hx_debug_set_option": (c_int, [c_char_p, c_int]),
    "hx_set_kv_cache": (c_int, [c_void_p] * 5 + [c_int64] * 8 + [c_int, c_void_p]),
    "hx_set_image_cache": (c_int, [c_void_p] * 3 + [c_int64] * 6 + [c_int, c_void_p]),
    "hx_rms_norm": (c_int, [c_void_p] * 3 + [c_float, c_int64, c_int64, c_int, c_void_p]),
    "hx_add_rms_norm": (c_int, [c_void_p] * 4 + [c_float, c_int64, c_int64, c_int, c_void_p]),
    "hx_apply_rotary_pos_emb": (c_int, [c_void_p] * 4 + [c_int64] * 7 + [c_int, c_int, c_void_p]),
    "hx_rope_set_kv_cache": (c_int, [c_void_p] * 8 + [c_int64] * 11 + [c_int, c_void_p]),
    "hx_silu": (c_int, [c_void_p] * 2 + [c_int64] * 3 + [c_int, c_void_p]),
    "hx_silu_and_mul": (c_int, [c_void_p] * 3 + [c_int64] * 4 + [c_int, c_void_p]),
    "hx_quick_gelu": (c_int, [c_void_p] * 2 + [c_int64] * 3 + [c_int, c_void_p]),
    "hx_add_layer_norm": (c_int, [c_void_p] * 5 + [c_float, c_int64, c_int64, c_int, c_void_p]),
    "hx_linear_decode_workspace_bytes": (c_int64, [c_int64] * 3),
    "hx_linear_decode": (c_int, [c_void_p] * 3 + [c_int64] * 6 + [c_void_p, c_int64, c_int, c_void_p]),
    "hx_linear_decode_partial": (c_int, [c_void_p] * 3 + [c_int64] * 6 + [c_int, c_void_p]),
    "hx_pack_decode_weight": (c_int, [c_void_p, c_void_p, c_int64, c_int64, c_int64, c_int, c_void_p]),
    "hx_linear_decode_partial_packed": (c_int, [c_void_p] * 3 + [c_int64] * 5 + [c_int, c_void_p]),
    "hx_linear_decode_xreg_supported": (c_int, [c_int64] * 3),
    "hx_linear_decode_xreg_splits": (c_int, [c_int64] * 2),
    "hx_linear_decode_xreg_workspace_bytes": (c_int64, [c_int64] * 3),
    "hx_fragment_major_elems": (c_int64, [c_int64] * 2),
    "hx_pack_decode_weight_xreg": (c_int, [c_void_p, c_void_p, c_int64, c_int64, c_int64, c_int, c_int, c_void_p]),
    "hx_linear_decode_partial_xreg": (c_int, [c_void_p] * 3 + [c_int64] * 4 + [c_int, c_int64, c_int, c_void_p]),
    "hx_embed_rms_norm": (c_int, [c_void_p] * 3 + [c_int, c_void_p, c_void_p, c_float, c_int64, c_int64, c_int64, c_int, c_void_p]),
    "hx_argmax_rows": (c_int, [c_void_p, c_void_p, c_int64, c_int64, c_int64, c_int, c_void_p]),
    "hx_norm_xreg_supported": (c_int, [c_int64] * 3 + [c_int]),
    "hx_norm_linear_decode_xreg": (c_int, [c_void_p] * 3 + [c_int32, c_void_p, c_float, c_void_p, c_void_p] + [c_int64] * 3 + [c_void_p, c_int64, c_int, c_void_p]),
    "hx_norm_gate_up_silu_xreg": (c_int, [c_void_p] * 3 + [c_int32, c_void_p, c_float, c_void_p, c_void_p] + [c_int64] * 3 + [c_void_p, c_int, c_void_p]),
    "hx_gate_up_silu_xreg_supported": (c_int, [c_int64] * 3),
    "hx_gate_up_silu_xreg": (c_int, [c_void_p] * 3 + [c_int64] * 4 + [c_int, c_int, c_void_p]),
    "hx_add_rms_norm_slabs_ex": (c_int, [c_void_p] * 3 + [c_int32, c_void_p, c_float, c_int64, c_int64, c_int, c_int, c_void_p]),
    "hx_silu_and_mul_slabs_ex": (c_int, [c_void_p] * 2 + [c_int32, c_int64, c_int64, c_int, c_int, c_void_p]),
    "hx_add_rms_norm_slabs": (c_int, [c_void_p] * 3 + [c_int32, c_void_p, c_float, c_int64, c_int64, c_int, c_void_p]),
    "hx_silu_and_mul_slabs": (c_int, [c_void_p] * 2 + [c_int32, c_int64, c_int64, c_int, c_void_p]),
    "hx_mha_varlen_fwd_workspace_bytes": (c_int64, [POINTER(hx_attn_args)]),
    "hx_mha_varlen_fwd": (c_int, [POINTER(hx_attn_args), c_void_p]),
    "hx_decode_attention_fused": (c_int, [POINTER(hx_attn_args), POINTER(hx_fused_decode_args), c_void_p]),
    "hx_ipc_get_mem_handle": (c_int, [c_void_p, c_void_p, POINTER(c_int64)]),
    "hx_ipc_open_mem_handle": (c_int, [c_void_p, POINTER(c_void_p)]),
    "hx_ipc_close_all": (c_int, []),
    "hx_migrate_blocks": (c_int, [c_void_p, c_void_p, c_int64, c_void_p, c_void_p] + [c_int64] * 5 + [c_void_p]),
    "hx_pack_blocks": (c_int, [c_void_p, c_int64, c_void_p, c_void_p] + [c_int64] * 4 + [c_void_p]),
    "hx_unpack_blocks": (c_int, [c_void_p, c_int64, c_void_p, c_void_p] + [c_int64] * 4 + [c_void_p]),
    "hx_migrate_blocks_planes": (c_int, [c_void_p, c_void_p, c_int64, c_void_p, c_void_p] + [c_int64] * 6 + [c_void_p]),
    "hx_pack_blocks_planes": (c_int, [c_void_p, c_int64, c_void_p, c_void_p] + [c_int64] * 4 + [c_void_p]),
    "hx_unpack_blocks_planes": (c_int, [c_void_p, c_int64, c_void_p, c_void_p] + [c_int64] * 4 + [c_void_p]),
    "hx_topk_softmax": (c_int, [c_void_p] * 3 + [c_int64] * 3 + [c_void_p]),
    "hx_grouped_topk_sigmoid": (c_int, [c_void_p] * 4 + [c_int64] * 5 + [c_float, c_void_p]),
    "hx_moe_sort_workspace_bytes": (c_int64, [c_int64, c_int64]),
    "hx_moe_row_id_map_from_indices": (c_int, [c_void_p, c_void_p, c_int64, c_int64, c_void_p, c_int64, c_void_p]),
    "hx_moe_row_id_map_from_mask": (c_int, [c_void_p, c_void_p, c_int64, c_int64, c_void_p, c_int64, c_void_p]),
    "hx_moe_permute": (c_int, [c_void_p] * 3 + [c_int64] * 3 + [c_int, c_void_p]),
    "hx_moe_unpermute": (c_int, [c_void_p] * 4 + [c_int64] * 3 + [c_int, c_void_p]),
    "hx_moe_sum_out": (c_int, [c_void_p] * 2 + [c_int64] * 3 + [c_int, c_void_p]),
    "hx_decode_advance": (c_int, [c_void_p] * 6 + [c_int32, c_int32, c_int32, c_void_p]),
    "hx_decode_advance_ranked": (c_int, [c_void_p] * 6 + [c_int32, c_int32, c_int32, c_void_p, c_void_p]),
    "hx_decode_rank": (c_int, [c_void_p, c_int32, c_void_p, c_void_p]),
    "hx_stage_decode": (c_int, [c_void_p, c_int64, c_void_p, c_int32, c_void_p]),
    "hx_decode_feed_ids": (c_int, [c_void_p] * 4 + [c_int32, c_void_p]),
    "hx_collect_errors": (c_int, [c_void_p, c_void_p, c_int32, c_int64, c_int32, c_void_p, c_void_p]),
    "hx_copy_words2": (c_int, [c_void_p, c_void_p, c_int32, c_void_p, c_void_p, c_int32, c_void_p]),
    "hx_decode_weight_plan": (c_int, [POINTER(hx_decode_weight), c_int64, c_int64, c_int, c_int, c_int]),
    "hx_decode_weight_pack": (c_int, [POINTER(hx_decode_weight), c_void_p, c_void_p, c_int64, c_void_p]),
    "hx_linear_decode_ex_workspace_bytes": (c_int64, [POINTER(hx_decode_weight), c_int64]),
    "hx_linear_decode_ex": (c_int, [c_void_p, c_int64, c_void_p, c_int64, c_int, POINTER(hx_decode_weight), c_int64, c_void_p]),
    "hx_plan_begin": (c_int, [POINTER(c_void_p)]),
    "hx_plan_end": (c_int, [c_void_p]),
    "hx_plan_size": (c_int, [c_void_p]),
    "hx_plan_launch": (c_int, [c_void_p, c_void_p]),
    "hx_plan_destroy": (c_int, [c_void_p]),
    "hx_memset_zero": (c_int, [c_void_p, c_int64, c_void_p]),
    "hx_decode_step_head": (c_int, [POINTER(hx_step_head_args), c_void_p]),
    "hx_gate_up_xreg_supported": (c_int, [c_int64, c_int64, c_int64, c_int]),
    "hx_gate_up_xreg_workspace_bytes": (c_int64, [c_int64, c_int64, c_int64]),
    "hx_gate_up_xreg": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_int64, c_int64, c_int64, c_int, c_int64, c_int, c_void_p]),
    "hx_norm_gate_up_xreg": (c_int, [c_void_p, c_void_p, c_void_p, c_int32, c_void_p, c_float, c_void_p, c_void_p, c_int64, c_int64,
                                     c_int64, c_void_p, c_int64, c_int, c_void_p]),
    "hx_gate_up_silu_wide_xreg_supported": (c_int, [c_int64, c_int64, c_int64]),
    "hx_norm_gate_up_silu_wide_xreg": (c_int, [c_void_p, c_void_p, c_void_p, c_int32, c_void_p, c_float, c_void_p, c_void_p,
                                               c_int64, c_int64, c_int64, c_void_p, c_int, c_void_p]),
    "hx_measure_read_stream": (c_int, [c_void_p, c_int64, c_void_p, c_void_p]),
    "hx_measure_read_grid": (c_int, [c_void_p, c_int64, c_int, c_void_p, c_void_p]),
    "hx_measure_paged_read": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_int, c_int, c_int, c_int64, c_int64, c_int, c_void_p, c_void_p]),
}

# Only in a library built with `make EXPERIMENTS=1` (include/hydra_hip_experimental.h): rejected experiments and
# microbenchmarks.  Bound when present; the product path never calls them.
_EXPERIMENTAL_SIGNATURES = {
    "hx_debug_stream_read": (c_int, [c_void_p, c_int64, c_int, c_int64, c_int, c_int, c_int, c_void_p, c_void_p]),
    "hx_debug_paged_read": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int64, c_int] + [c_int] * 4 + [c_void_p, c_void_p]),
    "hx_debug_fwd_stamps": (c_int, [c_void_p]),
}

_lib = None


def lib() -> ctypes.CDLL:
    """Open libhydra_hip.so once and type every entry point."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise HydraHipError(
                f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "or `make -C hydrainfer_amd/csrc`. There is no fallback path.")
        handle = ctypes.CDLL(LIB_PATH)
        for name, (restype, argtypes) in _SIGNATURES.items():
            fn = getattr(handle, name)  # AttributeError if the symbol is not exported
            fn.restype = restype
            fn.argtypes = argtypes
        for name, (restype, argtypes) in _EXPERIMENTAL_SIGNATURES.items():
            if hasattr(handle, name):
                fn = getattr(handle, name)
                fn.restype = restype
                fn.argtypes = argtypes
        if handle.hx_abi_version() != HX_ABI_VERSION:
            raise HydraHipError(f"libhydra_hip.so ABI version {handle.hx_abi_version()} != {HX_ABI_VERSION} of this binding: "
                                "rebuild with `make -C hydrainfer_amd/csrc`")
        # A/B runs of whole programs: HX_DEBUG_OPTIONS="decode_small=0,fwd_row_blocks=1"
        for item in filter(None, os.environ.get("HX_DEBUG_OPTIONS", "").split(",")):
            name, _, value = item.partition("=")
            if handle.hx_debug_set_option(name.strip().encode(), int(value)) != 0:
                raise HydraHipError(f"HX_DEBUG_OPTIONS: unknown option {name!r}")
        _lib = handle
    return _lib


def exported_symbols():
    return sorted(_SIGNATURES)


def has_experiments() -> bool:
    """True if libhydra_hip.so was built with `make EXPERIMENTS=1` (four-heads decode attention, read-stream probes)."""
    return hasattr(lib(), "hx_debug_stream_read")


def check(status: int, what: str) -> None:
    if status != 0:
        l = lib()
        msg = l.hx_strerror(status).decode()
        if status == -7:
            msg += f" (hipError_t {l.hx_last_hip_error()})"
        raise HydraHipError(f"{what}: {msg}")


def dtype_code(t: torch.Tensor) -> int:
    try:
        return _DTYPE[t.dtype]
    except KeyError:
        raise HydraHipError(f"failed to dispatch data type {t.dtype}")


def require_gpu(*tensors: torch.Tensor) -> None:
    """The product path is GPU-only; refuse CPU tensors instead of falling back."""
    current = None
    for t in tensors:
        if t is None:
            continue
        if not t.is_cuda:
            raise HydraHipError(
                "hydrainfer_amd ops run only on MI355X device tensors; got a CPU tensor "
                "(the CPU restatement lives in oracle/ and is test infrastructure only)")
        # ops launch on the CURRENT device's current stream (like the reference's
        # at::cuda::getCurrentCUDAStream(), kv_cache_kernels.cu:83); a tensor that lives elsewhere
        # would be touched from the wrong device's queue — the current device is per thread
        if current is None:
            current = torch.cuda.current_device()
        if t.device.index != current:
            raise HydraHipError(
                f"tensor on {t.device} but the calling thread's current device is cuda:{current}; "
                "call torch.cuda.set_device(...) (or use torch.cuda.device(...)) in this thread first")


def current_stream() -> int:
    return torch.cuda.current_stream().cuda_stream


def memset_zero(t: torch.Tensor) -> None:
    """t.zero_() as a launch the library knows about: recordable in a launch plan, capturable in a hipGraph."""
    require_gpu(t)
    if not t.is_contiguous():
        raise HydraHipError("memset_zero: contiguous tensors only")
    check(lib().hx_memset_zero(t.data_ptr(), t.numel() * t.element_size(), current_stream()), "memset_zero")
