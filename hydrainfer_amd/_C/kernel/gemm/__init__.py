"""Extension op (no counterpart in hydrainfer._C): decode-batch linear layer on the
weight-streaming HIP kernel (csrc/gemm_skinny.hip)."""
import ctypes
from typing import Optional

import torch
from torch import Tensor

from hydrainfer_amd import _lib


def supported(x: Tensor, weight: Tensor) -> bool:
    M, K = x.shape
    N = weight.shape[0]
    return (x.dtype in (torch.float16, torch.bfloat16) and 1 <= M <= 64 and N % 16 == 0 and K % 256 == 0
            and x.stride(1) == 1 and weight.stride(1) == 1 and x.stride(0) % 8 == 0 and weight.stride(0) % 8 == 0)


def linear_decode(x: Tensor, weight: Tensor, out: Optional[Tensor] = None) -> Tensor:
    """out[M, N] = x[M, K] @ weight[N, K]^T, M <= 64."""
    _lib.require_gpu(x, weight)
    if x.dim() != 2 or weight.dim() != 2 or x.shape[1] != weight.shape[1] or x.dtype != weight.dtype:
        raise _lib.HydraHipError("linear_decode: x [M, K], weight [N, K], same dtype")
    if not supported(x, weight):
        raise _lib.HydraHipError("linear_decode: needs M <= 64, N % 16 == 0, K % 256 == 0, fp16/bf16, contiguous rows")
    M, K = x.shape
    N = weight.shape[0]
    if out is None:
        out = torch.empty((M, N), dtype=x.dtype, device=x.device)
    l = _lib.lib()
    nbytes = l.hx_linear_decode_workspace_bytes(M, N, K)
    ws = torch.empty(nbytes, dtype=torch.uint8, device=x.device)
    _lib.check(l.hx_linear_decode(out.data_ptr(), x.data_ptr(), weight.data_ptr(), M, N, K, x.stride(0),
                                  weight.stride(0), out.stride(0), ws.data_ptr(), nbytes,
                                  _lib.dtype_code(x), _lib.current_stream()), "linear_decode")
    return out


def workspace_floats(M: int, N: int, K: int) -> int:
    return _lib.lib().hx_linear_decode_workspace_bytes(M, N, K) // 4


def linear_decode_partial(x: Tensor, weight: Tensor, partial: Tensor) -> int:
    """GEMM only: fp32 split-K slabs [n_splits, M, N] are written into `partial` (a float32
    buffer of at least workspace_floats(M, N, K) elements) for a fused consumer.  Returns n_splits."""
    _lib.require_gpu(x, weight, partial)
    if not supported(x, weight) or x.dtype != weight.dtype:
        raise _lib.HydraHipError("linear_decode_partial: needs M <= 64, N % 16 == 0, K % 256 == 0, fp16/bf16")
    if partial.dtype != torch.float32 or not partial.is_contiguous():
        raise _lib.HydraHipError("linear_decode_partial: partial must be contiguous float32")
    M, K = x.shape
    N = weight.shape[0]
    rc = _lib.lib().hx_linear_decode_partial(partial.data_ptr(), x.data_ptr(), weight.data_ptr(), M, N, K,
                                             x.stride(0), weight.stride(0), partial.numel() * 4,
                                             _lib.dtype_code(x), _lib.current_stream())
    if rc < 0:
        _lib.check(rc, "linear_decode_partial")
    return rc


# ------------------------------------------------------------------------------------------------
def pack_weight(weight: Tensor) -> Tensor:
    """weight [N, K] -> flat [N*K] in MFMA-fragment order (hx_pack_decode_weight)."""
    _lib.require_gpu(weight)
    N, K = weight.shape
    if weight.dtype not in (torch.float16, torch.bfloat16) or N % 16 or K % 256 or weight.stride(1) != 1:
        raise _lib.HydraHipError("pack_weight: fp16/bf16 [N % 16 == 0, K % 256 == 0], contiguous rows")
    packed = torch.empty(N * K, dtype=weight.dtype, device=weight.device)
    _lib.check(_lib.lib().hx_pack_decode_weight(packed.data_ptr(), weight.data_ptr(), N, K, weight.stride(0),
                                                _lib.dtype_code(weight), _lib.current_stream()), "pack_weight")
    return packed


def linear_decode_partial_packed(x: Tensor, packed: Tensor, N: int, partial: Tensor) -> int:
    """As linear_decode_partial with `packed` = pack_weight(weight [N, K])."""
    _lib.require_gpu(x, packed, partial)
    M, K = x.shape
    if packed.numel() != N * K or packed.dtype != x.dtype or not packed.is_contiguous() or x.stride(1) != 1:
        raise _lib.HydraHipError("linear_decode_partial_packed: packed must be pack_weight(weight [N, K]) of x's dtype")
    if partial.dtype != torch.float32 or not partial.is_contiguous():
        raise _lib.HydraHipError("linear_decode_partial_packed: partial must be contiguous float32")
    rc = _lib.lib().hx_linear_decode_partial_packed(partial.data_ptr(), x.data_ptr(), packed.data_ptr(), M, N, K,
                                                    x.stride(0), partial.numel() * 4, _lib.dtype_code(x),
                                                    _lib.current_stream())
    if rc < 0:
        _lib.check(rc, "linear_decode_partial_packed")
    return rc


# ------------------------------------------------------------------------------------------------
# activations-in-registers kernel (csrc/gemm_xreg.hip): M <= 32 with one slab for K <= 4096; M <= 64 ("wide") with two
# ------------------------------------------------------------------------------------------------
def xreg_supported(M: int, N: int, K: int, dtype: torch.dtype) -> bool:
    return dtype in (torch.float16, torch.bfloat16) and _lib.lib().hx_linear_decode_xreg_supported(M, N, K) == 1


def gate_up_silu_supported(M: int, inter: int, K: int, dtype: torch.dtype) -> bool:
    return dtype in (torch.float16, torch.bfloat16) and _lib.lib().hx_gate_up_silu_xreg_supported(M, inter, K) == 1


def xreg_workspace_floats(M: int, N: int, K: int) -> int:
    return _lib.lib().hx_linear_decode_xreg_workspace_bytes(M, N, K) // 4


def fragment_major_elems(rows: int, K: int) -> int:
    """Elements of the fragment-major form of a [rows, K] activation (rows padded to 16)."""
    return (rows + 15) // 16 * 16 * K


def to_fragment_major(x: Tensor) -> Tensor:
    """[M, K] -> flat fragment-major tensor (torch ops; tests and tools — the product path gets this
    layout straight from the producing kernels)."""
    M, K = x.shape
    MB = (M + 15) // 16
    xp = torch.zeros((MB * 16, K), dtype=x.dtype, device=x.device)
    xp[:M] = x
    return xp.view(MB, 16, K // 32, 4, 8).permute(2, 0, 3, 1, 4).contiguous().view(-1)


def from_fragment_major(flat: Tensor, M: int, K: int) -> Tensor:
    MB = (M + 15) // 16
    return flat[: MB * 16 * K].view(K // 32, MB, 4, 16, 8).permute(1, 3, 0, 2, 4).reshape(MB * 16, K)[:M]


def pack_weight_xreg(weight: Tensor, interleave_halves: bool = False) -> Tensor:
    """weight [N, K] -> flat [N*K] in the block order of the activations-in-registers kernel;
    interleave_halves: gate|up weight for gate_up_silu_xreg."""
    _lib.require_gpu(weight)
    N, K = weight.shape
    if weight.dtype not in (torch.float16, torch.bfloat16) or N % 16 or K % 32 or weight.stride(1) != 1:
        raise _lib.HydraHipError("pack_weight_xreg: fp16/bf16 [N % 16 == 0, K % 32 == 0], contiguous rows")
    packed = torch.empty(N * K, dtype=weight.dtype, device=weight.device)
    _lib.check(_lib.lib().hx_pack_decode_weight_xreg(packed.data_ptr(), weight.data_ptr(), N, K, weight.stride(0),
                                                     1 if interleave_halves else 0, _lib.dtype_code(weight),
                                                     _lib.current_stream()), "pack_weight_xreg")
    return packed


def _x_args(x: Tensor, frag_shape):
    if frag_shape is None:
        if x.dim() != 2 or x.stride(1) != 1:
            raise _lib.HydraHipError("xreg: x must be [M, K] with contiguous rows (or fragment-major with frag_shape)")
        return x.shape[0], x.shape[1], x.stride(0), 0
    M, K = frag_shape
    if not x.is_contiguous() or x.numel() < fragment_major_elems(M, K):
        raise _lib.HydraHipError("xreg: fragment-major x needs ceil16(M) * K contiguous elements")
    return M, K, K, 1


def linear_decode_partial_xreg(x: Tensor, packed: Tensor, N: int, partial: Tensor, frag_shape=None) -> int:
    """fp32 slabs [n_splits, M, N] of x[M <= 32, K] @ W^T with `packed` = pack_weight_xreg(W [N, K]).
    x is [M, K], or the flat fragment-major form with frag_shape = (M, K).  Returns n_splits."""
    _lib.require_gpu(x, packed, partial)
    M, K, ldx, fm = _x_args(x, frag_shape)
    if packed.numel() != N * K or packed.dtype != x.dtype or not packed.is_contiguous():
        raise _lib.HydraHipError("linear_decode_partial_xreg: packed must be pack_weight_xreg(weight [N, K]) of x's dtype")
    if partial.dtype != torch.float32 or not partial.is_contiguous():
        raise _lib.HydraHipError("linear_decode_partial_xreg: partial must be contiguous float32")
    rc = _lib.lib().hx_linear_decode_partial_xreg(partial.data_ptr(), x.data_ptr(), packed.data_ptr(), M, N, K,
                                                  ldx, fm, partial.numel() * 4, _lib.dtype_code(x),
                                                  _lib.current_stream())
    if rc < 0:
        _lib.check(rc, "linear_decode_partial_xreg")
    return rc


def gate_up_silu_xreg(x: Tensor, packed_gate_up: Tensor, inter: int, act: Tensor, frag_shape=None) -> None:
    """act (flat, fragment-major [M, inter]) = silu(x Wg^T) * (x Wu^T), one launch;
    packed_gate_up = pack_weight_xreg(W [2*inter, K], interleave_halves=True)."""
    _lib.require_gpu(x, packed_gate_up, act)
    M, K, ldx, fm = _x_args(x, frag_shape)
    if packed_gate_up.numel() != 2 * inter * K or packed_gate_up.dtype != x.dtype or act.dtype != x.dtype:
        raise _lib.HydraHipError("gate_up_silu_xreg: packed weight [2*inter, K] and act of x's dtype")
    if not act.is_contiguous() or act.numel() < fragment_major_elems(M, inter):
        raise _lib.HydraHipError("gate_up_silu_xreg: act needs ceil16(M) * inter contiguous elements")
    _lib.check(_lib.lib().hx_gate_up_silu_xreg(act.data_ptr(), x.data_ptr(), packed_gate_up.data_ptr(), M, inter, K,
                                               ldx, fm, _lib.dtype_code(x), _lib.current_stream()), "gate_up_silu_xreg")


# ---- add + RMSNorm fused in front of the product (hx_norm_*_xreg) ---------------------------------
XREG_SYNC_WORDS = _lib.HX_XREG_SYNC_WORDS


def norm_xreg_supported(M: int, N: int, K: int, dtype: torch.dtype, gate_up: bool = False) -> bool:
    return (dtype in (torch.float16, torch.bfloat16)
            and _lib.lib().hx_norm_xreg_supported(M, N, K, 1 if gate_up else 0) == 1)


def _norm_checks(residual: Tensor, slabs_in: Tensor, n_splits_in: int, norm_weight: Tensor, x_frag: Tensor, sync: Tensor):
    M, K = residual.shape
    if not residual.is_contiguous() or norm_weight.dtype != residual.dtype or norm_weight.numel() != K:
        raise _lib.HydraHipError("norm+xreg: residual [M, K] contiguous, norm_weight [K] of the same dtype")
    if slabs_in.dtype != torch.float32 or not slabs_in.is_contiguous() or slabs_in.numel() < n_splits_in * M * K:
        raise _lib.HydraHipError("norm+xreg: slabs_in must be contiguous float32 [n_splits_in, M, K]")
    if x_frag.dtype != residual.dtype or not x_frag.is_contiguous() or x_frag.numel() < fragment_major_elems(M, K):
        raise _lib.HydraHipError("norm+xreg: x_frag needs ceil16(M) * K contiguous elements of the activations' dtype")
    if sync.dtype != torch.int32 or not sync.is_contiguous() or sync.numel() < XREG_SYNC_WORDS:
        raise _lib.HydraHipError("norm+xreg: sync must be XREG_SYNC_WORDS contiguous int32 (zeroed)")
    return M, K


def norm_linear_decode_xreg(residual: Tensor, slabs_in: Tensor, n_splits_in: int, norm_weight: Tensor, epsilon: float,
                            x_frag: Tensor, packed: Tensor, N: int, partial: Tensor, sync: Tensor) -> int:
    """add_rms_norm_slabs(fragment-major) + linear_decode_partial_xreg as ONE launch: residual += (T) sum of
    slabs_in; x_frag = rms_norm(residual) * norm_weight; partial = slabs of x @ W^T.  `sync`: zeroed
    int32[XREG_SYNC_WORDS], one per launch in flight.  Returns the number of slabs written."""
    _lib.require_gpu(residual, slabs_in, norm_weight, x_frag, packed, partial, sync)
    M, K = _norm_checks(residual, slabs_in, n_splits_in, norm_weight, x_frag, sync)
    if packed.numel() != N * K or packed.dtype != residual.dtype or partial.dtype != torch.float32 or not partial.is_contiguous():
        raise _lib.HydraHipError("norm_linear_decode_xreg: packed = pack_weight_xreg(W [N, K]); partial contiguous float32")
    rc = _lib.lib().hx_norm_linear_decode_xreg(partial.data_ptr(), residual.data_ptr(), slabs_in.data_ptr(), int(n_splits_in),
                                               norm_weight.data_ptr(), float(epsilon), x_frag.data_ptr(), packed.data_ptr(),
                                               M, N, K, sync.data_ptr(), partial.numel() * 4, _lib.dtype_code(residual),
                                               _lib.current_stream())
    if rc < 0:
        _lib.check(rc, "norm_linear_decode_xreg")
    return rc


def norm_gate_up_silu_xreg(residual: Tensor, slabs_in: Tensor, n_splits_in: int, norm_weight: Tensor, epsilon: float,
                           x_frag: Tensor, packed_gate_up: Tensor, inter: int, act: Tensor, sync: Tensor) -> None:
    """add_rms_norm_slabs(fragment-major) + gate_up_silu_xreg as ONE launch (act fragment-major [M, inter])."""
    _lib.require_gpu(residual, slabs_in, norm_weight, x_frag, packed_gate_up, act, sync)
    M, K = _norm_checks(residual, slabs_in, n_splits_in, norm_weight, x_frag, sync)
    if packed_gate_up.numel() != 2 * inter * K or packed_gate_up.dtype != residual.dtype or act.dtype != residual.dtype \
            or not act.is_contiguous() or act.numel() < fragment_major_elems(M, inter):
        raise _lib.HydraHipError("norm_gate_up_silu_xreg: packed [2*inter, K] (interleaved), act ceil16(M) * inter elements")
    _lib.check(_lib.lib().hx_norm_gate_up_silu_xreg(act.data_ptr(), residual.data_ptr(), slabs_in.data_ptr(), int(n_splits_in),
                                                    norm_weight.data_ptr(), float(epsilon), x_frag.data_ptr(),
                                                    packed_gate_up.data_ptr(), M, inter, K, sync.data_ptr(),
                                                    _lib.dtype_code(residual), _lib.current_stream()),
               "norm_gate_up_silu_xreg")


# ---- batches of 33 .. 64 rows: the gate|up product over the interleaved packing, silu*mul as its own launch ----------
def gate_up_xreg_supported(M: int, inter: int, K: int, dtype: torch.dtype, with_norm: bool = False) -> bool:
    return (dtype in (torch.float16, torch.bfloat16)
            and _lib.lib().hx_gate_up_xreg_supported(M, inter, K, 1 if with_norm else 0) == 1)


def gate_up_xreg_workspace_floats(M: int, inter: int, K: int) -> int:
    return _lib.lib().hx_gate_up_xreg_workspace_bytes(M, inter, K) // 4


def gate_up_xreg(x: Tensor, packed_gate_up: Tensor, inter: int, partial: Tensor, frag_shape=None) -> int:
    """fp32 slabs [n_splits, M, 2*inter] ([gate | up] column order) of x[M <= 64, K] @ W_gate_up^T with
    packed_gate_up = pack_weight_xreg(W [2*inter, K], interleave_halves=True) — the packing hx_gate_up_silu_xreg reads,
    here WITHOUT the fused silu*mul (silu_and_mul_slabs follows).  Returns the number of slabs."""
    _lib.require_gpu(x, packed_gate_up, partial)
    M, K, ldx, fm = _x_args(x, frag_shape)
    if packed_gate_up.numel() != 2 * inter * K or packed_gate_up.dtype != x.dtype or partial.dtype != torch.float32 \
            or not partial.is_contiguous():
        raise _lib.HydraHipError("gate_up_xreg: packed weight [2*inter, K] of x's dtype, partial contiguous float32")
    rc = _lib.lib().hx_gate_up_xreg(partial.data_ptr(), x.data_ptr(), packed_gate_up.data_ptr(), M, inter, K, ldx, fm,
                                    partial.numel() * 4, _lib.dtype_code(x), _lib.current_stream())
    if rc < 0:
        _lib.check(rc, "gate_up_xreg")
    return rc


def norm_gate_up_xreg(residual: Tensor, slabs_in: Tensor, n_splits_in: int, norm_weight: Tensor, epsilon: float,
                      x_frag: Tensor, packed_gate_up: Tensor, inter: int, partial: Tensor, sync: Tensor) -> int:
    """add_rms_norm_slabs(fragment-major) + gate_up_xreg as ONE launch (M <= 64).  Returns the number of slabs."""
    _lib.require_gpu(residual, slabs_in, norm_weight, x_frag, packed_gate_up, partial, sync)
    M, K = _norm_checks(residual, slabs_in, n_splits_in, norm_weight, x_frag, sync)
    if packed_gate_up.numel() != 2 * inter * K or packed_gate_up.dtype != residual.dtype or partial.dtype != torch.float32 \
            or not partial.is_contiguous():
        raise _lib.HydraHipError("norm_gate_up_xreg: packed [2*inter, K] (interleaved), partial contiguous float32")
    rc = _lib.lib().hx_norm_gate_up_xreg(partial.data_ptr(), residual.data_ptr(), slabs_in.data_ptr(), int(n_splits_in),
                                         norm_weight.data_ptr(), float(epsilon), x_frag.data_ptr(), packed_gate_up.data_ptr(),
                                         M, inter, K, sync.data_ptr(), partial.numel() * 4, _lib.dtype_code(residual),
                                         _lib.current_stream())
    if rc < 0:
        _lib.check(rc, "norm_gate_up_xreg")
    return rc


def gate_up_silu_wide_supported(M: int, inter: int, K: int, dtype: torch.dtype) -> bool:
    return dtype in (torch.float16, torch.bfloat16) and _lib.lib().hx_gate_up_silu_wide_xreg_supported(M, inter, K) == 1


def norm_gate_up_silu_wide_xreg(residual: Tensor, slabs_in: Tensor, n_splits_in: int, norm_weight: Tensor, epsilon: float,
                                x_frag: Tensor, packed_gate_up: Tensor, inter: int, act: Tensor, sync: Tensor) -> None:
    """norm_gate_up_xreg + silu_and_mul_slabs(fragment_major=True) as ONE launch without slabs for 33 .. 64 rows: act
    (fragment-major [ceil(M / 16) * 16, inter]) = silu(gate) * up, bit-identical to the two-launch form."""
    _lib.require_gpu(residual, slabs_in, norm_weight, x_frag, packed_gate_up, sync, act)
    M, K = _norm_checks(residual, slabs_in, n_splits_in, norm_weight, x_frag, sync)
    if packed_gate_up.numel() != 2 * inter * K or packed_gate_up.dtype != residual.dtype:
        raise _lib.HydraHipError("norm_gate_up_silu_wide_xreg: packed [2*inter, K] (interleaved) of the model dtype")
    if act.dtype != residual.dtype or not act.is_contiguous() or act.numel() < (M + 15) // 16 * 16 * inter:
        raise _lib.HydraHipError("norm_gate_up_silu_wide_xreg: act holds ceil(M / 16) * 16 * inter elements of the model dtype")
    _lib.check(_lib.lib().hx_norm_gate_up_silu_wide_xreg(
        act.data_ptr(), residual.data_ptr(), slabs_in.data_ptr(), int(n_splits_in), norm_weight.data_ptr(), float(epsilon),
        x_frag.data_ptr(), packed_gate_up.data_ptr(), M, inter, K, sync.data_ptr(), _lib.dtype_code(residual),
        _lib.current_stream()), "norm_gate_up_silu_wide_xreg")


# ------------------------------------------------------------------------------------------------
# the single-entry form (include/hydra_hip.h hx_decode_weight / hx_linear_decode_ex): describe the weight once, the
# library picks the layout for the largest decode batch it has to serve, packs, and dispatches
# ------------------------------------------------------------------------------------------------
class DecodeWeight:
    """A [N, K] linear weight packed for decode batches of <= max_rows rows.  .layout is "xreg" or "lds_slice";
    .interleaved tells whether a gate|up weight was packed for the fused silu*mul epilogue."""

    def __init__(self, weight: Tensor, max_rows: int = 32, gate_up: bool = False, lds_slice: bool = False):
        _lib.require_gpu(weight)
        if weight.dim() != 2 or weight.stride(1) != 1 or weight.dtype not in (torch.float16, torch.bfloat16):
            raise _lib.HydraHipError("DecodeWeight: fp16 / bf16 [N, K] with contiguous rows")
        N, K = weight.shape
        self.desc = _lib.hx_decode_weight()
        _lib.check(_lib.lib().hx_decode_weight_plan(ctypes.byref(self.desc), N, K, _lib.dtype_code(weight), int(max_rows),
                                                    (_lib.HX_DW_GATE_UP if gate_up else 0)
                                                    | (_lib.HX_DW_FORCE_LDS_SLICE if lds_slice else 0)), "decode_weight_plan")
        self.packed = torch.empty(N * K, dtype=weight.dtype, device=weight.device)
        _lib.check(_lib.lib().hx_decode_weight_pack(ctypes.byref(self.desc), self.packed.data_ptr(), weight.data_ptr(),
                                                    weight.stride(0), _lib.current_stream()), "decode_weight_pack")
        self.N, self.K, self.max_rows = N, K, int(max_rows)

    @property
    def layout(self) -> str:
        return "xreg" if self.desc.layout == _lib.HX_DW_XREG else "lds_slice"

    @property
    def interleaved(self) -> bool:
        return bool(self.desc.flags & _lib.HX_DW_GATE_UP)

    def workspace_floats(self, M: int) -> int:
        return _lib.lib().hx_linear_decode_ex_workspace_bytes(ctypes.byref(self.desc), M) // 4


def linear_decode_ex(x: Tensor, w: DecodeWeight, partial: Tensor, frag_shape=None) -> int:
    """fp32 slabs [n_slabs, M, N] of x @ W^T through the one dispatching entry; returns n_slabs."""
    _lib.require_gpu(x, partial)
    M, K, ldx, fm = _x_args(x, frag_shape)
    if K != w.K or x.dtype != w.packed.dtype:
        raise _lib.HydraHipError("linear_decode_ex: x does not match the packed weight")
    if partial.dtype != torch.float32 or not partial.is_contiguous():
        raise _lib.HydraHipError("linear_decode_ex: partial must be contiguous float32")
    rc = _lib.lib().hx_linear_decode_ex(partial.data_ptr(), partial.numel() * 4, x.data_ptr(), ldx, fm, ctypes.byref(w.desc),
                                        M, _lib.current_stream())
    if rc < 0:
        _lib.check(rc, "linear_decode_ex")
    return rc
