"""Llama decoder (the language half of LLaVA-1.5) — the caller of the hot path.

Mirrors hydrainfer/model/llama.py:21-104 + hydrainfer/model/model_forward.py:39-105:
    h = h + o_proj(Attn(rope(q_proj, k_proj), v_proj)) ; h = h + down(silu(gate) * up)
with pre-RMSNorm, greedy argmax inside the model (llama.py:99-104), last-layer token
selection for prefill (model_forward.py:101-103).

MI355X-first differences (all rounding-neutral with respect to the reference's unfused ops):
  * q/k/v and gate/up weights are stored fused ([3h, h] and [2i, h]) so a layer is 4 library
    GEMMs instead of 7; `from_reference_state_dict` builds them from the reference's names.
  * residual-add + RMSNorm, RoPE (in place on the qkv buffer), set_kv_cache + attention and
    silu*mul each run as one HIP launch.
GEMMs are plain library GEMMs (torch.matmul -> hipBLASLt); everything else is libhydra_hip."""
import os
from dataclasses import dataclass
from typing import Dict, List, Optional

import torch
from torch import Tensor

from hydrainfer_amd._C.kernel.activation import silu_and_mul, silu_and_mul_slabs
from hydrainfer_amd._C.kernel.norm import (StepHead, add_rms_norm, add_rms_norm_slabs, argmax_rows, decode_step_head,
                                             embed_rms_norm, embed_rms_norm_supported, rms_norm)
from hydrainfer_amd._C.kernel.position_embedding import rope_set_kv_cache
from hydrainfer_amd.layer.causal_attention import AttentionParameters
from hydrainfer_amd._C.kernel.flash_attn import decode_attention_fused, mha_varlen_fwd
from hydrainfer_amd._C.kernel import gemm as hip_gemm
from hydrainfer_amd import _lib, launch_plan


@dataclass
class LlamaShape:
    hidden_size: int
    intermediate_size: int
    num_hidden_layers: int
    num_attention_heads: int
    num_key_value_heads: int
    head_dim: int
    vocab_size: int
    rms_norm_eps: float = 1e-5
    rope_theta: float = 10000.0
    max_position_embeddings: int = 4096


# SURVEY.md §8 model constants
LLAVA_1_5_7B = LlamaShape(4096, 11008, 32, 32, 32, 128, 32064)
LLAVA_1_5_13B = LlamaShape(5120, 13824, 40, 40, 40, 128, 32064)


@dataclass
class LanguageModelParameters:
    """hydrainfer/model/parameters.py:21-29 (fields used by the language model)."""
    attention_params: List[AttentionParameters]
    all_sequences_decode: bool
    selected_token_ids: Optional[Tensor] = None  # int64 index tensor on the device
    image_row_index: Optional[Tensor] = None     # int64 rows of the batch that are image tokens (host knows them:
                                                 # spares the nonzero() sync of a boolean-mask assignment)
    step_head: Optional[StepHead] = None         # decode loops: metadata advance / look-ahead feed folded into the step's
                                                 # first launch (hx_decode_step_head); the model runs it or raises


def build_cos_sin(shape: LlamaShape, dtype: torch.dtype, device) -> Tensor:
    inv = 1.0 / torch.pow(shape.rope_theta,
                          torch.arange(0, shape.head_dim, 2, dtype=torch.float) / shape.head_dim)
    t = torch.arange(shape.max_position_embeddings, dtype=torch.float)
    freqs = torch.einsum("i,j->ij", t, inv)
    cs = torch.cat([freqs.cos()[:, None, :], freqs.sin()[:, None, :]], dim=1)
    return cs.to(dtype).to(device)


class LlamaForCausalLM:
    """Weights live in fused device tensors; `state` maps fused names to tensors."""

    def __init__(self, shape: LlamaShape, dtype: torch.dtype, device, state: Dict[str, Tensor]):
        self.shape, self.dtype, self.device = shape, dtype, torch.device(device)
        self.state = state
        self.cos_sin = build_cos_sin(shape, dtype, self.device)
        self.q_size = shape.num_attention_heads * shape.head_dim
        self.kv_size = shape.num_key_value_heads * shape.head_dim
        # decode batches (<= 64 rows) stream the weights through the HIP kernel; larger
        # batches (prefill) use the library GEMM
        self.use_hip_gemm = True
        # decode steps: RoPE + cache append + attention as one launch.  Grouped-query models take
        # three launches instead (GEMM reduce, RoPE + append, attention): the fused kernel gives
        # every query head its own workgroup, the unfused entry picks attn_decode_gqa.hip, which
        # reads each KV head once (3x faster at group 4 and more than pays for the two launches)
        self.fuse_decode_attention = (shape.head_dim in (64, 128, 256)
                                      and shape.num_key_value_heads == shape.num_attention_heads)
        # decode GEMMs stream PACKED copies of the weights (MFMA-fragment order, contiguous 1 KiB
        # reads: csrc/gemm_skinny.hip gemm_packed_kernel); the row-major tensors stay for the
        # prefill GEMMs (library).  288 GB of HBM: the second copy of a 7B / 13B model is 13 / 26 GB.
        self.use_packed = True
        self.packed: Dict[str, Tensor] = {}
        # decode batches of <= 32 rows: the MLP runs on the activations-in-registers GEMM
        # (csrc/gemm_xreg.hip) — gate|up + silu*mul in ONE launch that needs no K split, down with 3
        # slabs instead of 11; the activations travel between these launches in MFMA-fragment order.
        # 7 launches per layer, -7.5 us per 7B layer (tools/bench_layer_xreg.py).
        self.use_xreg = True
        # qkv of layers >= 1 on the same kernel (its x comes fragment-major from the previous layer's
        # add+norm; the attention prologue then reads 1 slab instead of 4): -45 us per 7B step
        self.xreg_qkv = os.environ.get("HX_XREG_QKV", "1") == "1"
        # the two add+norm launches of a layer run INSIDE the gate|up and qkv launches (the first 32
        # workgroups produce x while all prefetch weights; in-kernel hand-over): 5 launches per layer
        self.fuse_norm = os.environ.get("HX_FUSE_NORM", "1") == "1"
        # batches of 33 .. 64 rows on the same layout (the wide kernel: 6 launches per layer, no LDS-slice copies)
        self.use_wide = os.environ.get("HX_WIDE", "1") == "1"
        # ... with silu * mul inside the norm + gate|up launch (round 5: both K halves in one workgroup, no slabs)
        self.use_wide_silu = os.environ.get("HX_WIDE_SILU", "1") == "1"
        self.sample_out: Optional[Tensor] = None   # int64 [rows]: forward() writes the sampled ids here (decode loops)
        self.xreg_sync: Optional[Tensor] = None   # [L, 2, XREG_SYNC_WORDS] of the last step (word 1 = wait gave up)
        self.packed_x: Dict[str, Tensor] = {}
        self.dw: Dict[str, "hip_gemm.DecodeWeight"] = {}       # descriptors of the packed copies (xreg layout)
        self.dw_lds: Dict[str, "hip_gemm.DecodeWeight"] = {}   # ... (LDS-slice layout)
        # decode-side weight layouts are built ONCE, by prepare_decode (engine build / runner construction), for
        # the largest decode batch the owner will ever run — never in the middle of serving
        self.decode_rows_prepared = 0
        self.decode_only = False                   # row-major decoder weights dropped (D-role node)

    # ------------------------------------------------------------------ decode-side weight layouts
    def prepare_decode(self, max_rows: int = 64, keep_row_major: bool = True) -> None:
        """Builds every packed layout a decode batch of <= max_rows rows can need, now (model load, before the
        KV pool is sized): the activations-in-registers layout for gate|up, down and qkv of layers >= 1 (<= 32
        rows), the LDS-slice layout for o and layer 0's qkv — and for all four projections when max_rows > 32
        (batches of 33..64 rows run on that kernel).  keep_row_major=False is for nodes that never prefill
        (parallel.epd_roles: "D"): the row-major decoder weights — only the library prefill GEMMs read them —
        are released, ONE copy of the weights stays resident.  Reference: one nn.Linear weight per projection
        (hydrainfer/model/llama.py:24-27,48-50)."""
        max_rows = min(int(max_rows), 64)
        if max_rows > self.decode_rows_prepared and not self.decode_only:
            self.pack_decode_weights(all_lds_slice=max_rows > 32 and not self._wide_ok(max_rows))
            self.decode_rows_prepared = max_rows
        if not keep_row_major and not self.decode_only and self._all_packed():
            for l in range(self.shape.num_hidden_layers):
                for n in ("wqkv", "wo", "wgu", "wdown"):
                    w = self.state[f"l{l}.{n}"]
                    # shape / stride / dtype carrier without storage: anything that tries to read it fails loudly
                    self.state[f"l{l}.{n}"] = torch.empty_strided(w.shape, w.stride(), dtype=w.dtype, device="meta")
            self.decode_only = True

    def _all_packed(self) -> bool:
        return all((f"l{l}.{n}" in self.packed or f"l{l}.{n}" in self.packed_x)
                   for l in range(self.shape.num_hidden_layers) for n in ("wqkv", "wo", "wgu", "wdown"))

    def weight_bytes_resident(self) -> int:
        """Bytes of HBM the weights occupy in all their layouts (row-major + packed copies)."""
        tensors = [v for v in self.state.values() if v.device.type != "meta"]
        tensors += list(self.packed.values()) + list(self.packed_x.values())
        return sum(t.numel() * t.element_size() for t in tensors)

    def release(self) -> None:
        """Drops every weight tensor (bench.py frees the 7B model before its 13B leg)."""
        self.state, self.packed, self.packed_x, self.dw, self.dw_lds = {}, {}, {}, {}, {}
        self.xreg_sync = None

    def handover_failed(self) -> bool:
        """True if an in-kernel hand-over of the last decode step gave up waiting (word 1 of a norm-fused
        launch's sync area): that step's activations are garbage.
        One small D2H sync — callers on the hot path fold the flag into their token copy instead
        (engine/graph_decode.py)."""
        bad = False
        if self.xreg_sync is not None:
            bad = bool(int(self.xreg_sync[:, :, 1].abs().sum()) != 0)
        return bad

    def handover_error_word(self) -> Optional[Tensor]:
        """int32 scalar tensor on the device, nonzero iff a hand-over of the step just enqueued gave up; None when
        the step had no in-kernel hand-over.  Enqueued on the current stream (capturable)."""
        if self.xreg_sync is not None:
            return self.xreg_sync[:, :, 1].abs().sum().to(torch.int32)
        return None

    def pack_decode_weights(self, all_lds_slice: bool = False) -> None:
        """Builds the packed copies of the decoder-layer weights (prepare_decode calls this; never during a capture)
        through the library's one decode-weight entry (hx_decode_weight_plan / _pack, gemm.DecodeWeight): planned for
        <= 32 rows it picks the activations-in-registers layout where the shape allows it (gate|up with its halves
        interleaved for the fused silu*mul epilogue), planned for 64 rows the LDS-slice layout.  o and layer 0's qkv
        always take the LDS-slice layout (o: at 33 MB the x broadcast of the other kernel would dominate; layer 0's
        qkv: its x comes row-major from the embedding launch); the LDS-slice copies of the other projections are
        built only when batches of 33..64 rows are announced: all_lds_slice."""
        if not (self.use_packed and self.use_hip_gemm and self.dtype in (torch.float16, torch.bfloat16)):
            return
        if self.use_xreg and self._xreg_mlp_ok(32):
            for l in range(self.shape.num_hidden_layers):
                if f"l{l}.wgu" not in self.packed_x:
                    for n, gu in (("wgu", True), ("wdown", False)):
                        dw = hip_gemm.DecodeWeight(self.state[f"l{l}.{n}"], max_rows=32, gate_up=gu)
                        assert dw.layout == "xreg"
                        self.dw[f"l{l}.{n}"], self.packed_x[f"l{l}.{n}"] = dw, dw.packed
                    wq = self.state[f"l{l}.wqkv"]
                    if self.xreg_qkv and l > 0 and wq.stride(1) == 1 and hip_gemm.xreg_supported(32, wq.shape[0], wq.shape[1], self.dtype):
                        dw = hip_gemm.DecodeWeight(wq, max_rows=32)
                        self.dw[f"l{l}.wqkv"], self.packed_x[f"l{l}.wqkv"] = dw, dw.packed
        for l in range(self.shape.num_hidden_layers):
            for n in ("wqkv", "wo", "wgu", "wdown"):
                key = f"l{l}.{n}"
                if all_lds_slice or key not in self.packed_x:
                    self._pack_lds_slice(key)

    def _pack_lds_slice(self, key: str) -> Optional[Tensor]:
        w = self.state[key]
        if key not in self.packed and w.shape[0] % 16 == 0 and w.shape[1] % 256 == 0 and w.stride(1) == 1:
            dw = hip_gemm.DecodeWeight(w, max_rows=64, lds_slice=True)     # row-major x in (o, layer 0's qkv) / 13B-like shapes
            assert dw.layout == "lds_slice"
            self.dw_lds[key], self.packed[key] = dw, dw.packed
        return self.packed.get(key)

    def _new_sync(self, device) -> Tensor:
        return torch.empty((self.shape.num_hidden_layers, 2, hip_gemm.XREG_SYNC_WORDS), dtype=torch.int32, device=device)

    def _xreg_mlp_ok(self, n: int) -> bool:
        hid, inter = self.shape.hidden_size, self.shape.intermediate_size
        if n > 32:
            return self._wide_ok(n)
        return (self.dtype in (torch.float16, torch.bfloat16) and inter % 32 == 0 and hid % 32 == 0
                and hip_gemm.xreg_supported(n, 2 * inter, hid, self.dtype) and hip_gemm.xreg_supported(n, hid, inter, self.dtype)
                and all(self.state[f"l0.{k}"].stride(1) == 1 for k in ("wgu", "wdown")))

    def _wide_ok(self, n: int) -> bool:
        """Batches of 33 .. 64 rows on the activations-in-registers layout (the wide kernel reads the <= 32-row packing:
        no second copy): gate|up, down and qkv of this shape must all be supported (LLaVA-1.5-7B, and since round 5
        LLaVA-1.5-13B: its k-steps per wave, 40 and 27, halve to 20 and 14 + 13; other shapes stay on LDS-slice copies)."""
        hid, inter = self.shape.hidden_size, self.shape.intermediate_size
        qkv_n = self.q_size + 2 * self.kv_size
        return (32 < n <= 64 and self.use_wide and self.use_xreg and self.xreg_qkv and self.dtype in (torch.float16, torch.bfloat16)
                and inter % 32 == 0 and hid % 32 == 0 and self._xreg_mlp_ok(32)
                and hip_gemm.gate_up_xreg_supported(n, inter, hid, self.dtype)
                and hip_gemm.xreg_supported(n, hid, inter, self.dtype) and hip_gemm.xreg_supported(n, qkv_n, hid, self.dtype)
                and hip_gemm.xreg_supported(32, qkv_n, hid, self.dtype))

    def _partial(self, x: Tensor, key: str, ws: Tensor) -> int:
        """split-K slabs of x @ state[key]^T into ws; packed weights when available."""
        dw = self.dw_lds.get(key)
        if dw is not None:
            return hip_gemm.linear_decode_ex(x, dw, ws)
        if self.decode_only:
            raise RuntimeError(f"decode-only model: no packed layout of {key} for {x.shape[0]} rows (prepare_decode("
                               f"max_rows={self.decode_rows_prepared}) was called) and the row-major weight is released")
        return hip_gemm.linear_decode_partial(x, self.state[key], ws)

    def linear(self, x: Tensor, w: Tensor) -> Tensor:
        if w.device.type == "meta":
            raise RuntimeError("decode-only model (prepare_decode(keep_row_major=False)): prefill / row-major GEMMs "
                               "are not available on this node")
        if self.use_hip_gemm and x.shape[0] <= 64 and hip_gemm.supported(x, w):
            return hip_gemm.linear_decode(x, w)
        return torch.matmul(x, w.t())

    # ------------------------------------------------------------------ construction
    @classmethod
    def random_init(cls, shape: LlamaShape, dtype: torch.dtype, device, seed: int = 0,
                    std: float = 0.02) -> "LlamaForCausalLM":
        """N(0, std) linears/embeddings, norm weights 1 (SURVEY.md §8d synthetic weights)."""
        g = torch.Generator(device=device).manual_seed(seed)
        h, i, L = shape.hidden_size, shape.intermediate_size, shape.num_hidden_layers
        q, kv = shape.num_attention_heads * shape.head_dim, shape.num_key_value_heads * shape.head_dim

        def w(*size):
            return (torch.randn(size, generator=g, device=device, dtype=torch.float32) * std).to(dtype)

        st: Dict[str, Tensor] = {"embed": w(shape.vocab_size, h), "lm_head": w(shape.vocab_size, h),
                                 "norm": torch.ones(h, dtype=dtype, device=device)}
        for l in range(L):
            st[f"l{l}.wqkv"] = w(q + 2 * kv, h)
            st[f"l{l}.wo"] = w(h, q)
            st[f"l{l}.wgu"] = w(2 * i, h)
            st[f"l{l}.wdown"] = w(h, i)
            st[f"l{l}.norm1"] = torch.ones(h, dtype=dtype, device=device)
            st[f"l{l}.norm2"] = torch.ones(h, dtype=dtype, device=device)
        return cls(shape, dtype, device, st)

    @classmethod
    def from_reference_state_dict(cls, shape: LlamaShape, sd: Dict[str, Tensor], dtype, device,
                                  prefix: str = "") -> "LlamaForCausalLM":
        """Names of hydrainfer/model/llama.py (== HF Llama): model.layers.N.self_attn.q_proj.weight ..."""
        def t(name):
            return sd[prefix + name].to(dtype).to(device)
        st = {"embed": t("model.embed_tokens.weight"), "lm_head": t("lm_head.weight"),
              "norm": t("model.norm.weight")}
        for l in range(shape.num_hidden_layers):
            p = f"model.layers.{l}."
            st[f"l{l}.wqkv"] = torch.cat([t(p + "self_attn.q_proj.weight"), t(p + "self_attn.k_proj.weight"),
                                          t(p + "self_attn.v_proj.weight")], dim=0).contiguous()
            st[f"l{l}.wo"] = t(p + "self_attn.o_proj.weight")
            st[f"l{l}.wgu"] = torch.cat([t(p + "mlp.gate_proj.weight"), t(p + "mlp.up_proj.weight")],
                                        dim=0).contiguous()
            st[f"l{l}.wdown"] = t(p + "mlp.down_proj.weight")
            st[f"l{l}.norm1"] = t(p + "input_layernorm.weight")
            st[f"l{l}.norm2"] = t(p + "post_attention_layernorm.weight")
        return cls(shape, dtype, device, st)

    def to_reference_state_dict(self) -> Dict[str, Tensor]:
        """Unfused CPU copy under the reference's parameter names (consumed by the oracle)."""
        s, q, kv, i = self.state, self.q_size, self.kv_size, self.shape.intermediate_size
        sd = {"model.embed_tokens.weight": s["embed"].cpu(), "lm_head.weight": s["lm_head"].cpu(),
              "model.norm.weight": s["norm"].cpu()}
        for l in range(self.shape.num_hidden_layers):
            p = f"model.layers.{l}."
            wqkv, wgu = s[f"l{l}.wqkv"].cpu(), s[f"l{l}.wgu"].cpu()
            sd[p + "self_attn.q_proj.weight"] = wqkv[:q]
            sd[p + "self_attn.k_proj.weight"] = wqkv[q:q + kv]
            sd[p + "self_attn.v_proj.weight"] = wqkv[q + kv:]
            sd[p + "self_attn.o_proj.weight"] = s[f"l{l}.wo"].cpu()
            sd[p + "mlp.gate_proj.weight"] = wgu[:i]
            sd[p + "mlp.up_proj.weight"] = wgu[i:]
            sd[p + "mlp.down_proj.weight"] = s[f"l{l}.wdown"].cpu()
            sd[p + "input_layernorm.weight"] = s[f"l{l}.norm1"].cpu()
            sd[p + "post_attention_layernorm.weight"] = s[f"l{l}.norm2"].cpu()
        return sd

    def weight_bytes(self) -> int:
        """Bytes a decode step must stream: all linears + lm_head (embedding gather ignored),
        the `W` term of SURVEY.md §8d."""
        n = sum(v.numel() for k, v in self.state.items() if k != "embed" and v.dim() == 2)
        return n * self.state["lm_head"].element_size()

    # ------------------------------------------------------------------ forward
    def embed(self, input_ids: Tensor) -> Tensor:
        return torch.nn.functional.embedding(input_ids, self.state["embed"])

    def _decode_plan(self, n: int, dtype) -> dict:
        """Which launches a decode step of n rows is made of (decided before the step's first launch, because that
        launch — hx_decode_step_head — also zeroes the hand-over areas of the norm-fused launches)."""
        sh = self.shape
        L, hid, inter = sh.num_hidden_layers, sh.hidden_size, sh.intermediate_size
        qkv_n = self.q_size + 2 * self.kv_size
        xreg = self.use_xreg and self._xreg_mlp_ok(n) and f"l{L - 1}.wdown" in self.packed_x
        wide = bool(xreg and n > 32)       # 33 .. 64 rows: two K splits per product; 6 launches per layer, 5 where wide_silu
        fused = xreg and not wide and hip_gemm.gate_up_silu_supported(n, inter, hid, dtype)
        if wide:
            nf_gu = bool(self.fuse_norm and hip_gemm.gate_up_xreg_supported(n, inter, hid, dtype, with_norm=True))
        else:
            nf_gu = bool(xreg and self.fuse_norm and fused and hip_gemm.norm_xreg_supported(n, 2 * inter, hid, dtype, gate_up=True))
        nf_qkv = bool(xreg and self.fuse_norm and f"l{L - 1}.wqkv" in self.packed_x
                      and hip_gemm.norm_xreg_supported(n, qkv_n, hid, dtype))
        wide_silu = bool(wide and nf_gu and self.use_wide_silu and hip_gemm.gate_up_silu_wide_supported(n, inter, hid, dtype))
        return {"xreg": xreg, "fused": fused, "nf_gu": nf_gu, "nf_qkv": nf_qkv, "wide": wide, "wide_silu": wide_silu}

    def _decode_hidden_hip_gemm(self, h: Tensor, position_ids: Tensor,
                                model_params: LanguageModelParameters, x0: Optional[Tensor] = None,
                                sync: Optional[Tensor] = None) -> Tensor:
        """All-decode step with the weight-streaming HIP GEMMs and fused slab consumers: 8
        launches per layer — qkv GEMM, [slab reduce + RoPE + cache append + attention], o GEMM,
        [slab reduce + residual add + RMSNorm], gate|up GEMM, [slab reduce + silu*mul], down GEMM,
        [slab reduce + residual add + RMSNorm]; 7 with use_xreg at <= 32 rows ([gate|up GEMM +
        silu*mul] is one launch).  Same rounding points as the unfused path."""
        sh, st = self.shape, self.state
        n = h.shape[0]
        H, HK, D = sh.num_attention_heads, sh.num_key_value_heads, sh.head_dim
        q_size, kv_size, inter, hid = self.q_size, self.kv_size, sh.intermediate_size, sh.hidden_size
        eps, L = sh.rms_norm_eps, sh.num_hidden_layers
        ws_n = max(hip_gemm.workspace_floats(n, q_size + 2 * kv_size, hid), hip_gemm.workspace_floats(n, hid, q_size),
                   hip_gemm.workspace_floats(n, 2 * inter, hid), hip_gemm.workspace_floats(n, hid, inter),
                   hip_gemm.xreg_workspace_floats(n, hid, inter))
        ws = torch.empty(ws_n, dtype=torch.float32, device=h.device)
        x = torch.empty_like(h)
        if n > self.decode_rows_prepared and not self.decode_only:
            if torch.cuda.is_current_stream_capturing():
                raise RuntimeError(f"decode batch of {n} rows captured before prepare_decode(max_rows >= {n})")
            # a caller that never announced its batch size (unit tests, ad-hoc scripts): everything a <= 64-row
            # batch can need, once; the engine / runners call prepare_decode at build time
            self.prepare_decode(max_rows=64)
        dp = self._decode_plan(n, h.dtype)
        xreg, fused, nf_gu, nf_qkv, wide = dp["xreg"], dp["fused"], dp["nf_gu"], dp["nf_qkv"], dp["wide"]
        qkv_n = q_size + 2 * kv_size
        ws_q = ws          # where the current layer's qkv slabs live
        if xreg:
            xf = torch.empty(hip_gemm.fragment_major_elems(n, hid), dtype=h.dtype, device=h.device)
            actf = torch.empty(hip_gemm.fragment_major_elems(n, inter), dtype=h.dtype, device=h.device) if fused else None
            # add+norm folded into the launch that consumes its output (hx_norm_*_xreg): 5 launches per layer
            if nf_gu or nf_qkv:
                # one zeroed hand-over area per fused launch of this step: zeroed by the step's first launch
                # (hx_decode_step_head) when the caller came through it, by a launch of its own otherwise
                if sync is None:
                    sync = self._new_sync(h.device)
                    _lib.memset_zero(sync)
                self.xreg_sync = sync
            if nf_qkv:   # the fused launch reads the down slabs (ws) while it writes the qkv slab
                ws_q = torch.empty(max(hip_gemm.xreg_workspace_floats(n, qkv_n, hid), hip_gemm.workspace_floats(n, qkv_n, hid)),
                                   dtype=torch.float32, device=h.device)
            wide_silu = bool(wide and nf_gu and dp["wide_silu"])
            if wide and not wide_silu:     # gate|up slabs of the wide product (the norm-fused form reads the o slabs in ws meanwhile)
                ws_gu = torch.empty(hip_gemm.gate_up_xreg_workspace_floats(n, inter, hid), dtype=torch.float32, device=h.device)
            actf_w = torch.empty(hip_gemm.fragment_major_elems(n, inter), dtype=h.dtype, device=h.device) if wide_silu else None
        if x0 is not None:
            x = x0          # the first layer's norm came with the embedding gather
        else:
            rms_norm(x, h, st["l0.norm1"], eps)
        s_qkv = None
        for l in range(L):
            ap = model_params.attention_params[l]
            kc, vc = ap.kv_cache.get_kv_cache()
            if s_qkv is None:
                if xreg and f"l{l}.wqkv" in self.packed_x:
                    if wide:
                        s_qkv = hip_gemm.linear_decode_partial_xreg(xf, self.packed_x[f"l{l}.wqkv"], qkv_n, ws_q, frag_shape=(n, hid))
                    else:
                        s_qkv = hip_gemm.linear_decode_ex(xf, self.dw[f"l{l}.wqkv"], ws_q, frag_shape=(n, hid))
                else:
                    s_qkv = self._partial(x, f"l{l}.wqkv", ws_q)
            o = torch.empty((n, H, D), dtype=h.dtype, device=h.device)
            # q / k_new / v_new arguments are shape carriers here: the kernel reads the slabs
            decode_attention_fused(o, o, o[:, :HK], o[:, :HK], kc, vc, position_ids, self.cos_sin,
                                   ap.new_cache_slots, ap.q_cu_seq_lens, ap.kv_cu_seq_lens, ap.block_tables,
                                   ap.cu_blocks_lens, ap.kv_max_seq_len, D ** -0.5, 0, ws_q, s_qkv, ap.decode_rank)
            s_qkv = None
            s_o = self._partial(o.view(n, q_size), f"l{l}.wo", ws)
            if xreg:
                # fragment-major activations from here to the down projection
                pgu, pdn = self.packed_x[f"l{l}.wgu"], self.packed_x[f"l{l}.wdown"]
                if wide:
                    # 33 .. 64 rows: (norm +) gate|up to two slabs, then silu*mul (fragment-major for the down product)
                    if wide_silu:
                        # silu * mul inside the same launch (both K halves in one workgroup): 5 launches per layer, no slabs
                        hip_gemm.norm_gate_up_silu_wide_xreg(h, ws, s_o, st[f"l{l}.norm2"], eps, xf, pgu, inter, actf_w, sync[l, 0])
                        a_f = actf_w
                    else:
                        if nf_gu:
                            s_gu = hip_gemm.norm_gate_up_xreg(h, ws, s_o, st[f"l{l}.norm2"], eps, xf, pgu, inter, ws_gu, sync[l, 0])
                        else:
                            add_rms_norm_slabs(xf, h, ws, s_o, st[f"l{l}.norm2"], eps, fragment_major=True)
                            s_gu = hip_gemm.gate_up_xreg(xf, pgu, inter, ws_gu, frag_shape=(n, hid))
                        a_f = silu_and_mul_slabs(ws_gu, s_gu, n, inter, h.dtype, fragment_major=True)
                elif nf_gu:
                    hip_gemm.norm_gate_up_silu_xreg(h, ws, s_o, st[f"l{l}.norm2"], eps, xf, pgu, inter, actf, sync[l, 0])
                    a_f = actf
                else:
                    add_rms_norm_slabs(xf, h, ws, s_o, st[f"l{l}.norm2"], eps, fragment_major=True)
                    if fused:
                        hip_gemm.gate_up_silu_xreg(xf, pgu, inter, actf, frag_shape=(n, hid))
                        a_f = actf
                    else:
                        s_gu = hip_gemm.linear_decode_partial_xreg(xf, pgu, 2 * inter, ws, frag_shape=(n, hid))
                        a_f = silu_and_mul_slabs(ws, s_gu, n, inter, h.dtype, fragment_major=True)
                if wide:
                    s_dn = hip_gemm.linear_decode_partial_xreg(a_f, pdn, hid, ws, frag_shape=(n, inter))
                else:
                    s_dn = hip_gemm.linear_decode_ex(a_f, self.dw[f"l{l}.wdown"], ws, frag_shape=(n, inter))
            else:
                add_rms_norm_slabs(x, h, ws, s_o, st[f"l{l}.norm2"], eps)
                s_gu = self._partial(x, f"l{l}.wgu", ws)
                act = silu_and_mul_slabs(ws, s_gu, n, inter, h.dtype)
                s_dn = self._partial(act, f"l{l}.wdown", ws)
            nxt = st[f"l{l + 1}.norm1"] if l + 1 < L else st["norm"]
            if xreg and f"l{l + 1}.wqkv" in self.packed_x:
                if nf_qkv:   # h += down; x = norm1(h); next layer's qkv slab — one launch
                    s_qkv = hip_gemm.norm_linear_decode_xreg(h, ws, s_dn, nxt, eps, xf, self.packed_x[f"l{l + 1}.wqkv"],
                                                             qkv_n, ws_q, sync[l, 1])
                else:
                    add_rms_norm_slabs(xf, h, ws, s_dn, nxt, eps, fragment_major=True)
            else:
                add_rms_norm_slabs(x, h, ws, s_dn, nxt, eps)
        return x

    def _decode_fast_path(self, n: int, dtype, model_params: LanguageModelParameters) -> bool:
        sh = self.shape
        return bool(self.use_hip_gemm and model_params.all_sequences_decode and self.fuse_decode_attention
                    and n <= 64 and dtype in (torch.float16, torch.bfloat16)
                    and sh.hidden_size % 256 == 0 and sh.intermediate_size % 256 == 0)

    def step_head_supported(self, n_rows: int) -> bool:
        """True if forward() of an all-decode batch of n_rows token ids runs LanguageModelParameters.step_head."""
        t = self.state["embed"]
        return bool(n_rows <= 64 and t.is_cuda and t.dtype in (torch.float16, torch.bfloat16) and t.is_contiguous()
                    and t.shape[1] % 8 == 0 and t.shape[1] <= 8192)

    def forward_hidden(self, input_ids_or_embeds: Tensor, position_ids: Tensor,
                       model_params: LanguageModelParameters) -> Tensor:
        sh, st = self.shape, self.state
        x0 = sync = None
        head = model_params.step_head
        if input_ids_or_embeds.dtype in (torch.int32, torch.int64):
            ids = input_ids_or_embeds
            if (model_params.all_sequences_decode and ids.dim() == 1 and ids.shape[0] <= 64
                    and embed_rms_norm_supported(ids, st["embed"])):
                # decode step: embedding gather + the first layer's norm + the zeroing of the hand-over areas of the
                # step's norm-fused launches + the caller's metadata advance / look-ahead feed: ONE launch
                if self._decode_fast_path(ids.shape[0], st["embed"].dtype, model_params):
                    dp = self._decode_plan(ids.shape[0], st["embed"].dtype)
                    if dp["nf_gu"] or dp["nf_qkv"]:
                        sync = self._new_sync(ids.device)
                h, x0 = decode_step_head(ids, st["embed"], st["l0.norm1"], sh.rms_norm_eps, zero=sync, head=head)
                head = None
            else:
                h = self.embed(ids)
        else:
            h = input_ids_or_embeds
        if head is not None:
            raise RuntimeError("LanguageModelParameters.step_head needs the decode fast path (<= 64 rows of token ids, "
                               "fp16 / bf16 embedding table): the caller must run its advance / feed itself")
        if not h.is_contiguous():
            h = h.contiguous()
        n = h.shape[0]
        if self._decode_fast_path(n, h.dtype, model_params):
            return self._decode_hidden_hip_gemm(h, position_ids, model_params, x0, sync)
        H, HK, D = sh.num_attention_heads, sh.num_key_value_heads, sh.head_dim
        q_size, kv_size, inter = self.q_size, self.kv_size, sh.intermediate_size
        eps = sh.rms_norm_eps
        L = sh.num_hidden_layers

        if x0 is not None:
            x = x0
        else:
            x = torch.empty_like(h)
            rms_norm(x, h, st["l0.norm1"], eps)
        for l in range(L):
            ap = model_params.attention_params[l]
            qkv = self.linear(x, st[f"l{l}.wqkv"])
            q = qkv[:, :q_size].view(n, H, D)
            k = qkv[:, q_size:q_size + kv_size].view(n, HK, D)
            v = qkv[:, q_size + kv_size:].view(n, HK, D)
            kc, vc = ap.kv_cache.get_kv_cache()
            o = torch.empty((n, H, D), dtype=h.dtype, device=h.device)
            if model_params.all_sequences_decode and self.fuse_decode_attention:
                # RoPE + cache append + paged attention, one launch
                decode_attention_fused(o, q, k, v, kc, vc, position_ids, self.cos_sin,
                                       ap.new_cache_slots, ap.q_cu_seq_lens, ap.kv_cu_seq_lens,
                                       ap.block_tables, ap.cu_blocks_lens, ap.kv_max_seq_len, D ** -0.5,
                                       rank_desc=ap.decode_rank)
            else:
                # RoPE in place + append k/v to the paged cache, one launch; then attention
                rope_set_kv_cache(q, k, v, position_ids, self.cos_sin, D, ap.new_cache_slots, kc, vc)
                mha_varlen_fwd(o, q, kc, vc, ap.q_cu_seq_lens, ap.kv_cu_seq_lens, ap.block_tables,
                               ap.cu_blocks_lens, None, ap.q_max_seq_len, ap.kv_max_seq_len,
                               D ** -0.5, 0, -1, 0, 0)
            a = self.linear(o.view(n, q_size), st[f"l{l}.wo"])
            # h += a ; x = norm2(h)
            add_rms_norm(x, h, a, st[f"l{l}.norm2"], eps)
            if (not model_params.all_sequences_decode) and l == L - 1 \
                    and model_params.selected_token_ids is not None:
                # last layer of a prefill: only sampled tokens go through the MLP
                h = h[model_params.selected_token_ids].contiguous()
                x = x[model_params.selected_token_ids].contiguous()
            gu = self.linear(x, st[f"l{l}.wgu"])
            act = silu_and_mul(gu[:, :inter], gu[:, inter:])
            m = self.linear(act, st[f"l{l}.wdown"])
            nxt = st[f"l{l + 1}.norm1"] if l + 1 < L else st["norm"]
            if x.shape != h.shape:
                x = torch.empty_like(h)
            add_rms_norm(x, h, m, nxt, eps)   # h += m ; x = next norm (final norm after last layer)
        return x

    def forward_logits(self, input_ids_or_embeds, position_ids, model_params) -> Tensor:
        x = self.forward_hidden(input_ids_or_embeds, position_ids, model_params)
        w = self.state["lm_head"]
        if launch_plan.current() is None:
            return torch.matmul(x, w.t())
        # a launch plan is being recorded: the library GEMM is a host-side step of the plan between two native
        # launch segments, writing into a buffer of the plan's pool
        logits = torch.empty((x.shape[0], w.shape[0]), dtype=x.dtype, device=x.device)
        launch_plan.host_op(lambda: torch.matmul(x, w.t(), out=logits))
        return logits

    def forward(self, input_ids_or_embeds, position_ids, model_params) -> Tensor:
        """Returns sampled token ids (greedy), like the reference model."""
        logits = self.forward_logits(input_ids_or_embeds, position_ids, model_params)
        if logits.is_cuda and logits.dim() == 2 and logits.stride(1) == 1 and logits.dtype in (torch.float16, torch.bfloat16):
            # one 4 us launch (the library reduction: 15 us); straight into the caller's next-input buffer if it set one
            out = self.sample_out if (self.sample_out is not None and self.sample_out.shape == (logits.shape[0],)) else None
            return argmax_rows(logits, out)
        return torch.argmax(logits, dim=-1)

    __call__ = forward
