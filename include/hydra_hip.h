/*
 * hydra_hip.h — C ABI of libhydra_hip.so: the MI355X (gfx950) implementation of
 * HydraInfer's attention + paged-KV operator surface (`hydrainfer._C.*`).
 *
 * Every entry point is `extern "C"`, takes raw device pointers, element
 * counts/strides (in ELEMENTS unless the name says bytes) and a hipStream_t
 * passed as `void*`, launches asynchronously on that stream, never
 * synchronises, never allocates, and returns 0 or a negative hx_status.
 * No torch types cross this boundary.  `hx_strerror` maps a status to text.
 *
 * Each function cites the reference interface it replaces
 * (paths relative to the dongxianzhe/hydrainfer tree).
 */
#ifndef HYDRA_HIP_H
#define HYDRA_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* 2: hx_attn_args.flags (was `reserved`) gates the local-window fields — a zero-filled tail of the struct means
 * "no softcap, no window"; launch plans; hx_decode_advance takes a stride; experiments moved to
 * hydra_hip_experimental.h */
#define HX_ABI_VERSION 3

typedef enum hx_dtype {
  HX_F32 = 0,
  HX_F16 = 1,
  HX_BF16 = 2,
} hx_dtype;

typedef enum hx_status {
  HX_OK = 0,
  HX_ERR_DTYPE = -1,       /* dtype not supported by this op                */
  HX_ERR_SHAPE = -2,       /* size / divisibility constraint violated       */
  HX_ERR_STRIDE = -3,      /* layout (contiguity / alignment) unsupported   */
  HX_ERR_NULL = -4,        /* required pointer is NULL                      */
  HX_ERR_UNSUPPORTED = -5, /* feature of the reference op not implemented   */
  HX_ERR_WORKSPACE = -6,   /* workspace too small                           */
  HX_ERR_HIP = -7,         /* a HIP runtime call failed (see hx_last_hip_error) */
  HX_ERR_HANDLE = -8,      /* bad IPC handle                                */
} hx_status;

typedef void* hx_stream; /* hipStream_t */

int hx_abi_version(void);
const char* hx_strerror(int status);
/* hipError_t of the most recent HX_ERR_HIP on this thread (0 if none). */
int hx_last_hip_error(void);
/* Tuning knobs for A/B measurements in one process (not part of the reference surface):
 * "decode_waves" = 4|8 waves per decode-attention workgroup, "decode_nt" = 0|1 non-temporal
 * K/V loads, "decode_small_lo" / "decode_small_hi" = range of (sequence, head) pair counts served
 * by the 8-wave no-split form, "fwd_row_blocks" = 0(auto)|1|2 query row blocks per wave,
 * "fwd_key_units" = 0(auto)|1|2 32-key units per tile and "fwd_xcd_remap" = 0|1 XCD-aware
 * workgroup numbering in the prefill kernel, "fwd_persistent" = 0|1(auto)|2 one workgroup per
 * (sequence, query tile, head) item | persistent workgroups where the launcher expects a gain |
 * persistent workgroups wherever their tables fit, "fwd_priority" = -1(auto)|0|1 and
 * "fwd_seq_group" = 0(auto)|1|2|4, "fwd_units" = -1(auto)|0|1 for the persistent form (equal
 * priorities or priority to the workgroup of a CU with more left to do; sequences per deal group;
 * single tiles or units of a sequence's k-th longest and k-th shortest tile),
 * "gemm_rows_per_wave" / "gemm_waves" / "gemm_slab_nt" for the decode GEMM — results are
 * identical for every setting of these.  "decode_gqa" = 0|1 selects the per-query-head or the
 * grouped-query decode kernel for n_heads > n_kv_heads (both within the stated tolerance; the
 * grouped kernel rounds P to T before P.V like the prefill kernel). */
int hx_debug_set_option(const char* name, int value);

/* ------------------------------------------------------------------------
 * Paged cache scatter.
 * replaces: csrc/kernel/kv_cache_kernels/kv_cache_kernels.cu:60-95 (set_kv_cache)
 *           hydrainfer/_C/kernel/kv_cache_kernels/__init__.pyi:5-10
 * cache[slot / block_size, slot % block_size, h, d] = src[token, h, d] for K and V.
 * keys/values: [n_tokens, n_kv_heads, head_dim], last two dims contiguous, row
 * stride k_stride / v_stride elements.  Caches: [n_blocks, block_size, n_kv_heads,
 * head_dim] with block stride cache_block_stride elements (rows inside a block
 * contiguous).  Bit-exact copy; slots < 0 are skipped.
 * ---------------------------------------------------------------------- */
int hx_set_kv_cache(const int32_t* slot_ids, const void* keys, const void* values,
                    void* key_cache, void* value_cache,
                    int64_t n_tokens, int64_t n_kv_heads, int64_t head_dim,
                    int64_t block_size, int64_t k_stride, int64_t v_stride,
                    int64_t kcache_block_stride, int64_t vcache_block_stride,
                    int dtype, hx_stream stream);

/* replaces: csrc/kernel/cache_kernels/cache_kernels.cu:55-83 (set_image_cache)
 *           hydrainfer/_C/kernel/cache_kernels/__init__.pyi:4-7 */
int hx_set_image_cache(const int32_t* slot_ids, const void* image_tokens, void* image_cache,
                       int64_t n_tokens, int64_t n_heads, int64_t head_dim,
                       int64_t block_size, int64_t token_stride, int64_t cache_block_stride,
                       int dtype, hx_stream stream);

/* ------------------------------------------------------------------------
 * replaces: csrc/kernel/norm/rms_norm.cu:43-63 (rms_norm)
 *           hydrainfer/_C/kernel/norm/__init__.pyi:4-9
 * out[r,i] = (T)(x[r,i] * rsqrt(mean_i(x[r,:]^2) + eps)) * w[i]   (T arithmetic for
 * the weight multiply, fp32 reduction) — rounding points of rms_norm.cu:39.
 * out/input contiguous [rows, hidden]; dtype f32/f16/bf16 (bf16 is an extension).
 * ---------------------------------------------------------------------- */
int hx_rms_norm(void* out, const void* input, const void* weight, float epsilon,
                int64_t rows, int64_t hidden, int dtype, hx_stream stream);

/* Extension: h = residual + x written back to `residual`,
 * out = rms_norm(h).  Same rounding as add-then-hx_rms_norm. */
int hx_add_rms_norm(void* out, void* residual, const void* x, const void* weight,
                    float epsilon, int64_t rows, int64_t hidden, int dtype, hx_stream stream);

/* ------------------------------------------------------------------------
 * replaces: csrc/kernel/position_embedding/rope.cu:82-117 (apply_rotary_pos_emb)
 *           hydrainfer/_C/kernel/position_embedding/__init__.pyi:5-11
 * In-place rotation of query [n_tokens, n_heads, head_dim] and key
 * [n_tokens, n_kv_heads, head_dim] (row strides q_stride / k_stride elements, last two
 * dims contiguous).  cos_sin: [max_positions, 2, rotary_dim/2] in the tensors' dtype.
 * x' = x*c - y*s ; y' = x*s + y*c, every operation rounded to T (rope.cu:22-27).
 * interleaved ? (x,y)=(2i,2i+1) : (i, i+rotary_dim/2).
 * ---------------------------------------------------------------------- */
int hx_apply_rotary_pos_emb(void* query, void* key, const int32_t* positions,
                            const void* cos_sin, int64_t n_tokens, int64_t n_heads,
                            int64_t n_kv_heads, int64_t head_dim, int64_t rotary_dim,
                            int64_t q_stride, int64_t k_stride, int interleaved,
                            int dtype, hx_stream stream);

/* Extension: apply_rotary_pos_emb (NeoX layout) on query/key in place AND
 * set_kv_cache(slot_ids, key, value) in one launch — the two consecutive calls of
 * hydrainfer/model/model_forward.py:78-83 + hydrainfer/layer/causal_attention.py:401-403.
 * Bit-identical to the two separate ops.  Vector path only (dims multiples of 16 bytes). */
int hx_rope_set_kv_cache(void* query, void* key, const void* value, const int32_t* positions,
                         const void* cos_sin, const int32_t* slot_ids, void* key_cache,
                         void* value_cache, int64_t n_tokens, int64_t n_heads, int64_t n_kv_heads,
                         int64_t head_dim, int64_t rotary_dim, int64_t q_stride, int64_t k_stride,
                         int64_t v_stride, int64_t block_size, int64_t kcache_block_stride,
                         int64_t vcache_block_stride, int dtype, hx_stream stream);

/* ------------------------------------------------------------------------
 * replaces: csrc/kernel/activation/activation.cu:52-56 (silu)
 *           hydrainfer/_C/kernel/activation/__init__.pyi:3-4
 * out[r,i] = (T)(x / (1 + exp(-x))) computed in fp32; input rows strided by
 * in_stride elements, out contiguous [rows, n].
 * ---------------------------------------------------------------------- */
int hx_silu(void* out, const void* input, int64_t rows, int64_t n, int64_t in_stride,
            int dtype, hx_stream stream);

/* Extension: out = (T)silu(gate) * up in T arithmetic, i.e. exactly
 * `silu(gate_proj(h)) * up_proj(h)` of hydrainfer/model/model_forward.py:36. */
int hx_silu_and_mul(void* out, const void* gate, const void* up, int64_t rows, int64_t n,
                    int64_t gate_stride, int64_t up_stride, int dtype, hx_stream stream);

/* Extensions for the vision tower (CLIP ViT; round 6): the reference runs these as separate torch ops, at 577 x 1024 each
 * a ~5 us launch-bound kernel.
 * hx_quick_gelu: out = x * sigmoid(1.702 x) — hydrainfer/layer/activation.py:17-22 (QuickGELU) — with its three T
 *   roundings (scaled copy, sigmoid, product); input rows strided by in_stride elements, out contiguous [rows, n];
 *   n and in_stride multiples of 8 (4 for fp32).
 * hx_add_layer_norm: residual[r,:] += x[r,:] (one T rounding, in place); out[r,:] = (T)((h - mean) * rstd * weight + bias),
 *   moments in fp32 over the T-rounded h, biased variance — `h = h + y; x = nn.LayerNorm(h)` of the encoder layer
 *   (hydrainfer/model/clip.py: CLIPEncoderLayer.forward) in one pass.  x == NULL: plain layer norm of `residual`
 *   (left untouched).  Contiguous rows, hidden % 8 == 0 (4 for fp32), hidden <= 8192 (4096 for fp32). */
int hx_quick_gelu(void* out, const void* input, int64_t rows, int64_t n, int64_t in_stride, int dtype, hx_stream stream);
int hx_add_layer_norm(void* out, void* residual, const void* x, const void* weight, const void* bias, float epsilon,
                      int64_t rows, int64_t hidden, int dtype, hx_stream stream);

/* ------------------------------------------------------------------------
 * Extension: decode-batch linear layer  out[M,N] = x[M,K] @ weight[N,K]^T  (M <= 64) as a
 * weight-streaming HIP kernel — the nn.Linear calls of hydrainfer/model/llama.py:24-27,48-50
 * at decode batch sizes.  fp32 accumulation, split-K partials summed in a fixed order.
 * Constraints: N % 16 == 0, K % 256 == 0, f16/bf16.  workspace >=
 * hx_linear_decode_workspace_bytes(M, N, K).  Strides in elements.
 * ---------------------------------------------------------------------- */
int64_t hx_linear_decode_workspace_bytes(int64_t M, int64_t N, int64_t K);
int hx_linear_decode(void* out, const void* x, const void* weight, int64_t M, int64_t N,
                     int64_t K, int64_t ldx, int64_t ldw, int64_t ldo, void* workspace,
                     int64_t workspace_bytes, int dtype, hx_stream stream);
/* ONE ENTRY for the decode-batch linear layer with pre-packed weights (what a maintainer binds; the layout-specific
 * entry points further down are what it dispatches to and what the fused decode layer of hydrainfer_amd/model uses).
 *   hx_decode_weight w;
 *   hx_decode_weight_plan(&w, N, K, dtype, max_rows, flags);      // host only: picks the layout for batches <= max_rows
 *   hx_decode_weight_pack(&w, packed_dev, weight, ldw, stream);   // once, at model load; packed_dev: N*K elements
 *   n_slabs = hx_linear_decode_ex(partial, bytes, x, ldx, 0, &w, M, stream);          // every step: partial[s][M][N] fp32
 * followed by a slab consumer (hx_add_rms_norm_slabs, hx_silu_and_mul_slabs, hx_decode_attention_fused, or a plain
 * sum).  layout: activations-in-registers (HX_DW_XREG: max_rows <= 32 and hx_linear_decode_xreg_supported) else
 * LDS-slice (HX_DW_LDS_SLICE: N % 16 == 0, K % 256 == 0, max_rows <= 64).  HX_DW_GATE_UP marks a [gate; up] weight:
 * on the XREG layout its halves are interleaved for hx_gate_up_silu_xreg / hx_norm_gate_up_silu_xreg (flag kept in
 * w->flags; hx_linear_decode_ex refuses such a packing), otherwise the flag is cleared and the rows stay in order.
 * x may be fragment-major (see hx_linear_decode_partial_xreg) on the XREG layout only.
 * Replaces: torch.nn.functional.linear of hydrainfer/model/llama.py:24-27,48-50 at decode batch sizes. */
typedef struct hx_decode_weight {
  const void* packed;   /* device, N*K elements; set by hx_decode_weight_pack */
  int64_t N, K;
  int32_t dtype;        /* HX_F16 | HX_BF16 */
  int32_t layout;       /* HX_DW_* chosen by hx_decode_weight_plan */
  int32_t flags;
  int32_t max_rows;
} hx_decode_weight;
#define HX_DW_LDS_SLICE 0
#define HX_DW_XREG 1
#define HX_DW_GATE_UP 1
#define HX_DW_FORCE_LDS_SLICE 2   /* plan flag only (not kept in w->flags): the LDS-slice layout whatever the batch range — the
                                   * consumer hands over ROW-major activations (the o projection behind the attention output) */
int hx_decode_weight_plan(hx_decode_weight* w, int64_t N, int64_t K, int dtype, int max_rows, int flags);
int hx_decode_weight_pack(hx_decode_weight* w, void* packed, const void* weight, int64_t ldw, hx_stream stream);
int64_t hx_linear_decode_ex_workspace_bytes(const hx_decode_weight* w, int64_t M);
int hx_linear_decode_ex(float* partial, int64_t partial_bytes, const void* x, int64_t ldx, int x_fragment_major,
                        const hx_decode_weight* w, int64_t M, hx_stream stream);
/* Same GEMM, but the fp32 split-K slabs partial[s][M][N] are left for a fused consumer.
 * Returns the number of slabs (>= 1) or a negative hx_status. */
int hx_linear_decode_partial(float* partial, const void* x, const void* weight, int64_t M,
                             int64_t N, int64_t K, int64_t ldx, int64_t ldw,
                             int64_t partial_bytes, int dtype, hx_stream stream);
/* The same GEMM on weights PACKED for streaming: hx_pack_decode_weight copies weight [N, K] (row
 * stride ldw elements) into packed [N*K], where the 1 KiB block (n/16, k/32) holds at lane
 * l = (r = l & 15, g = l >> 4) the 8 elements weight[16*(n/16) + r][32*(k/32) + 8g .. + 8] — the
 * MFMA A operand of that (row group, k-step) — and the blocks are ordered [K split of 1024][row
 * group][k-step in split].  The kernel then reads each (row group, K split) as one contiguous run,
 * adjacent to its neighbours' (the fastest read shape on MI355X, tools/bench_stream.py), and needs
 * no transpose.  Results are bit-identical to hx_linear_decode_partial.  N % 16 == 0, K % 256 == 0. */
int hx_pack_decode_weight(void* packed, const void* weight, int64_t N, int64_t K, int64_t ldw,
                          int dtype, hx_stream stream);
int hx_linear_decode_partial_packed(float* partial, const void* x, const void* packed_weight,
                                    int64_t M, int64_t N, int64_t K, int64_t ldx,
                                    int64_t partial_bytes, int dtype, hx_stream stream);
/* Step-edge fusions of a decode loop (extensions, bit-identical to the ops they replace).
 * hx_embed_rms_norm: h_out[r] = table[ids[r]] (torch.nn.functional.embedding; ids int32 or int64, out-of-range
 * ids clamp) and x_out[r] = rms_norm(h_out[r]) * weight, one launch (hydrainfer/model/llama.py:80-83 + layer/norm.py).
 * hx_argmax_rows: out[r] = argmax(logits[r, :n]) with torch.argmax's rule (NaN largest, ties to the smallest
 * index): the greedy sampler of hydrainfer/model/llama.py:99-104. */
int hx_embed_rms_norm(void* h_out, void* x_out, const void* ids, int ids_are_int64, const void* table,
                      const void* weight, float epsilon, int64_t rows, int64_t hidden, int64_t vocab,
                      int dtype, hx_stream stream);
int hx_argmax_rows(int64_t* out, const void* logits, int64_t rows, int64_t n, int64_t ld, int dtype,
                   hx_stream stream);
/* The same product for M <= 32 with the activations held in REGISTERS (csrc/gemm_xreg.hip): a
 * workgroup spans the whole K of its split, so K <= 4096 needs ONE slab (no K split) and K = 11008
 * three instead of eleven — the fp32 slab traffic of a decode layer drops from 25 MB to 6.5 MB.
 * The weight must be packed by hx_pack_decode_weight_xreg (same 1 KiB fragment format as
 * hx_pack_decode_weight, blocks ordered [K split][row group][k-step in split] with the split size
 * this kernel chooses for K; interleave_halves != 0 alternates the 16-row groups of the two halves
 * of N — gate and up of a fused gate|up weight — for hx_gate_up_silu_xreg).
 * x is row-major [M, K] (row stride ldx) or, with x_fragment_major != 0, FRAGMENT-MAJOR: the 16-byte
 * piece ((k/32) * MB + m/16) * 64 + ((k%32)/8) * 16 + m%16, MB = ceil(M/16), holds x[m][8*(k/8) .. +8]
 * — the MFMA B operands in the order the kernel loads them (buffer: hx_fragment_major_elems(M, K)
 * elements; rows M .. 16*MB-1 are never read for a stored result).  hx_add_rms_norm_slabs_ex /
 * hx_silu_and_mul_slabs_ex / hx_gate_up_silu_xreg produce that layout.
 * hx_linear_decode_partial_xreg returns the number of slabs written (>= 1) or a negative HX_ERR_*.
 * Accumulation order differs from hx_linear_decode_partial (per-wave k ranges, then the four waves
 * in order): deterministic, not bit-identical to it.  N % 16 == 0, K % 32 == 0.
 * hx_gate_up_silu_xreg: act = silu(x Wg^T) * (x Wu^T) in one launch for K with a single split
 * (hx_gate_up_silu_xreg_supported), act fragment-major [inter]; rounding exactly as
 * hx_linear_decode_partial_xreg -> hx_silu_and_mul_slabs (bit-identical, tested).
 * Replaces: torch.nn.functional.linear at decode batch sizes and silu(gate) * up
 * (hydrainfer/model/llama.py:24-27,48-50, model_forward.py:36). */
int hx_linear_decode_xreg_supported(int64_t M, int64_t N, int64_t K);
int hx_linear_decode_xreg_splits(int64_t N, int64_t K);
int64_t hx_linear_decode_xreg_workspace_bytes(int64_t M, int64_t N, int64_t K);
int64_t hx_fragment_major_elems(int64_t rows, int64_t K);
int hx_pack_decode_weight_xreg(void* packed, const void* weight, int64_t N, int64_t K, int64_t ldw,
                               int interleave_halves, int dtype, hx_stream stream);
int hx_linear_decode_partial_xreg(float* partial, const void* x, const void* packed_weight,
                                  int64_t M, int64_t N, int64_t K, int64_t ldx, int x_fragment_major,
                                  int64_t partial_bytes, int dtype, hx_stream stream);
/* hx_add_rms_norm_slabs FUSED IN FRONT of the product (one launch instead of two): workgroup r < M of
 * the GEMM grid first computes row r of
 *     residual += (T) sum of the n_splits_in slabs of slabs_in [n_splits_in][M][K];  x = rms_norm(residual) * norm_weight
 * (bit-identical to hx_add_rms_norm_slabs_ex with fragment-major output), writes it to x_frag
 * (hx_fragment_major_elems(M, K) elements, a scratch the caller may read afterwards) and counts itself
 * in at `sync`; all workgroups prefetch their weights meanwhile, wait for the last row, then load x.
 * sync: HX_XREG_SYNC_WORDS int32 words, ZERO before the launch, one area per launch in flight (word 1
 * is set if a workgroup gave up waiting after 1 s — never expected).  Needs K in one split of the packing
 * (hx_norm_xreg_supported; M <= 32 for the gate|up + silu*mul form, M <= 64 for the plain product).  slabs_in must not alias the outputs.  Results equal
 * hx_add_rms_norm_slabs_ex + hx_linear_decode_partial_xreg resp. hx_gate_up_silu_xreg bit for bit. */
#define HX_XREG_SYNC_WORDS 512
int hx_norm_xreg_supported(int64_t M, int64_t N, int64_t K, int gate_up);
int hx_norm_linear_decode_xreg(float* partial, void* residual, const float* slabs_in, int32_t n_splits_in,
                               const void* norm_weight, float epsilon, void* x_frag,
                               const void* packed_weight, int64_t M, int64_t N, int64_t K, void* sync,
                               int64_t partial_bytes, int dtype, hx_stream stream);
int hx_norm_gate_up_silu_xreg(void* act, void* residual, const float* slabs_in, int32_t n_splits_in,
                              const void* norm_weight, float epsilon, void* x_frag,
                              const void* packed_gate_up, int64_t M, int64_t inter, int64_t K, void* sync,
                              int dtype, hx_stream stream);
/* Batches of 33 .. 64 rows (the reference's linear layers have no batch limit: hydrainfer/model/llama.py:24-27,48-50):
 * hx_linear_decode_partial_xreg and hx_norm_linear_decode_xreg take M <= 64 — above 32 rows a workgroup spans HALF of a
 * packed K split (x for 64 rows fills the registers at half the k-steps) over the SAME packing, so the slab count doubles
 * (2 for K <= 4096) and no second copy of the weights is needed; hx_*_supported / _workspace_bytes answer per M.  The
 * gate|up product then comes WITHOUT the fused silu*mul (it needs the whole K): hx_gate_up_xreg / hx_norm_gate_up_xreg
 * multiply by the interleaved gate|up packing and write fp32 slabs [n_splits][M][2*inter] in plain [gate | up] column
 * order for hx_silu_and_mul_slabs; they return the slab count (any M <= 64). */
int hx_gate_up_xreg_supported(int64_t M, int64_t inter, int64_t K, int with_norm);
int64_t hx_gate_up_xreg_workspace_bytes(int64_t M, int64_t inter, int64_t K);
int hx_gate_up_xreg(float* partial, const void* x, const void* packed_gate_up, int64_t M, int64_t inter, int64_t K,
                    int64_t ldx, int x_fragment_major, int64_t partial_bytes, int dtype, hx_stream stream);
int hx_norm_gate_up_xreg(float* partial, void* residual, const float* slabs_in, int32_t n_splits_in,
                         const void* norm_weight, float epsilon, void* x_frag, const void* packed_gate_up,
                         int64_t M, int64_t inter, int64_t K, void* sync, int64_t partial_bytes, int dtype,
                         hx_stream stream);
/* hx_norm_gate_up_silu_wide_xreg (round 5): the norm-fused gate|up product of 33 .. 64 rows WITH silu * mul — one launch
 * and no slabs instead of hx_norm_gate_up_xreg + hx_silu_and_mul_slabs.  A workgroup does BOTH K halves of its (gate, up)
 * row-group pairs one after the other (x of the first half, all its units — their reduced tiles wait in LDS —, x of the
 * second half, the units again) and writes act (fragment-major, ceil(M / 16) row blocks, as hx_silu_and_mul_slabs_ex with
 * out_fragment_major) with the same order of summation and the same roundings: bit-identical to the two-launch form.
 * Supported where the packing's one split is two full halves of 16 k-steps per wave (K = 4096: LLaVA-1.5-7B). */
int hx_gate_up_silu_wide_xreg_supported(int64_t M, int64_t inter, int64_t K);
int hx_norm_gate_up_silu_wide_xreg(void* act, void* residual, const float* slabs_in, int32_t n_splits_in,
                                   const void* norm_weight, float epsilon, void* x_frag,
                                   const void* packed_gate_up, int64_t M, int64_t inter, int64_t K, void* sync,
                                   int dtype, hx_stream stream);
int hx_gate_up_silu_xreg_supported(int64_t M, int64_t inter, int64_t K);
int hx_gate_up_silu_xreg(void* act, const void* x, const void* packed_gate_up, int64_t M,
                         int64_t inter, int64_t K, int64_t ldx, int x_fragment_major, int dtype,
                         hx_stream stream);
/* Slab consumers: sum the n_splits slabs in order, round once to T (the projection's output
 * rounding), then behave exactly like hx_add_rms_norm / hx_silu_and_mul on that tensor.
 * partial: [n_splits][rows][hidden] resp. [n_splits][rows][2*inter] (gate | up columns). */
int hx_add_rms_norm_slabs(void* out, void* residual, const float* partial, int32_t n_splits,
                          const void* weight, float epsilon, int64_t rows, int64_t hidden,
                          int dtype, hx_stream stream);
int hx_silu_and_mul_slabs(void* out, const float* partial, int32_t n_splits, int64_t rows,
                          int64_t inter, int dtype, hx_stream stream);
/* The same two consumers with the output optionally FRAGMENT-MAJOR (see hx_linear_decode_partial_xreg;
 * hidden resp. inter % 32 == 0): values bit-identical to the row-major form. */
int hx_add_rms_norm_slabs_ex(void* out, void* residual, const float* partial, int32_t n_splits,
                             const void* weight, float epsilon, int64_t rows, int64_t hidden,
                             int dtype, int out_fragment_major, hx_stream stream);
int hx_silu_and_mul_slabs_ex(void* out, const float* partial, int32_t n_splits, int64_t rows,
                             int64_t inter, int dtype, int out_fragment_major, hx_stream stream);

/* ------------------------------------------------------------------------
 * Variable-length attention forward, dense or paged.
 * replaces: csrc/kernel/flash_attn/flash_api.cpp:216-355 (mha_varlen_fwd)
 *           hydrainfer/_C/kernel/flash_attn/__init__.pyi:23-40
 *
 * out, q : [n_tokens, n_heads, head_dim], row strides o_row_stride / q_row_stride,
 *          head stride = head_dim (heads contiguous).
 * dense  : block_table == NULL; k, v: [total_k, n_kv_heads, head_dim] with
 *          kv_row_stride / kv_head_stride.
 * paged  : block_table != NULL (flat int32 [sum blocks]) with cu_block_lens int32
 *          [batch+1]; k, v: [n_blocks, block_size, n_kv_heads, head_dim] with
 *          kv_block_stride / kv_row_stride / kv_head_stride; block_size % 16 == 0.
 * cu_seqlens_q / cu_seqlens_k: int32 [batch+1] (device).
 * causal != 0: key j visible to query i iff j <= i + (kv_len - q_len)
 *          (bottom-right aligned; reference mask.h:173-193, window (-1,0)).
 * softcap and local windows: see the tail of hx_attn_args (flags & HX_ATTN_LOCAL_WINDOW).
 * Unsupported reference feature (HX_ERR_UNSUPPORTED): alibi slopes.
 * workspace: device scratch of at least hx_mha_varlen_fwd_workspace_bytes() bytes
 * (used for split-KV partials; may be NULL when that returns 0).
 * ---------------------------------------------------------------------- */
typedef struct hx_attn_args {
  void* out;
  const void* q;
  const void* k;
  const void* v;
  const int32_t* cu_seqlens_q;
  const int32_t* cu_seqlens_k;
  const int32_t* block_table;   /* NULL => dense */
  const int32_t* cu_block_lens; /* required iff block_table */
  int32_t batch;
  int32_t n_heads;
  int32_t n_kv_heads;
  int32_t head_dim;
  int32_t block_size;    /* paged only */
  int32_t max_seqlen_q;
  int32_t max_seqlen_k;
  int32_t total_q;       /* rows of q */
  int64_t q_row_stride;
  int64_t o_row_stride;
  int64_t k_block_stride, k_row_stride, k_head_stride;
  int64_t v_block_stride, v_row_stride, v_head_stride;
  float softmax_scale;
  int32_t causal;
  int32_t dtype;         /* HX_F16 | HX_BF16 */
  int32_t num_splits;    /* 0 = choose automatically, 1 = never split */
  void* workspace;
  int64_t workspace_bytes;
  /* flash_api.cpp:93-111.  softcap > 0: scores = softcap * tanh(q.k * softmax_scale / softcap)
   * before masking; 0 = off.  Local attention only when flags & HX_ATTN_LOCAL_WINDOW: query row i (of
   * q_len rows, kv_len keys) sees keys [i + kv_len - q_len - window_left, i + kv_len - q_len + window_right]
   * (mask.h:173-193); a negative bound = unbounded on that side; `causal` must be 0 then (causal IS the
   * window (-1, 0)).  Without the flag window_left / window_right are ignored, so an all-zero tail of
   * the struct (the usual C initialisation) means "no softcap, no window". */
  float softcap;
  int32_t window_left;
  int32_t window_right;
  int32_t flags;         /* HX_ATTN_* bits; unknown bits -> HX_ERR_UNSUPPORTED */
} hx_attn_args;
#define HX_ATTN_LOCAL_WINDOW 1

int64_t hx_mha_varlen_fwd_workspace_bytes(const hx_attn_args* args);
int hx_mha_varlen_fwd(const hx_attn_args* args, hx_stream stream);

/* Extension (SURVEY §8f-2 taken to the end): one launch for the three consecutive steps of a
 * decode layer — apply_rotary_pos_emb(q, k) (NeoX, rotary_dim == head_dim), set_kv_cache(slots,
 * k, v) and the paged attention itself (hydrainfer/model/model_forward.py:78-84 +
 * hydrainfer/layer/causal_attention.py:401-406).  `args` is an all-decode batch (q_len 1) whose
 * cu_seqlens_k already counts the new token; args->q, k_new, v_new are UN-rotated; the kernel
 * rotates in registers (same T-arithmetic rounding), attends with the new token taken from
 * registers and appends its rotated key / value to the cache at new_cache_slots.
 * Results are bit-identical to the three separate ops. */
typedef struct hx_fused_decode_args {
  const void* k_new;          /* [batch, n_kv_heads, head_dim] */
  const void* v_new;          /* [batch, n_kv_heads, head_dim] */
  int64_t k_new_row_stride;
  int64_t v_new_row_stride;
  const int32_t* positions;   /* [batch] */
  const void* cos_sin;        /* [max_pos, 2, head_dim/2] in the tensors' dtype */
  const int32_t* new_cache_slots; /* [batch] */
  int32_t rotary_dim;
  int32_t interleaved;
  /* optional: take q / k_new / v_new from the split-K slabs of the fused qkv projection
   * ([qkv_splits][batch][(n_heads + 2*n_kv_heads) * head_dim] fp32, hx_linear_decode_partial);
   * args->q, k_new, v_new are then ignored */
  const float* qkv_partial;
  int32_t qkv_splits;
  /* optional (may be NULL): the batch's RANK DESCRIPTOR, int32 [1 + batch] on the device — [0] = 1 when the batch is
   * ragged, [1 + r] = the sequence with the r-th most keys (hx_decode_rank / hx_decode_advance_ranked /
   * hx_step_head_args.rank_desc write it).  With it a big ragged batch is laid over the CUs in length-ranked snake
   * order (same results bit for bit, csrc/attn_decode.hip RANKED); without it: the static (head, sequence) grid. */
  const int32_t* rank_desc;
} hx_fused_decode_args;
int hx_decode_attention_fused(const hx_attn_args* args, const hx_fused_decode_args* fused,
                              hx_stream stream);

/* ------------------------------------------------------------------------
 * Cache-block migration between GPUs / processes.
 * replaces: csrc/data_transfer/block_migration.cpp:55-59 (get_ipc_mem_handle),
 *           :69-80 (register_ipc_mem_handle), :194-245 (migrate_blocks)
 *           hydrainfer/_C/data_transfer/block_migration/__init__.pyi:6-18
 * ---------------------------------------------------------------------- */
#define HX_IPC_HANDLE_BYTES 64
/* Writes the 64-byte hipIpcMemHandle_t of the allocation containing dev_ptr and the
 * byte offset of dev_ptr inside it. */
int hx_ipc_get_mem_handle(const void* dev_ptr, uint8_t handle_out[HX_IPC_HANDLE_BYTES],
                          int64_t* offset_out);
/* Opens (or returns the cached mapping of) a peer handle.  Mappings are cached by
 * handle bytes for the life of the process: the reference re-opens per call
 * (block_migration.cpp:213-215) which HIP rejects for an already-open handle. */
int hx_ipc_open_mem_handle(const uint8_t handle[HX_IPC_HANDLE_BYTES], void** dev_ptr_out);
int hx_ipc_close_all(void);

/* dst[l, t, dst_table[i]] = src[l, t, src_table[i]] for every layer l, token-kind t
 * (k/v) and i < n_pairs, as ONE gather-copy kernel per <=HX_MIGRATE_MAX_PAIRS pairs.
 * Pools are contiguous 6-D (n_layers, n_tokens, n_blocks, block_size, n_heads, head_size);
 * block_bytes = block_size*n_heads*head_size*itemsize (multiple of 16).
 * src may be an IPC-mapped peer pointer (xGMI read) or a local pointer. */
#define HX_MIGRATE_MAX_PAIRS 448
int hx_migrate_blocks(const int32_t* src_table_host, const int32_t* dst_table_host,
                      int64_t n_pairs, const void* src_pool, void* dst_pool,
                      int64_t n_layers, int64_t n_tokens, int64_t src_n_blocks,
                      int64_t dst_n_blocks, int64_t block_bytes, hx_stream stream);

/* Pack / unpack selected blocks to / from a contiguous staging buffer
 * [n_layers, n_tokens, n_pairs, block_bytes] — the RCCL send/recv path
 * (replaces the per-(block,layer,k/v) P2POp list of
 * hydrainfer/memory/communication.py:57-74). */
int hx_pack_blocks(const int32_t* table_host, int64_t n_pairs, const void* pool, void* staging,
                   int64_t n_layers, int64_t n_tokens, int64_t n_blocks, int64_t block_bytes,
                   hx_stream stream);
int hx_unpack_blocks(const int32_t* table_host, int64_t n_pairs, const void* staging, void* pool,
                     int64_t n_layers, int64_t n_tokens, int64_t n_blocks, int64_t block_bytes,
                     hx_stream stream);

/* The same three copies over pools whose (layer, k/v) PLANES are not back to back: plane p of a pool starts
 * p * plane_bytes past the pool's base and holds n_blocks blocks contiguously (plane_bytes >= n_blocks * block_bytes,
 * multiple of 16; n_planes = n_layers * n_tokens).  The entries above are these with plane_bytes = n_blocks * block_bytes —
 * the reference's contiguous 6-D tensor (hydrainfer/memory/token_cache_manger.py:65).
 * Why a pool would want spare bytes between planes (round 5, tools/probes/attn_placement.py): K and V of one
 * (block, token, head) are read by the same wave at the same time, and in the contiguous pool they lie n_blocks *
 * block_bytes apart — a large power-of-two multiple — so both land on the same HBM channel; with the V plane an odd
 * multiple of 256 bytes further on, decode attention of 64 sequences runs 5 % faster, the 64-row decode step 3 %, the
 * 13B step 1 % (hydrainfer_amd/memory/kv_pool.py allocates pools that way; the plane stride travels with the pool's
 * IPC handle). */
int hx_migrate_blocks_planes(const int32_t* src_table_host, const int32_t* dst_table_host,
                             int64_t n_pairs, const void* src_pool, void* dst_pool, int64_t n_planes,
                             int64_t src_n_blocks, int64_t dst_n_blocks, int64_t src_plane_bytes,
                             int64_t dst_plane_bytes, int64_t block_bytes, hx_stream stream);
int hx_pack_blocks_planes(const int32_t* table_host, int64_t n_pairs, const void* pool, void* staging,
                          int64_t n_planes, int64_t n_blocks, int64_t pool_plane_bytes, int64_t block_bytes,
                          hx_stream stream);
int hx_unpack_blocks_planes(const int32_t* table_host, int64_t n_pairs, const void* staging, void* pool,
                            int64_t n_planes, int64_t n_blocks, int64_t pool_plane_bytes, int64_t block_bytes,
                            hx_stream stream);

/* ------------------------------------------------------------------------
 * Decode-step metadata advance: device-resident equivalent of one
 * AttentionParametersBuilder pass for an all-decode batch
 * (hydrainfer/layer/causal_attention.py:147-168).  For every sequence b:
 *   positions[b] += stride; kv_len[b] += stride; cu_seqlens_k = prefix-sum(kv_len);
 *   new_cache_slots[b] = block_table[cu_block_lens[b] + pos/bs]*bs + pos%bs.
 * stride = 1 is the reference's step; a benchmark that samples the generation's contexts at a fixed spacing
 * uses a larger one.  Lets a whole decode step live inside one hipGraph / launch plan.
 * hx_decode_feed_ids: the step's input ids with one step of look-ahead (engine/graph_decode.py): row r takes the
 * token the PREVIOUS launch sampled for row src[r] (prev, int64, still on the device) when src[r] >= 0, else the
 * host-written ids[r].  hx_collect_errors: out[0] = OR over i < n_areas of areas[i * stride_words + word], OR
 * extra[0] if given — the give-up words of a step's in-kernel hand-overs folded into ONE word that travels to the
 * host with the step's tokens.
 * ---------------------------------------------------------------------- */
int hx_decode_advance(int32_t* positions, int32_t* kv_lens, int32_t* cu_seqlens_k,
                      int32_t* new_cache_slots, const int32_t* block_table,
                      const int32_t* cu_block_lens, int32_t batch, int32_t block_size,
                      int32_t stride, hx_stream stream);
/* hx_decode_advance that also leaves the advanced batch's rank descriptor (hx_fused_decode_args.rank_desc) in rank_desc
 * (int32 [1 + batch]); hx_decode_rank: the descriptor alone, from cu_seqlens_k (int32 [batch + 1]).  A batch of more
 * than 256 sequences is declared even ([0] = 0).  The engine's host-built steps write the same words from the host
 * (hydrainfer_amd/layer/causal_attention.py::decode_rank_descriptor). */
int hx_decode_advance_ranked(int32_t* positions, int32_t* kv_lens, int32_t* cu_seqlens_k,
                             int32_t* new_cache_slots, const int32_t* block_table,
                             const int32_t* cu_block_lens, int32_t batch, int32_t block_size,
                             int32_t stride, int32_t* rank_desc, hx_stream stream);
int hx_decode_rank(const int32_t* cu_seqlens_k, int32_t batch, int32_t* rank_desc, hx_stream stream);
/* hx_decode_step_head: everything a decode step does before its first GEMM as ONE launch — hx_embed_rms_norm for
 * `rows` rows (each id optionally replaced by the previous launch's sample like hx_decode_feed_ids: feed_src /
 * feed_prev, both or neither; fed_out, if given, receives the ids used), hx_memset_zero of zero_bytes at zero_ptr
 * (the hand-over areas of the step's norm-fused launches; 0 = none) and hx_decode_advance for `batch` sequences
 * (0 = none).  The three parts touch disjoint memory; each is bit-identical to the separate call. */
typedef struct hx_step_head_args {
  void* h_out;               /* [rows, hidden] embedding rows */
  void* x_out;               /* [rows, hidden] rms_norm(h_out) * weight */
  const void* ids;           /* [rows] int32 or int64 */
  const int32_t* feed_src;   /* [rows] or NULL */
  const int64_t* feed_prev;  /* previous launch's samples, or NULL */
  int64_t* fed_out;          /* [rows] or NULL */
  const void* table;         /* [vocab, hidden] */
  const void* weight;        /* [hidden] */
  void* zero_ptr;            /* 16-byte aligned, or NULL */
  int64_t zero_bytes;        /* multiple of 4 */
  int32_t* positions;        /* hx_decode_advance arguments (batch = 0: unused) */
  int32_t* kv_lens;
  int32_t* cu_seqlens_k;
  int32_t* new_cache_slots;
  const int32_t* block_table;
  const int32_t* cu_block_lens;
  int64_t rows, hidden, vocab;
  float epsilon;
  int32_t ids_are_int64;
  int32_t dtype;             /* HX_F16 | HX_BF16 */
  int32_t batch, block_size, stride;
  int32_t* rank_desc;        /* NULL, or int32 [1 + batch]: the advanced batch's rank descriptor (hx_decode_advance_ranked) */
} hx_step_head_args;
int hx_decode_step_head(const hx_step_head_args* args, hx_stream stream);
/* hx_stage_decode: a decode step's integer inputs into the engine's RESIDENT device buffer (engine/graph_decode.py) —
 * the step's head (ids, positions, slots, cumulative lengths, table offsets, rank descriptor: head_words words, copied to
 * dst[0 ..)) and the block-table DELTAS behind it: staging[head_words] = n_runs, then per run [dst word offset] [count]
 * [count values].  A sequence's table stays where it is on the device; a step writes only the block ids that are new
 * (the reference rebuilds and re-copies every table every step: hydrainfer/engine/parameters_builder.py:46-97).
 * staging: host memory the device can read (pinned); dst_words bounds every write. */
int hx_stage_decode(void* dst, int64_t dst_words, const void* staging, int32_t head_words, hx_stream stream);
int hx_decode_feed_ids(int64_t* out, const int32_t* ids, const int32_t* src, const int64_t* prev,
                       int32_t n, hx_stream stream);
int hx_collect_errors(uint32_t* out, const uint32_t* areas, int32_t n_areas, int64_t stride_words,
                      int32_t word, const uint32_t* extra, hx_stream stream);
/* hx_copy_words2: dst0[0 .. n0) = src0[..], dst1[0 .. n1) = src1[..] (32-bit words, 4-byte aligned, n1 may be 0) as ONE
 * kernel launch.  Any of the four pointers may be host-mapped pinned memory (hipHostMalloc / torch pin_memory): this is how
 * the engine's decode loop moves a step's few hundred integers in and its sampled tokens out (engine/graph_decode.py) —
 * a hipMemcpyAsync between two graph launches cost the stream ~40-120 us of idle time per copy on this ROCm (round 5,
 * rocprofv3 trace of the serving leg: three copyBuffer launches per step with 9 / 0 / 123 us in front of them); a kernel
 * is just the next packet in the queue.  Results written to pinned memory are visible to the host once an event recorded
 * behind the launch has completed (the reference's step does a blocking .tolist(), hydrainfer/engine/executor.py). */
int hx_copy_words2(void* dst0, const void* src0, int32_t n0_words, void* dst1, const void* src1, int32_t n1_words,
                   hx_stream stream);

/* ------------------------------------------------------------------------
 * Launch plans: record the launches of a fixed sequence of hx_* calls once, replay them with one
 * native loop — the role of the hipGraph in the reference's unfinished
 * hydrainfer/model_runner/cuda_graph_model_runner.py:1-72.  Between hx_plan_begin and hx_plan_end every hx_* entry
 * point called on the recording thread appends its kernel launches (hx_memset_zero included) to the plan instead of
 * executing them; nothing runs, no stream is touched.  hx_plan_launch issues them on a stream, in order, with the
 * recorded arguments (pointers are recorded by value: the buffers must stay allocated and in place, like a graph's).
 * One recording per thread at a time; a plan may be replayed from any thread.  hx_plan_size: recorded launches.
 * ---------------------------------------------------------------------- */
typedef struct hx_plan hx_plan;
int hx_plan_begin(hx_plan** plan);
int hx_plan_end(hx_plan* plan);
int hx_plan_size(const hx_plan* plan);
int hx_plan_launch(const hx_plan* plan, hx_stream stream);
int hx_plan_destroy(hx_plan* plan);
/* memset(p, 0, bytes) on the stream as a kernel launch; recordable in a plan, capturable in a hipGraph. */
int hx_memset_zero(void* p, int64_t bytes, hx_stream stream);
/* Measurement aid (SURVEY §8d "measured streaming ceiling"; no reference counterpart): reads `bytes` bytes at p
 * once, the way the weight-streaming kernels of this library read — 1 KiB contiguous per wave instruction,
 * non-temporal, ONE 8 KiB chunk (eight 1 KiB loads) in flight per wave, consumed before the next eight are requested,
 * 512 workgroups of 4 waves — and does nothing with them.  bench.py times it to report the read rate this GPU reaches
 * in this run beside the 8 TB/s vendor peak.  bytes % 8192 == 0, p 16-byte aligned; sink: one float the kernel never
 * writes (keeps the loads alive). */
int hx_measure_read_stream(const void* p, int64_t bytes, float* sink, hx_stream stream);
/* Measurement aids for the "null layer" (round 5; tools/null_layer.py, bench.py `whole_step.null_step`): the decode
 * layer's launches with the arithmetic, the activations and the hand-overs removed — what ANY design that runs a layer
 * as the same number of launches over the same bytes can reach on this GPU.
 *   hx_measure_read_grid   the kernel of hx_measure_read_stream over `n_workgroups` workgroups (the real grid of the
 *                          weight-streaming launch it stands in for: 256 = one per CU);
 *   hx_measure_paged_read  the decode attention kernel's read pattern and grid — (head, sequence) workgroups of 4
 *                          waves, wave w owns the 16-key tiles w, w + 4, ..., a tile = one page of the paged cache
 *                          (block_size 16): this head's 256 B (head_bytes) of 16 K rows and 16 V rows at a pitch of
 *                          row_bytes, pages looked up in `table` (n_seq rows of table_stride int32), two tiles in
 *                          flight per wave — nothing computed, nothing written.  head_bytes must be 256.
 * Both record into launch plans like every other launch. */
int hx_measure_read_grid(const void* p, int64_t bytes, int n_workgroups, float* sink, hx_stream stream);
int hx_measure_paged_read(const void* kbase, const void* vbase, const int32_t* table, int64_t table_stride, int n_seq,
                          int n_heads, int tiles, int64_t page_bytes, int64_t row_bytes, int head_bytes, float* sink,
                          hx_stream stream);

/* ------------------------------------------------------------------------
 * MoE routing / permutation ops (named in north_star; no production caller in the reference).
 * replaces: csrc/kernel/moe/moe_kernel.h:6-39, moe_kernels_pybind.cpp:7-15
 *           hydrainfer/_C/kernel/moe/__init__.pyi:4-145
 * ---------------------------------------------------------------------- */
/* gating_logits f32 [n_tokens, n_experts] -> topk_weights f32 / topk_indices i32
 * [n_tokens, topk]: softmax over experts then iterative arg-max, lower index wins ties
 * (topk_softmax_kernel.cu:108-180).  n_experts <= 1024 (any value, not only powers of 2). */
int hx_topk_softmax(const float* gating_logits, float* topk_weights, int32_t* topk_indices,
                    int64_t n_tokens, int64_t n_experts, int64_t topk, hx_stream stream);
/* DeepSeek-V3 routing (grouped_topk_sigmoid_kernel.cu:15-181): score = sigmoid(logit);
 * choice = score + bias; drop (n_groups - topk_group) groups ranked by the sum of their two
 * largest choices (ties: higher group dropped first); top-k over the kept experts by choice
 * (ties: lower expert); weights = score of the chosen experts, NOT renormalised;
 * scaling_factor is accepted and, like the reference (:180), not applied. */
int hx_grouped_topk_sigmoid(const float* gating_logits, const float* correction_bias,
                            float* topk_weights, int32_t* topk_indices, int64_t n_tokens,
                            int64_t n_experts, int64_t n_groups, int64_t topk_group,
                            int64_t topk, float scaling_factor, hx_stream stream);
/* row_id_map i32 [topk, n_tokens] from a STABLE sort of the flattened topk_indices by expert
 * (permutation_index_kernel.cu:39-77): the element with flat index f = t*topk + k that lands at
 * sorted position p gives row_id_map[k*n_tokens + t] = p. */
int64_t hx_moe_sort_workspace_bytes(int64_t n_tokens, int64_t topk);
int hx_moe_row_id_map_from_indices(const int32_t* topk_indices, int32_t* row_id_map,
                                   int64_t n_tokens, int64_t topk, void* workspace,
                                   int64_t workspace_bytes, hx_stream stream);
/* row_id_map i32 [n_experts, n_tokens] from a boolean routing map [n_tokens, n_experts]
 * (permutation_mask_kernel.cu:43-130): running count in expert-major, token-minor order where
 * routed, -1 elsewhere.  workspace >= 4*n_experts bytes. */
int hx_moe_row_id_map_from_mask(const uint8_t* routing_map, int32_t* row_id_map,
                                int64_t n_tokens, int64_t n_experts, void* workspace,
                                int64_t workspace_bytes, hx_stream stream);
/* permuted[row_id_map[r*n_tokens + t], :] = tokens[t, :] for every map row r with entry >= 0
 * (n_rows = topk for an index map, n_experts for a mask map). */
int hx_moe_permute(const void* tokens, void* permuted, const int32_t* row_id_map,
                   int64_t n_tokens, int64_t n_rows, int64_t dim, int dtype, hx_stream stream);
/* out[t,:] = sum_r probs[t,r] * permuted[row_id_map[r*n_tokens + t], :] over entries >= 0,
 * product and running sum in T arithmetic like the reference's frag_sum
 * (permutation_index_kernel.cu:146-160); probs has the tokens' dtype, [n_tokens, n_rows],
 * NULL = weight 1. */
int hx_moe_unpermute(const void* permuted, void* out, const int32_t* row_id_map,
                     const void* probs, int64_t n_tokens, int64_t n_rows, int64_t dim,
                     int dtype, hx_stream stream);
/* out[t,:] = sum_k in[t,k,:] with the reference's arithmetic (align_block_kernel.cu:172-188,242-272): topk in {2,3,4,8} —
 * topk_sum_kernel's instantiations — keep the running sum in T (every partial sum rounded), any other topk accumulates in
 * fp32 and rounds once (torch::sum_out). */
int hx_moe_sum_out(const void* in, void* out, int64_t n_tokens, int64_t topk, int64_t dim,
                   int dtype, hx_stream stream);

#ifdef __cplusplus
}
#endif
#endif /* HYDRA_HIP_H */
