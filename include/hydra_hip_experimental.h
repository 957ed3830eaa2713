/*
 * hydra_hip_experimental.h — entry points that exist only in a library built with
 * `make -C hydrainfer_amd/csrc EXPERIMENTS=1`: kernels that were built, tested, measured and REJECTED
 * (DESIGN.md §6c), and the read-stream microbenchmarks behind tools/bench_stream.py /
 * tools/bench_attn_ceiling.py.  None of this is on the product path; the default libhydra_hip.so does
 * not export these symbols and hydrainfer_amd binds them only when they are present.
 *   hx_decode_chain          — o-proj .. next qkv as ticket-ordered work items of one launch (slower than
 *                              the separate launches: 118-122 vs 96 us per 7B layer)
 *   "decode_hpw4" option      — four heads per decode-attention workgroup (attn_decode4.hip: -4 % standalone,
 *                              +-0 in the decode step)
 *   hx_debug_stream_read / hx_debug_paged_read — access-shape probes
 */
#ifndef HYDRA_HIP_EXPERIMENTAL_H
#define HYDRA_HIP_EXPERIMENTAL_H

#include "hydra_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

/* ------------------------------------------------------------------------
 * Extension: the dense part of ONE decoder layer of a decode step as one launch
 * (hydrainfer/model/model_forward.py:84-105 for this layer + the qkv projection :72-77 of the
 * next one), M <= 32 rows:
 *   a = attn_out @ w_o^T;            h_mid = h_in + a;      x_post = rms_norm(h_mid) * norm_post
 *   gu = x_post @ w_gate_up^T;       act = silu(gu[:, :inter]) * gu[:, inter:]
 *   d = act @ w_down^T;              h_out = h_mid + d;     x_next = rms_norm(h_out) * norm_next
 *   qkv_partial[s] = split-K slabs of x_next @ w_qkv_next^T   (skipped when qkv_n == 0)
 * Same rounding points as hx_linear_decode_partial + hx_add_rms_norm_slabs +
 * hx_silu_and_mul_slabs run one after the other: bit-identical results.  The phases are work
 * items of one grid handed out by atomic ticket; `sync` is HX_CHAIN_SYNC_WORDS zeroed uint32
 * words per launch (the caller zeroes them on the same stream before the launch); word
 * HX_CHAIN_SYNC_ERR is non-zero afterwards if a dependency wait timed out (results invalid).
 * h_in, h_mid, h_out, x_post, x_next, act and attn_out must be distinct buffers.
 * Returns the number of qkv slabs (>= 0) or a negative hx_status.
 * ---------------------------------------------------------------------- */
#define HX_CHAIN_SYNC_WORDS 18432
#define HX_CHAIN_SYNC_ERR 480
typedef struct hx_chain_args {
  int32_t M;
  int32_t hidden;
  int32_t inter;
  int32_t q_size;            /* n_heads * head_dim = K of the o projection */
  int32_t qkv_n;             /* rows of w_qkv_next, 0 = no next layer */
  int32_t dtype;             /* HX_F16 | HX_BF16 */
  float eps;
  int32_t reserved;
  const void* attn_out;      /* [M, q_size], row stride attn_out_stride elements */
  int64_t attn_out_stride;
  const void* h_in;          /* [M, hidden] residual stream entering the layer's o projection */
  /* weights PACKED by hx_pack_decode_weight (the chain streams fragment-order weights only) */
  const void* w_o;           /* pack of [hidden, q_size] */
  const void* w_gate_up;     /* pack of [2*inter, hidden] */
  const void* w_down;        /* pack of [hidden, inter] */
  const void* w_qkv_next;    /* pack of [qkv_n, hidden] or NULL */
  const void* norm_post_weight;  /* [hidden] post-attention RMSNorm */
  const void* norm_next_weight;  /* [hidden] next layer's input RMSNorm (or the final norm) */
  void* h_mid;               /* [M, hidden] out */
  void* h_out;               /* [M, hidden] out: residual stream leaving the layer */
  void* x_post;              /* [M, hidden] out */
  void* act;                 /* [M, inter] out */
  void* x_next;              /* [M, hidden] out */
  float* qkv_partial;        /* [splits][M][qkv_n] fp32 out, as hx_linear_decode_partial */
  int64_t qkv_partial_bytes;
  void* workspace;           /* >= hx_decode_chain_workspace_bytes(...) */
  int64_t workspace_bytes;
  uint32_t* sync;            /* HX_CHAIN_SYNC_WORDS zeroed words */
} hx_chain_args;

int64_t hx_decode_chain_workspace_bytes(int64_t M, int64_t hidden, int64_t inter, int64_t q_size);
int hx_decode_chain(const hx_chain_args* args, hx_stream stream);

/* Debug / tooling (not on the product path): read-streaming microbenchmark used by
 * tools/bench_stream.py to choose load shapes.  variant 0: contiguous 1 KiB per wave instruction;
 * 1..4: 8x128 B, 4x256 B, 2x512 B, 1x1024 B (rows x bytes per instruction) of a row-major matrix
 * with row pitch `pitch` bytes.  unroll = loads in flight per wave (4, 8, 16, 32); policy 1 =
 * non-temporal loads.  Reads `bytes` bytes once. */
int hx_debug_stream_read(const void* p, int64_t bytes, int variant, int64_t pitch, int unroll,
                         int policy, int wgs, float* sink, hx_stream stream);
/* measurement aid: the decode attention kernel's read pattern (paged, one head's 256 B of every key row) with
 * no arithmetic — the ceiling that kernel can reach (tools/bench_attn_ceiling.py) */
int hx_debug_paged_read(const void* kbase, const void* vbase, const int32_t* table, int n_seq, int n_heads,
                        int tiles, int64_t page_bytes, int row_bytes, int heads_per_wg, int waves, int depth,
                        int n_splits, float* sink, hx_stream stream);

#ifdef __cplusplus
}
#endif
#endif /* HYDRA_HIP_EXPERIMENTAL_H */
