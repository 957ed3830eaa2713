// attn_common.h — MFMA wrappers and parameter block shared by the attention kernels.
#pragma once
#include "hx_common.h"

namespace hx {

template <typename T> struct Mfma;
template <> struct Mfma<F16> {
  // D[16x16] += A[16x32] * B[32x16]; lane l: A[row l&15][k 8(l>>4)+j], B[k 8(l>>4)+j][col l&15]
  static __device__ __forceinline__ f32x4 mma(u16x8 a, u16x8 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a),
                                                  __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
  }
};
template <> struct Mfma<BF16> {
  static __device__ __forceinline__ f32x4 mma(u16x8 a, u16x8 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a),
                                                   __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
  }
};

__device__ __forceinline__ float fast_exp2(float x) { return __builtin_amdgcn_exp2f(x); }

// ds_read_b64_tr_b16: lane (g = l>>4, c = l&15) of a [16 rows][16 cols] 2-byte tile whose rows
// start at lds_ptr(row-group base) receives column-major data suitable as an MFMA A operand of
// the transposed tile (see attn_fwd.hip for the addressing used with it).
typedef short hx_s16x4_t __attribute__((__vector_size__(4 * sizeof(short))));
__device__ __forceinline__ u16x4 lds_tr_read(const char* lds_ptr) {
  typedef __attribute__((address_space(3))) hx_s16x4_t lds_s4;
  hx_s16x4_t r = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4*)lds_ptr);
  return __builtin_bit_cast(u16x4, r);
}

// Finite "minus infinity" for running maxima: keeps exp2(m_old - m_new) == 1 when a
// lane group has not seen any unmasked key yet (no NaN from inf - inf).
#define HX_NEG_BIG (-1.0e30f)

// key index -> (page slot in the block table, row inside the page)
__device__ __forceinline__ int page_slot(int key, int block_size, int block_shift) {
  return block_shift >= 0 ? key >> block_shift : key / block_size;
}
__device__ __forceinline__ int page_row(int key, int block_size, int block_shift) {
  return block_shift >= 0 ? key & (block_size - 1) : key % block_size;
}

struct AttnParams {
  void* out;
  const void* q;
  const void* k;
  const void* v;
  const int32_t* cu_q;
  const int32_t* cu_k;
  const int32_t* block_table;
  const int32_t* cu_block_lens;
  int64_t q_row_stride, o_row_stride;
  int64_t k_block_stride, k_row_stride, k_head_stride;
  int64_t v_block_stride, v_row_stride, v_head_stride;
  int32_t n_heads, group;   // group = n_heads / n_kv_heads
  int32_t batch;            // number of sequences
  int64_t total_q;          // query rows over all sequences
  int32_t block_size;       // paged: tokens per page (multiple of 16)
  int32_t block_shift;      // log2(block_size) when it is a power of two, else -1 (runtime integer
                            // division costs ~30 VALU instructions; the kernels do two per key row)
  int32_t causal;
  int32_t window_left, window_right;   // local attention (general kernel only): -1 = unbounded
  float softcap_scale;                 // softmax_scale / softcap when softcap > 0 (then scale_log2 = softcap * log2 e), else 0
  int32_t xcd_remap;        // prefill kernel: renumber workgroups so that one head's query tiles share an XCD
  int32_t wg_priority;      // prefill kernel (32x32 form): the two workgroups of a CU run at different priorities
  unsigned long long* stamps;   // EXPERIMENTS builds: time stamps of the persistent prefill kernel (null otherwise)
  int32_t max_seqlen_k;     // the caller's bound on a sequence's keys (host side: chooses between the forms of the prefill kernel)
  int32_t cu_pairing;       // prefill kernel, persistent form: the second workgroup of a CU takes a round's items from the short end
  int32_t unit_mode;        // prefill kernel, persistent form: deal units (the k-th longest + k-th shortest tile of a sequence)
  int32_t seq_group;        // prefill kernel, persistent form: sequences per deal group, 1 / 2 / 4 (set by its launcher)
  int32_t n_cus;            // prefill kernel, persistent form: CUs of the device (set by its launcher)
  int32_t n_tile_slots;     // prefill kernel, persistent form: upper bound of the (sequence, query tile) pairs (set by its launcher)
  float scale_log2;         // softmax_scale * log2(e)
  int32_t n_splits;
  float* ws_o;              // [batch, n_heads, n_splits, D] unnormalised partial outputs
  float* ws_ml;             // [batch, n_heads, n_splits, 2] (max, sum)
  // fused RoPE + cache append (decode only; all null/0 otherwise)
  const void* k_new;        // [batch, n_kv_heads, D] un-rotated key of the new token
  const void* v_new;        // [batch, n_kv_heads, D]
  int64_t kn_row_stride, vn_row_stride;
  const int32_t* positions; // [batch] rotary position of the new token
  const void* cos_sin;      // [max_pos, 2, D/2] in T
  const int32_t* new_slots; // [batch] cache slot of the new token
  // optional: q / k_new / v_new arrive as split-K fp32 slabs of the fused qkv projection
  const float* qkv_partial; // [n_splits][batch][(n_heads + 2*n_kv_heads) * D]
  int32_t qkv_splits;
  int64_t qkv_slab_stride;  // batch * row length
  int64_t qkv_row;          // (n_heads + 2*n_kv_heads) * D
  const int32_t* rank_desc; // decode kernel, RANKED form: [0] ragged?, [1 + r] the sequence with the r-th most keys (null: static grid)
};

}  // namespace hx
