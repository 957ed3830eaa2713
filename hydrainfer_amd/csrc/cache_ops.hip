// cache_ops.hip — paged-cache scatter kernels (set_kv_cache / set_image_cache) and the
// decode-step metadata advance.  Pure byte movement: HBM-bound, 16 B per lane,
// one launch per call.
//
// Behaviour follows csrc/kernel/kv_cache_kernels/kv_cache_kernels.cu:16-95 and
// csrc/kernel/cache_kernels/cache_kernels.cu:16-83 of the reference (slot -> block,
// offset addressing; bit-exact element copy), re-designed as a row-vector copy:
// a token row (n_heads*head_dim elements) is contiguous in both source and cache,
// so the kernel moves it as 16-byte vectors instead of per-element index math.
#include "hx_common.h"

namespace {

using namespace hx;

// One thread moves one 16-byte vector of K and (if NCACHE==2) one of V.
// grid.x = ceil(vecs_per_row / 256), grid.y = n_tokens.
template <int NCACHE>
__global__ __launch_bounds__(256) void scatter_rows_vec16(
    const int32_t* __restrict__ slot_ids, const uint4* __restrict__ src0,
    const uint4* __restrict__ src1, uint4* __restrict__ dst0, uint4* __restrict__ dst1,
    int64_t src0_stride_v, int64_t src1_stride_v, int64_t dst0_block_stride_v,
    int64_t dst1_block_stride_v, int32_t vecs_per_row, int32_t block_size) {
  const int token = blockIdx.y;
  const int v = blockIdx.x * 256 + threadIdx.x;
  const int slot = slot_ids[token];
  if (slot < 0 || v >= vecs_per_row) return;
  const int64_t blk = slot / block_size;
  const int64_t off = slot % block_size;
  const int64_t row_in_block = off * (int64_t)vecs_per_row + v;
  uint4 a = src0[token * src0_stride_v + v];
  uint4 b;
  if (NCACHE == 2) b = src1[token * src1_stride_v + v];
  dst0[blk * dst0_block_stride_v + row_in_block] = a;
  if (NCACHE == 2) dst1[blk * dst1_block_stride_v + row_in_block] = b;
}

// Element-granular fallback for rows whose byte size or base alignment is not a
// multiple of 16 (the reference test grid has none, kept for contract completeness).
template <typename E, int NCACHE>
__global__ __launch_bounds__(256) void scatter_rows_elem(
    const int32_t* __restrict__ slot_ids, const E* __restrict__ src0, const E* __restrict__ src1,
    E* __restrict__ dst0, E* __restrict__ dst1, int64_t src0_stride, int64_t src1_stride,
    int64_t dst0_block_stride, int64_t dst1_block_stride, int32_t row_elems,
    int32_t block_size) {
  const int token = blockIdx.y;
  const int slot = slot_ids[token];
  if (slot < 0) return;
  const int64_t blk = slot / block_size;
  const int64_t off = slot % block_size;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < row_elems; i += gridDim.x * 256) {
    const int64_t r = off * (int64_t)row_elems + i;
    dst0[blk * dst0_block_stride + r] = src0[token * src0_stride + i];
    if (NCACHE == 2) dst1[blk * dst1_block_stride + r] = src1[token * src1_stride + i];
  }
}

template <int NCACHE>
int launch_scatter(const int32_t* slot_ids, const void* s0, const void* s1, void* d0, void* d1,
                   int64_t n_tokens, int64_t row_elems, int64_t block_size, int64_t s0_stride,
                   int64_t s1_stride, int64_t d0_bstride, int64_t d1_bstride, int dtype,
                   hipStream_t stream) {
  const int64_t es = dtype_size(dtype);
  if (es == 0) return HX_ERR_DTYPE;
  if (n_tokens == 0 || row_elems == 0) return HX_OK;
  if (n_tokens > 65535 * 1024LL) return HX_ERR_SHAPE;
  const int64_t row_bytes = row_elems * es;
  const bool vec_ok = (row_bytes % 16 == 0) && (s0_stride * es % 16 == 0) &&
                      (d0_bstride * es % 16 == 0) && aligned16(s0) && aligned16(d0) &&
                      (NCACHE == 1 || ((s1_stride * es % 16 == 0) && (d1_bstride * es % 16 == 0) &&
                                       aligned16(s1) && aligned16(d1)));
  // tokens go on grid.y (max 65535): fold larger counts by looping launches
  for (int64_t t0 = 0; t0 < n_tokens; t0 += 65535) {
    const int64_t nt = (n_tokens - t0 < 65535) ? (n_tokens - t0) : 65535;
    const int32_t* sl = slot_ids + t0;
    if (vec_ok) {
      const int vpr = (int)(row_bytes / 16);
      dim3 grid((vpr + 255) / 256, (unsigned)nt);
      const int64_t epv = 16 / es;
      hx::launcher(scatter_rows_vec16<NCACHE>, grid, 256, 0, stream)(
          sl, (const uint4*)((const char*)s0 + t0 * s0_stride * es),
          NCACHE == 2 ? (const uint4*)((const char*)s1 + t0 * s1_stride * es) : nullptr,
          (uint4*)d0, (uint4*)d1, s0_stride / epv, s1_stride / epv, d0_bstride / epv,
          d1_bstride / epv, vpr, (int)block_size);
    } else {
      dim3 grid((unsigned)((row_elems + 255) / 256 > 64 ? 64 : (row_elems + 255) / 256),
                (unsigned)nt);
      if (es == 2) {
        hx::launcher(scatter_rows_elem<uint16_t, NCACHE>, grid, 256, 0, stream)(
            sl, (const uint16_t*)s0 + t0 * s0_stride,
            NCACHE == 2 ? (const uint16_t*)s1 + t0 * s1_stride : nullptr, (uint16_t*)d0,
            (uint16_t*)d1, s0_stride, s1_stride, d0_bstride, d1_bstride, (int)row_elems,
            (int)block_size);
      } else {
        hx::launcher(scatter_rows_elem<uint32_t, NCACHE>, grid, 256, 0, stream)(
            sl, (const uint32_t*)s0 + t0 * s0_stride,
            NCACHE == 2 ? (const uint32_t*)s1 + t0 * s1_stride : nullptr, (uint32_t*)d0,
            (uint32_t*)d1, s0_stride, s1_stride, d0_bstride, d1_bstride, (int)row_elems,
            (int)block_size);
      }
    }
    int rc = check_launch();
    if (rc) return rc;
  }
  return HX_OK;
}

// One wave per launch is plenty: batch <= a few hundred sequences.
__global__ __launch_bounds__(256) void decode_advance_kernel(
    int32_t* __restrict__ positions, int32_t* __restrict__ kv_lens,
    int32_t* __restrict__ cu_seqlens_k, int32_t* __restrict__ new_cache_slots,
    const int32_t* __restrict__ block_table, const int32_t* __restrict__ cu_block_lens,
    int32_t batch, int32_t block_size, int32_t stride, int32_t* __restrict__ rank_desc) {
  __shared__ int32_t scan[256];
  hx::decode_advance_block(positions, kv_lens, cu_seqlens_k, new_cache_slots, block_table, cu_block_lens, batch,
                           block_size, stride, scan);
  if (rank_desc) hx::decode_rank_block(kv_lens, batch, rank_desc, scan);
}

__global__ __launch_bounds__(256) void decode_rank_kernel(const int32_t* __restrict__ cu_seqlens_k, int32_t batch,
                                                          int32_t* __restrict__ rank_desc) {
  __shared__ int32_t lens[256], scratch[256];
  if ((int)threadIdx.x < batch && batch <= 256) lens[threadIdx.x] = cu_seqlens_k[threadIdx.x + 1] - cu_seqlens_k[threadIdx.x];
  __syncthreads();
  hx::decode_rank_block(lens, batch, rank_desc, scratch);
}

}  // namespace

extern "C" int hx_set_kv_cache(const int32_t* slot_ids, const void* keys, const void* values,
                               void* key_cache, void* value_cache, int64_t n_tokens,
                               int64_t n_kv_heads, int64_t head_dim, int64_t block_size,
                               int64_t k_stride, int64_t v_stride, int64_t kcache_block_stride,
                               int64_t vcache_block_stride, int dtype, hx_stream stream) {
  if (!slot_ids || !keys || !values || !key_cache || !value_cache) {
    return n_tokens == 0 ? HX_OK : HX_ERR_NULL;
  }
  if (n_tokens < 0 || n_kv_heads <= 0 || head_dim <= 0 || block_size <= 0) return HX_ERR_SHAPE;
  const int64_t row = n_kv_heads * head_dim;
  if (k_stride < row || v_stride < row) return HX_ERR_STRIDE;
  if (kcache_block_stride < block_size * row || vcache_block_stride < block_size * row)
    return HX_ERR_STRIDE;
  return launch_scatter<2>(slot_ids, keys, values, key_cache, value_cache, n_tokens, row,
                           block_size, k_stride, v_stride, kcache_block_stride,
                           vcache_block_stride, dtype, (hipStream_t)stream);
}

extern "C" int hx_set_image_cache(const int32_t* slot_ids, const void* image_tokens,
                                  void* image_cache, int64_t n_tokens, int64_t n_heads,
                                  int64_t head_dim, int64_t block_size, int64_t token_stride,
                                  int64_t cache_block_stride, int dtype, hx_stream stream) {
  if (!slot_ids || !image_tokens || !image_cache) return n_tokens == 0 ? HX_OK : HX_ERR_NULL;
  if (n_tokens < 0 || n_heads <= 0 || head_dim <= 0 || block_size <= 0) return HX_ERR_SHAPE;
  const int64_t row = n_heads * head_dim;
  if (token_stride < row) return HX_ERR_STRIDE;
  if (cache_block_stride < block_size * row) return HX_ERR_STRIDE;
  return launch_scatter<1>(slot_ids, image_tokens, nullptr, image_cache, nullptr, n_tokens, row,
                           block_size, token_stride, 0, cache_block_stride, 0, dtype,
                           (hipStream_t)stream);
}

extern "C" int hx_decode_advance(int32_t* positions, int32_t* kv_lens, int32_t* cu_seqlens_k,
                                 int32_t* new_cache_slots, const int32_t* block_table,
                                 const int32_t* cu_block_lens, int32_t batch, int32_t block_size,
                                 int32_t stride, hx_stream stream) {
  if (!positions || !kv_lens || !cu_seqlens_k || !new_cache_slots || !block_table ||
      !cu_block_lens)
    return HX_ERR_NULL;
  if (batch <= 0 || block_size <= 0 || stride < 1) return HX_ERR_SHAPE;
  hx::launcher(decode_advance_kernel, 1, 256, 0, (hipStream_t)stream)(positions, kv_lens, cu_seqlens_k,
                                                            new_cache_slots, block_table,
                                                            cu_block_lens, batch, block_size, stride, (int32_t*)nullptr);
  return hx::check_launch();
}

extern "C" int hx_decode_advance_ranked(int32_t* positions, int32_t* kv_lens, int32_t* cu_seqlens_k,
                                        int32_t* new_cache_slots, const int32_t* block_table,
                                        const int32_t* cu_block_lens, int32_t batch, int32_t block_size,
                                        int32_t stride, int32_t* rank_desc, hx_stream stream) {
  if (!positions || !kv_lens || !cu_seqlens_k || !new_cache_slots || !block_table || !cu_block_lens || !rank_desc)
    return HX_ERR_NULL;
  if (batch <= 0 || block_size <= 0 || stride < 1) return HX_ERR_SHAPE;
  hx::launcher(decode_advance_kernel, 1, 256, 0, (hipStream_t)stream)(positions, kv_lens, cu_seqlens_k, new_cache_slots,
                                                                      block_table, cu_block_lens, batch, block_size, stride,
                                                                      rank_desc);
  return hx::check_launch();
}

extern "C" int hx_decode_rank(const int32_t* cu_seqlens_k, int32_t batch, int32_t* rank_desc, hx_stream stream) {
  if (!cu_seqlens_k || !rank_desc) return HX_ERR_NULL;
  if (batch <= 0) return HX_ERR_SHAPE;
  hx::launcher(decode_rank_kernel, 1, 256, 0, (hipStream_t)stream)(cu_seqlens_k, batch, rank_desc);
  return hx::check_launch();
}

namespace {
__global__ __launch_bounds__(256) void decode_feed_ids_kernel(int64_t* __restrict__ out, const int32_t* __restrict__ ids,
                                                              const int32_t* __restrict__ src,
                                                              const int64_t* __restrict__ prev, int32_t n) {
  const int r = blockIdx.x * 256 + threadIdx.x;
  if (r < n) {
    const int32_t s = src[r];
    out[r] = s >= 0 ? prev[s] : (int64_t)ids[r];
  }
}

__global__ __launch_bounds__(64) void collect_errors_kernel(uint32_t* __restrict__ out, const uint32_t* __restrict__ areas,
                                                            int32_t n_areas, int64_t stride_words, int32_t word,
                                                            const uint32_t* __restrict__ extra) {
  uint32_t v = 0;
  for (int i = threadIdx.x; i < n_areas; i += 64) v |= areas[(int64_t)i * stride_words + word];
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v |= __shfl_xor(v, off, 64);
  if (threadIdx.x == 0) out[0] = v | (extra ? extra[0] : 0u);
}
// two small word arrays moved by ONE launch; either side of either pair may be host-mapped (pinned) memory
__global__ __launch_bounds__(256) void copy_words2_kernel(uint32_t* __restrict__ dst0, const uint32_t* __restrict__ src0, int32_t n0,
                                                          uint32_t* __restrict__ dst1, const uint32_t* __restrict__ src1, int32_t n1) {
  const int stride = gridDim.x * 256;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < n0; i += stride) dst0[i] = src0[i];
  for (int i = blockIdx.x * 256 + threadIdx.x; i < n1; i += stride) dst1[i] = src1[i];
}
}  // namespace

namespace {
// staging: [head words] [n_runs] then n_runs x ([dst word offset] [count] [count values]): the head is copied to dst[0 ..),
// every run to dst[offset ..).  One workgroup: the runs are few and short (a block id per sequence that crossed a block
// boundary, a whole table for a sequence that has just joined the batch).
__global__ __launch_bounds__(256) void stage_decode_kernel(uint32_t* __restrict__ dst, const uint32_t* __restrict__ staging,
                                                           int32_t head_words, int64_t dst_words) {
  for (int i = threadIdx.x; i < head_words; i += 256) dst[i] = staging[i];
  const uint32_t n_runs = staging[head_words];
  int64_t at = (int64_t)head_words + 1;
  for (uint32_t r = 0; r < n_runs; ++r) {
    const int64_t off = staging[at];
    const uint32_t cnt = staging[at + 1];
    if (off + cnt <= dst_words)
      for (uint32_t i = threadIdx.x; i < cnt; i += 256) dst[off + i] = staging[at + 2 + i];
    at += 2 + cnt;
  }
}
}  // namespace

extern "C" int hx_stage_decode(void* dst, int64_t dst_words, const void* staging, int32_t head_words, hx_stream stream) {
  if (!dst || !staging) return HX_ERR_NULL;
  if (head_words < 0 || dst_words < head_words) return HX_ERR_SHAPE;
  if ((((uintptr_t)dst | (uintptr_t)staging) & 3) != 0) return HX_ERR_STRIDE;
  hx::launcher(stage_decode_kernel, 1, 256, 0, (hipStream_t)stream)((uint32_t*)dst, (const uint32_t*)staging, head_words, dst_words);
  return hx::check_launch();
}

extern "C" int hx_copy_words2(void* dst0, const void* src0, int32_t n0_words, void* dst1, const void* src1, int32_t n1_words,
                              hx_stream stream) {
  if (n0_words < 0 || n1_words < 0) return HX_ERR_SHAPE;
  if ((n0_words > 0 && (!dst0 || !src0)) || (n1_words > 0 && (!dst1 || !src1))) return HX_ERR_NULL;
  if (n0_words + n1_words == 0) return HX_OK;
  if ((((uintptr_t)dst0 | (uintptr_t)src0 | (uintptr_t)dst1 | (uintptr_t)src1) & 3) != 0) return HX_ERR_STRIDE;
  const int n = n0_words > n1_words ? n0_words : n1_words;
  int blocks = (n + 255) / 256;
  if (blocks > 64) blocks = 64;
  hx::launcher(copy_words2_kernel, (unsigned)blocks, 256, 0, (hipStream_t)stream)(
      (uint32_t*)dst0, (const uint32_t*)src0, n0_words, (uint32_t*)dst1, (const uint32_t*)src1, n1_words);
  return hx::check_launch();
}

extern "C" int hx_decode_feed_ids(int64_t* out, const int32_t* ids, const int32_t* src, const int64_t* prev,
                                  int32_t n, hx_stream stream) {
  if (!out || !ids || !src || !prev) return HX_ERR_NULL;
  if (n <= 0) return HX_ERR_SHAPE;
  hx::launcher(decode_feed_ids_kernel, (unsigned)((n + 255) / 256), 256, 0, (hipStream_t)stream)(out, ids, src, prev, n);
  return hx::check_launch();
}

extern "C" int hx_collect_errors(uint32_t* out, const uint32_t* areas, int32_t n_areas, int64_t stride_words,
                                 int32_t word, const uint32_t* extra, hx_stream stream) {
  if (!out || (n_areas > 0 && !areas)) return HX_ERR_NULL;
  if (n_areas < 0 || stride_words < 0 || word < 0) return HX_ERR_SHAPE;
  hx::launcher(collect_errors_kernel, 1, 64, 0, (hipStream_t)stream)(out, areas, n_areas, stride_words, word, extra);
  return hx::check_launch();
}
