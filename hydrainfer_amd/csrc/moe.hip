// moe.hip — MoE routing / permutation ops (csrc/kernel/moe/* of the reference; surface
// hydrainfer/_C/kernel/moe/__init__.pyi).  Not on the LLaVA path (no production caller in
// the reference); built because north_star names them.  Integer results (indices, row maps)
// follow the reference's tie-breaking exactly:
//   topk_softmax          : softmax then iterative arg-max, lower index wins ties, chosen
//                           entry cleared to -10000 (topk_softmax_kernel.cu:108-180)
//   grouped_topk_sigmoid  : one lane per expert group, drop (n_groups - topk_group) groups by
//                           min(top1+top2 of sigmoid+bias) with ties -> higher group, then
//                           top-k by sigmoid+bias, ties -> lower expert; weights = raw sigmoid
//                           (scaling_factor not applied, grouped_topk_sigmoid_kernel.cu:180)
//   index / mask row maps : stable expert-major order (cub radix sort -> rocPRIM here)
// wave64: one wavefront per token for the routing kernels.
#include <cstring>
#include <rocprim/device/device_radix_sort.hpp>
#include <cfloat>
#include "hx_common.h"

namespace {

using namespace hx;

constexpr int kMaxExperts = 1024;

// ---------------------------------------------------------------------------------------
// topk_softmax: 4 tokens per 256-thread workgroup, one wave per token; lane l owns experts
// l, l+64, ...
// ---------------------------------------------------------------------------------------
// (kPerLane = experts per lane, 64 kPerLane >= n_experts: with 16 slots per lane for every expert count, 256 experts —
// 4 per lane — spent three quarters of the kernel on predicated-off slots; 4096 tokens x 256 experts, top 8: 14.5 us)
template <int kPerLane>
__global__ __launch_bounds__(256) void topk_softmax_kernel(const float* __restrict__ logits,
                                                           float* __restrict__ weights,
                                                           int32_t* __restrict__ indices,
                                                           int64_t n_tokens, int n_experts,
                                                           int topk) {
  const int lane = threadIdx.x & 63;
  const int64_t token = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (token >= n_tokens) return;
  const float* row = logits + token * n_experts;
  float v[kPerLane];
  float mx = -FLT_MAX;
#pragma unroll
  for (int i = 0; i < kPerLane; ++i) {
    const int e = lane + 64 * i;
    v[i] = (e < n_experts) ? row[e] : -FLT_MAX;
    mx = fmaxf(mx, v[i]);
  }
  mx = wave_max(mx);
  float sum = 0.f;
#pragma unroll
  for (int i = 0; i < kPerLane; ++i) {
    const int e = lane + 64 * i;
    v[i] = (e < n_experts) ? expf(v[i] - mx) : 0.f;
    sum += v[i];
  }
  sum = wave_sum(sum);
  const float inv = 1.f / sum;
#pragma unroll
  for (int i = 0; i < kPerLane; ++i) {
    const int e = lane + 64 * i;
    v[i] = (e < n_experts) ? v[i] * inv : -FLT_MAX;
  }
  for (int k = 0; k < topk; ++k) {
    float best = -FLT_MAX;
    int col = 0x7fffffff;
#pragma unroll
    for (int i = 0; i < kPerLane; ++i) {
      const int e = lane + 64 * i;
      if (e < n_experts && (v[i] > best)) {  // ascending e within a lane: first max kept
        best = v[i];
        col = e;
      }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
      const float ob = __shfl_xor(best, off, 64);
      const int oc = __shfl_xor(col, off, 64);
      if (ob > best || (ob == best && oc < col)) {
        best = ob;
        col = oc;
      }
    }
    if (lane == 0) {
      weights[token * topk + k] = best;
      indices[token * topk + k] = col;
    }
    if ((col & 63) == lane) {
#pragma unroll
      for (int i = 0; i < kPerLane; ++i)
        if (lane + 64 * i == col) v[i] = -10000.f;
    }
  }
}

// ---------------------------------------------------------------------------------------
// grouped_topk_sigmoid: one wave per token, lane g < n_groups owns group g (contiguous
// experts), scores staged in LDS.
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void grouped_topk_sigmoid_kernel(
    const float* __restrict__ logits, const float* __restrict__ bias, float* __restrict__ weights,
    int32_t* __restrict__ indices, int64_t n_tokens, int n_experts, int n_groups, int topk_group,
    int topk) {
  extern __shared__ float smem[];  // [4 tokens][2][n_experts]
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int64_t token = (int64_t)blockIdx.x * 4 + w;
  if (token >= n_tokens) return;
  float* score = smem + (size_t)w * 2 * n_experts;
  float* choice = score + n_experts;
  for (int e = lane; e < n_experts; e += 64) {
    const float s = 1.0f / (1.f + expf(-logits[token * n_experts + e]));
    score[e] = s;
    choice[e] = s + bias[e];
  }
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  const int per = n_experts / n_groups;
  const int start = lane * per;
  const bool active = lane < n_groups;

  for (int it = 0; it < n_groups - topk_group; ++it) {
    float m1 = -FLT_MAX, m2 = -FLT_MAX;
    if (active)
      for (int i = 0; i < per; ++i) {
        const float val = choice[start + i];
        if (val > m1) { m2 = m1; m1 = val; }
        else if (val > m2) m2 = val;
      }
    float msum = active ? m1 + m2 : INFINITY;
    int mcol = active ? start : -1;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
      const float os = __shfl_xor(msum, off, 64);
      const int oc = __shfl_xor(mcol, off, 64);
      if (os < msum || (os == msum && oc > mcol)) { msum = os; mcol = oc; }
    }
    if (active && mcol / per == lane)
      for (int i = 0; i < per; ++i) choice[start + i] = FLT_MAX;
    __builtin_amdgcn_wave_barrier();
  }

  for (int k = 0; k < topk; ++k) {
    float best = -FLT_MAX;
    int col = 0x7fffffff;
    if (active) {
      best = choice[start];
      col = start;
      if (best != FLT_MAX) {
        for (int i = 1; i < per; ++i) {
          const float val = choice[start + i];
          if (val > best) { best = val; col = start + i; }
        }
      } else {
        best = -FLT_MAX;
      }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
      const float ob = __shfl_xor(best, off, 64);
      const int oc = __shfl_xor(col, off, 64);
      if (ob > best || (ob == best && oc < col)) { best = ob; col = oc; }
    }
    if (active && col / per == lane) {
      choice[col] = -FLT_MAX;
      weights[token * topk + k] = score[col];
      indices[token * topk + k] = col;
    }
    __builtin_amdgcn_wave_barrier();
  }
}

// ---------------------------------------------------------------------------------------
// row maps
// ---------------------------------------------------------------------------------------
__global__ void iota_kernel(int32_t* p, int64_t n) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) p[i] = (int32_t)i;
}

// sorted_row_id[p] = flat index f = t*topk + k  ->  row_id_map[k*n_tokens + t] = p
__global__ void row_id_map_from_sorted_kernel(const int32_t* __restrict__ sorted_row_id,
                                              int32_t* __restrict__ row_id_map, int64_t n_tokens,
                                              int topk) {
  const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= n_tokens * topk) return;
  const int f = sorted_row_id[p];
  row_id_map[(int64_t)(f % topk) * n_tokens + f / topk] = (int32_t)p;
}

// mask map, pass 1: counts[e] = number of tokens routed to expert e
__global__ __launch_bounds__(256) void mask_count_kernel(const uint8_t* __restrict__ routing_map,
                                                         int32_t* __restrict__ counts,
                                                         int64_t n_tokens, int n_experts) {
  __shared__ int red[4];
  const int e = blockIdx.x;
  int c = 0;
  for (int64_t t = threadIdx.x; t < n_tokens; t += 256) c += routing_map[t * n_experts + e] ? 1 : 0;
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) c += __shfl_xor(c, off, 64);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = c;
  __syncthreads();
  if (threadIdx.x == 0) counts[e] = red[0] + red[1] + red[2] + red[3];
}

// pass 2: expert e writes offset(e) + rank for its routed tokens, -1 elsewhere
__global__ __launch_bounds__(256) void mask_assign_kernel(const uint8_t* __restrict__ routing_map,
                                                          const int32_t* __restrict__ counts,
                                                          int32_t* __restrict__ row_id_map,
                                                          int64_t n_tokens, int n_experts) {
  __shared__ int wave_tot[4];
  __shared__ int base_s;
  const int e = blockIdx.x;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  if (threadIdx.x == 0) {
    int b = 0;
    for (int i = 0; i < e; ++i) b += counts[i];
    base_s = b;
  }
  __syncthreads();
  int base = base_s;
  for (int64_t t0 = 0; t0 < n_tokens; t0 += 256) {
    const int64_t t = t0 + threadIdx.x;
    const int flag = (t < n_tokens && routing_map[t * n_experts + e]) ? 1 : 0;
    const unsigned long long bal = __ballot(flag);
    const int before = __popcll(bal & ((1ull << lane) - 1ull));
    if (lane == 0) wave_tot[w] = __popcll(bal);
    __syncthreads();
    int woff = 0;
    for (int i = 0; i < w; ++i) woff += wave_tot[i];
    const int tot = wave_tot[0] + wave_tot[1] + wave_tot[2] + wave_tot[3];
    if (t < n_tokens) row_id_map[(int64_t)e * n_tokens + t] = flag ? base + woff + before : -1;
    base += tot;
    __syncthreads();
  }
}

// ---------------------------------------------------------------------------------------
// permute / unpermute (shared by the index map [topk, n_tokens] and the mask map
// [n_experts, n_tokens]; entries < 0 are skipped)
// ---------------------------------------------------------------------------------------
template <typename E>
__global__ __launch_bounds__(256) void permute_kernel(const E* __restrict__ tokens,
                                                      E* __restrict__ permuted,
                                                      const int32_t* __restrict__ row_id_map,
                                                      int64_t n_tokens, int n_rows, int64_t n_elem) {
  const int64_t t = blockIdx.x;
  for (int64_t i = threadIdx.x; i < n_elem; i += 256) {
    const E val = tokens[t * n_elem + i];
    for (int k = 0; k < n_rows; ++k) {
      const int p = row_id_map[(int64_t)k * n_tokens + t];
      if (p >= 0) permuted[(int64_t)p * n_elem + i] = val;
    }
  }
}

// accumulate in T arithmetic like the reference (frag_sum is T; product and sum each rounded)
template <typename T>
__global__ __launch_bounds__(256) void unpermute_kernel(
    const typename T::storage* __restrict__ permuted, typename T::storage* __restrict__ out,
    const int32_t* __restrict__ row_id_map, const typename T::storage* __restrict__ probs,
    int64_t n_tokens, int n_rows, int64_t dim) {
  const int64_t t = blockIdx.x;
  for (int64_t i = threadIdx.x; i < dim; i += 256) {
    float acc = 0.f;
    for (int k = 0; k < n_rows; ++k) {
      const int p = row_id_map[(int64_t)k * n_tokens + t];
      if (p < 0) continue;
      const float pr = probs ? T::to_float(probs[t * n_rows + k]) : 1.0f;
      const float prod = round_to<T>(T::to_float(permuted[(int64_t)p * dim + i]) * pr);
      acc = round_to<T>(acc + prod);
    }
    out[t * dim + i] = T::from_float(acc);
  }
}

// sum_out: the reference's topk_sum_kernel (align_block_kernel.cu:172-188) keeps its running sum in scalar_t — every partial
// sum is rounded to T — for topk in {2, 3, 4, 8}; any other topk goes to torch::sum_out (fp32 accumulation, one rounding;
// align_block_kernel.cu:262-271).  t_accum selects the former.
template <typename T>
__global__ __launch_bounds__(256) void sum_out_kernel(const typename T::storage* __restrict__ in,
                                                      typename T::storage* __restrict__ out,
                                                      int topk, int64_t dim, int t_accum) {
  const int64_t t = blockIdx.x;
  for (int64_t i = threadIdx.x; i < dim; i += 256) {
    float acc = 0.f;
    for (int k = 0; k < topk; ++k) {
      acc += T::to_float(in[(t * topk + k) * dim + i]);
      if (t_accum) acc = round_to<T>(acc);
    }
    out[t * dim + i] = T::from_float(acc);
  }
}

// The two-byte forms as streaming kernels: 16 bytes per lane, the token's valid rows gathered ONCE (an ordered compaction of
// its column of the map into LDS: the sum is taken in map order, each product and each partial sum rounded to T like the
// scalar kernel above and the reference), all of a chunk's rows requested before the first is used, and the row cut
// into chunks over blockIdx.y so that few tokens still fill the chip.  Scalar form (2-byte loads, the map re-read for
// every element): 4096 tokens x 7168, top 8: 450 us = 1.2 TB/s; 32 tokens: 54 us.
template <typename T>
__global__ __launch_bounds__(256) void unpermute_vec_kernel(
    const u16* __restrict__ permuted, u16* __restrict__ out, const int32_t* __restrict__ row_id_map,
    const u16* __restrict__ probs, int64_t n_tokens, int n_rows, int64_t dim) {
  __shared__ int s_row[256];
  __shared__ float s_prob[256];
  __shared__ int s_cnt[4];
  const int64_t t = blockIdx.x;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  {
    const int k = threadIdx.x;
    const int prow = k < n_rows ? row_id_map[(int64_t)k * n_tokens + t] : -1;
    const uint64_t mask = __ballot(prow >= 0);
    if (lane == 0) s_cnt[w] = __builtin_popcountll(mask);
    __syncthreads();
    int pos = __builtin_popcountll(mask & ((1ull << lane) - 1));
    for (int i = 0; i < w; ++i) pos += s_cnt[i];
    if (prow >= 0) {
      s_row[pos] = prow;
      s_prob[pos] = probs ? T::to_float(probs[t * n_rows + k]) : 1.0f;
    }
    __syncthreads();
  }
  const int n = s_cnt[0] + s_cnt[1] + s_cnt[2] + s_cnt[3];
  const int64_t n_chunks = dim >> 3;
  for (int64_t ch = (int64_t)blockIdx.y * 256 + threadIdx.x; ch < n_chunks; ch += (int64_t)gridDim.y * 256) {
    float acc[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) acc[e] = 0.f;
    int j = 0;
    for (; j + 4 <= n; j += 4) {
      u16x8 v[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) v[u] = __builtin_nontemporal_load(reinterpret_cast<const u16x8*>(permuted + (int64_t)s_row[j + u] * dim) + ch);
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const float pr = s_prob[j + u];
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[e] = round_to<T>(acc[e] + round_to<T>(T::to_float(v[u][e]) * pr));
      }
    }
    for (; j < n; ++j) {
      const u16x8 v = __builtin_nontemporal_load(reinterpret_cast<const u16x8*>(permuted + (int64_t)s_row[j] * dim) + ch);
      const float pr = s_prob[j];
#pragma unroll
      for (int e = 0; e < 8; ++e) acc[e] = round_to<T>(acc[e] + round_to<T>(T::to_float(v[e]) * pr));
    }
    u16x8 o;
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = T::from_float(acc[e]);
    reinterpret_cast<u16x8*>(out + t * dim)[ch] = o;
  }
}

template <typename T>
__global__ __launch_bounds__(256) void sum_out_vec_kernel(const u16* __restrict__ in, u16* __restrict__ out, int topk, int64_t dim,
                                                          int t_accum) {
  const int64_t t = blockIdx.x;
  const int64_t n_chunks = dim >> 3;
  for (int64_t ch = (int64_t)blockIdx.y * 256 + threadIdx.x; ch < n_chunks; ch += (int64_t)gridDim.y * 256) {
    float acc[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) acc[e] = 0.f;
    int k = 0;
    for (; k + 4 <= topk; k += 4) {
      u16x8 v[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) v[u] = __builtin_nontemporal_load(reinterpret_cast<const u16x8*>(in + (t * topk + k + u) * dim) + ch);
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          acc[e] += T::to_float(v[u][e]);
          if (t_accum) acc[e] = round_to<T>(acc[e]);      // the reference's scalar_t running sum (sum_out_kernel above)
        }
    }
    for (; k < topk; ++k) {
      const u16x8 v = __builtin_nontemporal_load(reinterpret_cast<const u16x8*>(in + (t * topk + k) * dim) + ch);
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        acc[e] += T::to_float(v[e]);
        if (t_accum) acc[e] = round_to<T>(acc[e]);
      }
    }
    u16x8 o;
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = T::from_float(acc[e]);
    reinterpret_cast<u16x8*>(out + t * dim)[ch] = o;
  }
}

// chunks of the row over blockIdx.y: enough workgroups for the chip when the tokens are few
inline unsigned row_splits(int64_t n_tokens, int64_t dim) {
  const int64_t per_row = (dim / 8 + 255) / 256;
  int64_t y = (1024 + n_tokens - 1) / n_tokens;
  if (y > per_row) y = per_row;
  return (unsigned)(y < 1 ? 1 : y);
}

inline int64_t align_up(int64_t x, int64_t a) { return (x + a - 1) / a * a; }

size_t sort_temp_bytes(int64_t n) {
  size_t bytes = 0;
  (void)rocprim::radix_sort_pairs(nullptr, bytes, (const int*)nullptr, (int*)nullptr,
                                  (const int*)nullptr, (int*)nullptr, (size_t)n, 0, 32,
                                  (hipStream_t)0);
  return bytes;
}

}  // namespace

extern "C" int hx_topk_softmax(const float* gating_logits, float* topk_weights,
                               int32_t* topk_indices, int64_t n_tokens, int64_t n_experts,
                               int64_t topk, hx_stream stream) {
  if (n_tokens < 0 || n_experts <= 0 || n_experts > kMaxExperts || topk <= 0 || topk > n_experts)
    return HX_ERR_SHAPE;
  if (n_tokens == 0) return HX_OK;
  if (!gating_logits || !topk_weights || !topk_indices) return HX_ERR_NULL;
  const unsigned grid = (unsigned)((n_tokens + 3) / 4);
#define HX_TOPK(PL)                                                                                           \
  hx::launcher(topk_softmax_kernel<PL>, grid, 256, 0, (hipStream_t)stream)(gating_logits, topk_weights, topk_indices, \
                                                                           n_tokens, (int)n_experts, (int)topk)
  if (n_experts <= 64) HX_TOPK(1);
  else if (n_experts <= 128) HX_TOPK(2);
  else if (n_experts <= 256) HX_TOPK(4);
  else if (n_experts <= 512) HX_TOPK(8);
  else HX_TOPK(16);
#undef HX_TOPK
  return check_launch();
}

extern "C" int hx_grouped_topk_sigmoid(const float* gating_logits, const float* correction_bias,
                                       float* topk_weights, int32_t* topk_indices,
                                       int64_t n_tokens, int64_t n_experts, int64_t n_groups,
                                       int64_t topk_group, int64_t topk, float scaling_factor,
                                       hx_stream stream) {
  (void)scaling_factor;  // accepted and ignored, exactly like the reference (TODO at :180 there)
  if (n_tokens < 0 || n_experts <= 0 || n_experts > kMaxExperts) return HX_ERR_SHAPE;
  if (n_groups <= 0 || n_groups > 64 || n_experts % n_groups != 0) return HX_ERR_SHAPE;
  if (topk_group <= 0 || topk_group > n_groups || topk <= 0) return HX_ERR_SHAPE;
  if (topk > topk_group * (n_experts / n_groups)) return HX_ERR_SHAPE;
  if (n_tokens == 0) return HX_OK;
  if (!gating_logits || !correction_bias || !topk_weights || !topk_indices) return HX_ERR_NULL;
  const size_t lds = (size_t)4 * 2 * n_experts * sizeof(float);
  hx::launcher(grouped_topk_sigmoid_kernel, (unsigned)((n_tokens + 3) / 4), 256, lds, (hipStream_t)stream)(
      gating_logits, correction_bias, topk_weights, topk_indices, n_tokens, (int)n_experts,
      (int)n_groups, (int)topk_group, (int)topk);
  return check_launch();
}

extern "C" int64_t hx_moe_sort_workspace_bytes(int64_t n_tokens, int64_t topk) {
  const int64_t n = n_tokens * topk;
  if (n <= 0) return 0;
  return 3 * align_up(n * 4, 256) + align_up((int64_t)sort_temp_bytes(n), 256);
}

extern "C" int hx_moe_row_id_map_from_indices(const int32_t* topk_indices, int32_t* row_id_map,
                                              int64_t n_tokens, int64_t topk, void* workspace,
                                              int64_t workspace_bytes, hx_stream stream) {
  if (n_tokens < 0 || topk <= 0) return HX_ERR_SHAPE;
  const int64_t n = n_tokens * topk;
  if (n == 0) return HX_OK;
  if (n > 0x7fffffff) return HX_ERR_SHAPE;
  if (!topk_indices || !row_id_map || !workspace) return HX_ERR_NULL;
  if (workspace_bytes < hx_moe_sort_workspace_bytes(n_tokens, topk)) return HX_ERR_WORKSPACE;
  hipStream_t s = (hipStream_t)stream;
  char* ws = (char*)workspace;
  const int64_t seg = align_up(n * 4, 256);
  int32_t* keys_out = (int32_t*)ws;
  int32_t* vals_in = (int32_t*)(ws + seg);
  int32_t* vals_out = (int32_t*)(ws + 2 * seg);
  void* temp = ws + 3 * seg;
  size_t temp_bytes = sort_temp_bytes(n);
  hx::launcher(iota_kernel, (unsigned)((n + 255) / 256), 256, 0, s)(vals_in, n);
  int rc = check_launch();
  if (rc) return rc;
  // stable LSD radix sort == cub::DeviceRadixSort::SortPairs (permutation_index_kernel.cu:39-53)
  rc = hip_rc(rocprim::radix_sort_pairs(temp, temp_bytes, topk_indices, keys_out, vals_in, vals_out,
                                        (size_t)n, 0, 32, s));
  if (rc) return rc;
  hx::launcher(row_id_map_from_sorted_kernel, (unsigned)((n + 255) / 256), 256, 0, s)(vals_out, row_id_map,
                                                                          n_tokens, (int)topk);
  return check_launch();
}

extern "C" int hx_moe_row_id_map_from_mask(const uint8_t* routing_map, int32_t* row_id_map,
                                           int64_t n_tokens, int64_t n_experts, void* workspace,
                                           int64_t workspace_bytes, hx_stream stream) {
  if (n_tokens < 0 || n_experts <= 0 || n_experts > 65535) return HX_ERR_SHAPE;
  if (n_tokens == 0) return HX_OK;
  if (!routing_map || !row_id_map || !workspace) return HX_ERR_NULL;
  if (workspace_bytes < n_experts * 4) return HX_ERR_WORKSPACE;
  hipStream_t s = (hipStream_t)stream;
  int32_t* counts = (int32_t*)workspace;
  hx::launcher(mask_count_kernel, (unsigned)n_experts, 256, 0, s)(routing_map, counts, n_tokens, (int)n_experts);
  int rc = check_launch();
  if (rc) return rc;
  hx::launcher(mask_assign_kernel, (unsigned)n_experts, 256, 0, s)(routing_map, counts, row_id_map, n_tokens,
                                                       (int)n_experts);
  return check_launch();
}

extern "C" int hx_moe_permute(const void* tokens, void* permuted, const int32_t* row_id_map,
                              int64_t n_tokens, int64_t n_rows, int64_t dim, int dtype,
                              hx_stream stream) {
  const int64_t es = dtype_size(dtype);
  if (es == 0) return HX_ERR_DTYPE;
  if (n_tokens < 0 || n_rows <= 0 || dim <= 0) return HX_ERR_SHAPE;
  if (n_tokens == 0) return HX_OK;
  if (!tokens || !permuted || !row_id_map) return HX_ERR_NULL;
  hipStream_t s = (hipStream_t)stream;
  const int64_t row_bytes = dim * es;
  if (row_bytes % 16 == 0 && aligned16(tokens) && aligned16(permuted)) {
    hx::launcher(permute_kernel<uint4>, (unsigned)n_tokens, 256, 0, s)((const uint4*)tokens, (uint4*)permuted,
                                                            row_id_map, n_tokens, (int)n_rows,
                                                            row_bytes / 16);
  } else if (es == 2) {
    hx::launcher(permute_kernel<uint16_t>, (unsigned)n_tokens, 256, 0, s)(
        (const uint16_t*)tokens, (uint16_t*)permuted, row_id_map, n_tokens, (int)n_rows, dim);
  } else {
    hx::launcher(permute_kernel<uint32_t>, (unsigned)n_tokens, 256, 0, s)(
        (const uint32_t*)tokens, (uint32_t*)permuted, row_id_map, n_tokens, (int)n_rows, dim);
  }
  return check_launch();
}

extern "C" int hx_moe_unpermute(const void* permuted, void* out, const int32_t* row_id_map,
                                const void* probs, int64_t n_tokens, int64_t n_rows, int64_t dim,
                                int dtype, hx_stream stream) {
  if (n_tokens < 0 || n_rows <= 0 || dim <= 0) return HX_ERR_SHAPE;
  if (n_tokens == 0) return HX_OK;
  if (!permuted || !out || !row_id_map) return HX_ERR_NULL;
  hipStream_t s = (hipStream_t)stream;
  if ((dtype == HX_F16 || dtype == HX_BF16) && n_rows <= 256 && dim % 8 == 0 && aligned16(permuted) && aligned16(out)) {
    const dim3 grid((unsigned)n_tokens, row_splits(n_tokens, dim));
    if (dtype == HX_F16)
      hx::launcher(unpermute_vec_kernel<F16>, grid, 256, 0, s)((const u16*)permuted, (u16*)out, row_id_map, (const u16*)probs,
                                                               n_tokens, (int)n_rows, dim);
    else
      hx::launcher(unpermute_vec_kernel<BF16>, grid, 256, 0, s)((const u16*)permuted, (u16*)out, row_id_map, (const u16*)probs,
                                                                n_tokens, (int)n_rows, dim);
    return check_launch();
  }
  switch (dtype) {
    case HX_F32:
      hx::launcher(unpermute_kernel<F32>, (unsigned)n_tokens, 256, 0, s)(
          (const float*)permuted, (float*)out, row_id_map, (const float*)probs, n_tokens,
          (int)n_rows, dim);
      break;
    case HX_F16:
      hx::launcher(unpermute_kernel<F16>, (unsigned)n_tokens, 256, 0, s)(
          (const u16*)permuted, (u16*)out, row_id_map, (const u16*)probs, n_tokens, (int)n_rows, dim);
      break;
    case HX_BF16:
      hx::launcher(unpermute_kernel<BF16>, (unsigned)n_tokens, 256, 0, s)(
          (const u16*)permuted, (u16*)out, row_id_map, (const u16*)probs, n_tokens, (int)n_rows, dim);
      break;
    default: return HX_ERR_DTYPE;
  }
  return check_launch();
}

extern "C" int hx_moe_sum_out(const void* in, void* out, int64_t n_tokens, int64_t topk,
                              int64_t dim, int dtype, hx_stream stream) {
  if (n_tokens < 0 || topk <= 0 || dim <= 0) return HX_ERR_SHAPE;
  if (n_tokens == 0) return HX_OK;
  if (!in || !out) return HX_ERR_NULL;
  hipStream_t s = (hipStream_t)stream;
  const int ta = (topk == 2 || topk == 3 || topk == 4 || topk == 8) ? 1 : 0;      // topk_sum_kernel's instantiations
  if ((dtype == HX_F16 || dtype == HX_BF16) && dim % 8 == 0 && aligned16(in) && aligned16(out)) {
    const dim3 grid((unsigned)n_tokens, row_splits(n_tokens, dim));
    if (dtype == HX_F16) hx::launcher(sum_out_vec_kernel<F16>, grid, 256, 0, s)((const u16*)in, (u16*)out, (int)topk, dim, ta);
    else hx::launcher(sum_out_vec_kernel<BF16>, grid, 256, 0, s)((const u16*)in, (u16*)out, (int)topk, dim, ta);
    return check_launch();
  }
  switch (dtype) {
    case HX_F32:
      hx::launcher(sum_out_kernel<F32>, (unsigned)n_tokens, 256, 0, s)((const float*)in, (float*)out, (int)topk, dim, ta);
      break;
    case HX_F16:
      hx::launcher(sum_out_kernel<F16>, (unsigned)n_tokens, 256, 0, s)((const u16*)in, (u16*)out, (int)topk, dim, ta);
      break;
    case HX_BF16:
      hx::launcher(sum_out_kernel<BF16>, (unsigned)n_tokens, 256, 0, s)((const u16*)in, (u16*)out, (int)topk, dim, ta);
      break;
    default: return HX_ERR_DTYPE;
  }
  return check_launch();
}
