// norm_rope_act.hip — rms_norm, rotary embedding, silu (+ fused silu*mul, add+rms_norm).
// All HBM-bound row kernels: 16-byte vector loads, one pass over the row held in
// registers, wave64 shuffle reductions.  Rounding points follow the reference CUDA
// kernels (csrc/kernel/norm/rms_norm.cu:14-41, csrc/kernel/position_embedding/
// rope.cu:12-79, csrc/kernel/activation/activation.cu:13-33).
#include "hx_common.h"

namespace {

using namespace hx;

// ---------------------------------------------------------------------------
// vector row access: VEC elements of T per lane per step (16 bytes)
// ---------------------------------------------------------------------------
template <typename T> struct VecOf;
template <> struct VecOf<F32> { static constexpr int N = 4; typedef f32x4 type; };
template <> struct VecOf<F16> { static constexpr int N = 8; typedef u16x8 type; };
template <> struct VecOf<BF16> { static constexpr int N = 8; typedef u16x8 type; };

__device__ __forceinline__ float block_sum_256(float v, float* red) {
  v = wave_sum(v);
  const int wid = threadIdx.x >> 6, lane = threadIdx.x & 63;
  if (lane == 0) red[wid] = v;
  __syncthreads();
  float t = red[0] + red[1] + red[2] + red[3];
  return t;
}

// rms_norm: one 256-thread workgroup per row; the row lives in registers
// (up to MAXV vectors per thread) between the reduction and the scaling pass.
// ADD: h = residual + x (T arithmetic), residual <- h, then norm(h).
template <typename T, int MAXV, bool ADD>
__global__ __launch_bounds__(256) void rms_norm_vec_kernel(
    typename T::storage* __restrict__ out, typename T::storage* __restrict__ residual,
    const typename T::storage* __restrict__ input, const typename T::storage* __restrict__ weight,
    float eps, int32_t hidden) {
  typedef typename VecOf<T>::type V;
  constexpr int N = VecOf<T>::N;
  __shared__ float red[4];
  const int64_t row = blockIdx.x;
  const int nvec = hidden / N;
  const V* in_v = reinterpret_cast<const V*>(input + row * hidden);
  V* res_v = ADD ? reinterpret_cast<V*>(residual + row * hidden) : nullptr;
  V* out_v = reinterpret_cast<V*>(out + row * hidden);
  const V* w_v = reinterpret_cast<const V*>(weight);

  float x[MAXV][N];
  float ss = 0.f;
#pragma unroll
  for (int j = 0; j < MAXV; ++j) {
    const int i = threadIdx.x + j * 256;
    if (i < nvec) {
      V v = in_v[i];
      if (ADD) {
        V r = res_v[i];
        V h;
#pragma unroll
        for (int e = 0; e < N; ++e) {
          float s = round_to<T>(T::to_float(v[e]) + T::to_float(r[e]));
          x[j][e] = s;
          h[e] = T::from_float(s);
        }
        res_v[i] = h;
      } else {
#pragma unroll
        for (int e = 0; e < N; ++e) x[j][e] = T::to_float(v[e]);
      }
#pragma unroll
      for (int e = 0; e < N; ++e) ss += x[j][e] * x[j][e];
    }
  }
  const float total = block_sum_256(ss, red);
  const float inv = rsqrtf(total / (float)hidden + eps);
#pragma unroll
  for (int j = 0; j < MAXV; ++j) {
    const int i = threadIdx.x + j * 256;
    if (i < nvec) {
      V w = w_v[i];
      V o;
#pragma unroll
      for (int e = 0; e < N; ++e) {
        // (T)(x * s_variance) * weight  — two T roundings (rms_norm.cu:39)
        float n = round_to<T>(x[j][e] * inv);
        o[e] = T::from_float(n * T::to_float(w[e]));
      }
      out_v[i] = o;
    }
  }
}

// generic (any hidden, any alignment): re-reads the row
template <typename T, bool ADD>
__global__ __launch_bounds__(256) void rms_norm_generic_kernel(
    typename T::storage* __restrict__ out, typename T::storage* __restrict__ residual,
    const typename T::storage* __restrict__ input, const typename T::storage* __restrict__ weight,
    float eps, int64_t hidden) {
  __shared__ float red[4];
  const int64_t row = blockIdx.x;
  const typename T::storage* in = input + row * hidden;
  typename T::storage* res = ADD ? residual + row * hidden : nullptr;
  float ss = 0.f;
  for (int64_t i = threadIdx.x; i < hidden; i += 256) {
    float x = T::to_float(in[i]);
    if (ADD) {
      x = round_to<T>(x + T::to_float(res[i]));
      res[i] = T::from_float(x);
    }
    ss += x * x;
  }
  const float total = block_sum_256(ss, red);
  const float inv = rsqrtf(total / (float)hidden + eps);
  for (int64_t i = threadIdx.x; i < hidden; i += 256) {
    const float x = ADD ? T::to_float(res[i]) : T::to_float(in[i]);
    float n = round_to<T>(x * inv);
    out[row * hidden + i] = T::from_float(n * T::to_float(weight[i]));
  }
}

template <typename T, bool ADD>
int launch_rms(void* out, void* residual, const void* input, const void* weight, float eps,
               int64_t rows, int64_t hidden, hipStream_t stream) {
  typedef typename T::storage S;
  constexpr int N = VecOf<T>::N;
  if (rows == 0) return HX_OK;
  const bool vec = (hidden % N == 0) && aligned16(out) && aligned16(input) && aligned16(weight) &&
                   (!ADD || aligned16(residual));
  dim3 grid((unsigned)rows);
  if (vec && hidden / N <= 256 * 1) {
    hx::launcher(rms_norm_vec_kernel<T, 1, ADD>, grid, 256, 0, stream)((S*)out, (S*)residual,
                                                            (const S*)input, (const S*)weight,
                                                            eps, (int)hidden);
  } else if (vec && hidden / N <= 256 * 2) {
    hx::launcher(rms_norm_vec_kernel<T, 2, ADD>, grid, 256, 0, stream)((S*)out, (S*)residual,
                                                            (const S*)input, (const S*)weight,
                                                            eps, (int)hidden);
  } else if (vec && hidden / N <= 256 * 4) {
    hx::launcher(rms_norm_vec_kernel<T, 4, ADD>, grid, 256, 0, stream)((S*)out, (S*)residual,
                                                            (const S*)input, (const S*)weight,
                                                            eps, (int)hidden);
  } else {
    hx::launcher(rms_norm_generic_kernel<T, ADD>, grid, 256, 0, stream)((S*)out, (S*)residual,
                                                             (const S*)input, (const S*)weight,
                                                             eps, hidden);
  }
  return check_launch();
}

// ---------------------------------------------------------------------------
// Vision tower (CLIP ViT) fusions — extensions; the reference runs these as separate torch ops
// (hydrainfer/model/clip.py: residual add + nn.LayerNorm; hydrainfer/layer/activation.py:17-22 QuickGELU).
// At 577 x 1024 every one of those ops is a ~5 us launch-bound kernel: two launches per fused op are saved.
// add_layer_norm: h = residual + x (one T rounding), residual <- h, out = (T)((h - mean) * rstd * w + b) with the
// moments in fp32 over the T-rounded h — torch.nn.functional.layer_norm's arithmetic (its reduction order is
// the library's own: results agree to the last place of T except where a tie falls between two roundings).
// ---------------------------------------------------------------------------
template <typename T, int MAXV, bool ADD>
__global__ __launch_bounds__(256) void layer_norm_vec_kernel(
    typename T::storage* __restrict__ out, typename T::storage* __restrict__ residual,
    const typename T::storage* __restrict__ input, const typename T::storage* __restrict__ weight,
    const typename T::storage* __restrict__ bias, float eps, int32_t hidden) {
  typedef typename VecOf<T>::type V;
  constexpr int N = VecOf<T>::N;
  __shared__ float red[2][4];
  const int64_t row = blockIdx.x;
  const int nvec = hidden / N;
  const V* in_v = ADD ? reinterpret_cast<const V*>(input + row * hidden) : nullptr;
  V* res_v = reinterpret_cast<V*>(residual + row * hidden);
  V* out_v = reinterpret_cast<V*>(out + row * hidden);
  float x[MAXV][N];
  float sum = 0.f;
#pragma unroll
  for (int j = 0; j < MAXV; ++j) {
    const int i = threadIdx.x + j * 256;
    if (i < nvec) {
      const V r = res_v[i];
      if (ADD) {
        const V v = in_v[i];
        V h;
#pragma unroll
        for (int e = 0; e < N; ++e) {
          x[j][e] = round_to<T>(T::to_float(v[e]) + T::to_float(r[e]));
          h[e] = T::from_float(x[j][e]);
        }
        res_v[i] = h;
      } else {
#pragma unroll
        for (int e = 0; e < N; ++e) x[j][e] = T::to_float(r[e]);
      }
#pragma unroll
      for (int e = 0; e < N; ++e) sum += x[j][e];
    }
  }
  const float mean = block_sum_256(sum, red[0]) / (float)hidden;
  float sq = 0.f;
#pragma unroll
  for (int j = 0; j < MAXV; ++j) {
    const int i = threadIdx.x + j * 256;
    if (i < nvec) {
#pragma unroll
      for (int e = 0; e < N; ++e) { const float d = x[j][e] - mean; sq += d * d; }
    }
  }
  const float rstd = rsqrtf(block_sum_256(sq, red[1]) / (float)hidden + eps);
  const V* w_v = reinterpret_cast<const V*>(weight);
  const V* b_v = reinterpret_cast<const V*>(bias);
#pragma unroll
  for (int j = 0; j < MAXV; ++j) {
    const int i = threadIdx.x + j * 256;
    if (i < nvec) {
      const V w = w_v[i], b = b_v[i];
      V o;
#pragma unroll
      for (int e = 0; e < N; ++e) o[e] = T::from_float((x[j][e] - mean) * rstd * T::to_float(w[e]) + T::to_float(b[e]));
      out_v[i] = o;
    }
  }
}

template <typename T, bool ADD>
int launch_layer_norm(void* out, void* residual, const void* input, const void* weight, const void* bias, float eps,
                      int64_t rows, int64_t hidden, hipStream_t stream) {
  typedef typename T::storage S;
  constexpr int N = VecOf<T>::N;
  if (hidden % N || hidden / N > 256 * 4) return HX_ERR_SHAPE;      // rows of up to 8192 (4096 fp32) elements
  if (!aligned16(out) || !aligned16(residual) || (ADD && !aligned16(input)) || !aligned16(weight) || !aligned16(bias))
    return HX_ERR_STRIDE;
  const dim3 grid((unsigned)rows);
  const int nv = (int)(hidden / N);
#define HX_LN(MV) hx::launcher(layer_norm_vec_kernel<T, MV, ADD>, grid, 256, 0, stream)((S*)out, (S*)residual, (const S*)input, \
                                                                                     (const S*)weight, (const S*)bias, eps, (int)hidden)
  if (nv <= 256) HX_LN(1);
  else if (nv <= 512) HX_LN(2);
  else HX_LN(4);
#undef HX_LN
  return check_launch();
}

// quick_gelu: out = x * sigmoid(1.702 x) with the reference's three T roundings (the scaled copy, the sigmoid, the
// product: activation.py:17-22 runs them as three torch ops); exact expf — the op is memory-bound
template <typename T>
__global__ __launch_bounds__(256) void quick_gelu_vec_kernel(typename T::storage* __restrict__ out,
                                                             const typename T::storage* __restrict__ in, int32_t nvec,
                                                             int64_t in_stride, int64_t out_stride) {
  typedef typename VecOf<T>::type V;
  constexpr int N = VecOf<T>::N;
  const int64_t row = blockIdx.y;
  const V* g = reinterpret_cast<const V*>(in + row * in_stride);
  V* o = reinterpret_cast<V*>(out + row * out_stride);
  for (int i = blockIdx.x * 256 + threadIdx.x; i < nvec; i += gridDim.x * 256) {
    const V v = g[i];
    V r;
#pragma unroll
    for (int e = 0; e < N; ++e) {
      // each product is rounded to fp32 FIRST, as the torch ops round it (opmath float, then the store's conversion): left
      // to itself the compiler folds product + conversion into v_fma_mixlo_f16 — ONE rounding — and 4 in 10^4 fp16
      // results land on the other side of a tie
      const float a = T::to_float(v[e]);
      float p1 = 1.702f * a;
      asm volatile("" : "+v"(p1));
      const float t = round_to<T>(p1);
      const float sg = round_to<T>(1.0f / (1.0f + expf(-t)));
      float p2 = a * sg;
      asm volatile("" : "+v"(p2));
      r[e] = T::from_float(p2);
    }
    o[i] = r;
  }
}

template <typename T>
int launch_quick_gelu(void* out, const void* in, int64_t rows, int64_t n, int64_t in_stride, hipStream_t stream) {
  typedef typename T::storage S;
  constexpr int N = VecOf<T>::N;
  if (n % N || in_stride % N) return HX_ERR_SHAPE;
  if (!aligned16(out) || !aligned16(in)) return HX_ERR_STRIDE;
  const int nvec = (int)(n / N);
  int gx = (nvec + 255) / 256;
  if (gx > 64) gx = 64;
  for (int64_t r0 = 0; r0 < rows; r0 += 65535) {
    const int64_t nr = rows - r0 < 65535 ? rows - r0 : 65535;
    hx::launcher(quick_gelu_vec_kernel<T>, dim3((unsigned)gx, (unsigned)nr), 256, 0, stream)(
        (S*)out + r0 * n, (const S*)in + r0 * in_stride, nvec, in_stride, n);
    const int rc = check_launch();
    if (rc) return rc;
  }
  return HX_OK;
}

// ---------------------------------------------------------------------------
// Step-edge fusions of the decode loop (extensions; each bit-identical to the two ops it replaces).
// embed_rms_norm: h = table[ids] (torch.nn.functional.embedding, hydrainfer/model/llama.py:80-83) and
// x = rms_norm(h) * w in one launch — same arithmetic and reduction as rms_norm_vec_kernel.
// ---------------------------------------------------------------------------
template <typename T, int MAXV>
__device__ __forceinline__ void embed_rms_norm_row(typename T::storage* __restrict__ h_out, typename T::storage* __restrict__ x_out,
                                                   int64_t row, int64_t id, const typename T::storage* __restrict__ table,
                                                   const typename T::storage* __restrict__ weight, float eps, int32_t hidden,
                                                   int64_t vocab, float* red) {
  typedef typename VecOf<T>::type V;
  constexpr int N = VecOf<T>::N;
  id = id < 0 ? 0 : (id >= vocab ? vocab - 1 : id);     // the torch op asserts; here out-of-range ids clamp
  const int nvec = hidden / N;
  const V* in_v = reinterpret_cast<const V*>(table + id * hidden);
  V* h_v = reinterpret_cast<V*>(h_out + row * hidden);
  V* out_v = reinterpret_cast<V*>(x_out + row * hidden);
  const V* w_v = reinterpret_cast<const V*>(weight);
  float x[MAXV][N];
  V wv[MAXV];
  float ss = 0.f;
#pragma unroll
  for (int j = 0; j < MAXV; ++j) {
    const int i = threadIdx.x + j * 256;
    if (i < nvec) {
      const V v = in_v[i];
      wv[j] = w_v[i];
      h_v[i] = v;
#pragma unroll
      for (int e = 0; e < N; ++e) {
        x[j][e] = T::to_float(v[e]);
        ss += x[j][e] * x[j][e];
      }
    }
  }
  const float total = block_sum_256(ss, red);
  const float inv = rsqrtf(total / (float)hidden + eps);
#pragma unroll
  for (int j = 0; j < MAXV; ++j) {
    const int i = threadIdx.x + j * 256;
    if (i < nvec) {
      V o;
#pragma unroll
      for (int e = 0; e < N; ++e) o[e] = T::from_float(round_to<T>(x[j][e] * inv) * T::to_float(wv[j][e]));
      out_v[i] = o;
    }
  }
}

template <typename T, int MAXV>
__global__ __launch_bounds__(256) void embed_rms_norm_kernel(
    typename T::storage* __restrict__ h_out, typename T::storage* __restrict__ x_out,
    const void* __restrict__ ids, int ids_i64, const typename T::storage* __restrict__ table,
    const typename T::storage* __restrict__ weight, float eps, int32_t hidden, int64_t vocab) {
  __shared__ float red[4];
  const int64_t row = blockIdx.x;
  const int64_t id = ids_i64 ? reinterpret_cast<const int64_t*>(ids)[row] : (int64_t)reinterpret_cast<const int32_t*>(ids)[row];
  embed_rms_norm_row<T, MAXV>(h_out, x_out, row, id, table, weight, eps, hidden, vocab, red);
}

// hx_decode_step_head: everything a decode step does before its first GEMM, ONE launch instead of three or four
// (round 3's timeline: zero_kernel 4.7 us, decode_advance 4.5, embed_rms_norm 4.8 — each a launch of microseconds of
// nothing).  Workgroup roles by index: [0, rows) embedding gather + first RMSNorm of row b (its id optionally taken
// from the previous launch's samples: hx_decode_feed_ids), [rows, rows + n_zero) zero 4 KiB each of the hand-over
// areas, the last one (if batch > 0) the metadata advance.  The roles touch disjoint memory; each is bit-identical to
// the separate launch it replaces.
struct StepHeadParams {
  void* h_out; void* x_out; const void* ids; const int32_t* feed_src; const int64_t* feed_prev; int64_t* fed_out;
  const void* table; const void* weight; uint32_t* zero_ptr;
  int32_t* positions; int32_t* kv_lens; int32_t* cu_seqlens_k; int32_t* new_cache_slots;
  const int32_t* block_table; const int32_t* cu_block_lens; int32_t* rank_desc;
  int64_t vocab, zero_words;
  float eps;
  int32_t ids_i64, rows, hidden, n_zero, batch, block_size, stride;
};

template <typename T, int MAXV>
__global__ __launch_bounds__(256) void decode_step_head_kernel(const StepHeadParams p) {
  __shared__ float red[4];
  __shared__ int32_t scan[256];
  const int b = blockIdx.x;
  if (b < p.rows) {
    int64_t id = p.ids_i64 ? reinterpret_cast<const int64_t*>(p.ids)[b] : (int64_t)reinterpret_cast<const int32_t*>(p.ids)[b];
    if (p.feed_src) {
      const int32_t s = p.feed_src[b];
      if (s >= 0) id = p.feed_prev[s];
    }
    if (p.fed_out && threadIdx.x == 0) p.fed_out[b] = id;
    embed_rms_norm_row<T, MAXV>(reinterpret_cast<u16*>(p.h_out), reinterpret_cast<u16*>(p.x_out), b, id,
                                reinterpret_cast<const u16*>(p.table), reinterpret_cast<const u16*>(p.weight), p.eps, p.hidden,
                                p.vocab, red);
  } else if (b < p.rows + p.n_zero) {
    const int64_t w0 = (int64_t)(b - p.rows) * 1024 + threadIdx.x * 4;      // 4 KiB per workgroup, 16 B per thread
    if (w0 + 4 <= p.zero_words) {
      *reinterpret_cast<u32x4*>(p.zero_ptr + w0) = u32x4{0u, 0u, 0u, 0u};
    } else {
      for (int64_t i = w0; i < p.zero_words; ++i) p.zero_ptr[i] = 0u;
    }
  } else {
    decode_advance_block(p.positions, p.kv_lens, p.cu_seqlens_k, p.new_cache_slots, p.block_table, p.cu_block_lens, p.batch,
                         p.block_size, p.stride, scan);
    if (p.rank_desc) decode_rank_block(p.kv_lens, p.batch, p.rank_desc, scan);
  }
}

// argmax over each row of logits [rows, n] (row stride ld): the greedy sampler
// (hydrainfer/model/llama.py:99-104, torch.argmax).  torch's rule: the largest value, NaN counts as
// larger than everything, ties go to the smallest index.
__device__ __forceinline__ bool arg_better(float a, int ia, float b, int ib) {
  const bool an = a != a, bn = b != b;
  if (an != bn) return an;
  if (!an && a != b) return a > b;
  return ia < ib;
}

template <typename T>
__global__ __launch_bounds__(1024) void argmax_rows_kernel(int64_t* __restrict__ out,
                                                           const typename T::storage* __restrict__ logits,
                                                           int32_t n, int64_t ld) {
  __shared__ float s_v[16];
  __shared__ int s_i[16];
  const int64_t row = blockIdx.x;
  const typename T::storage* p = logits + row * ld;
  float best = -INFINITY;
  int bi = 0x7fffffff;
  const bool vec = (ld % 8 == 0) && ((reinterpret_cast<uintptr_t>(logits) & 15) == 0);
  if (vec) {
    const int nvec = n / 8;
    // four independent 16-byte loads per thread in flight (a 32064-wide row is 3.9 per thread): the row is read in one
    // memory round trip instead of four dependent ones (round 4: 9.0 us per step for a 2 MB read; round 5: see profiles/)
    for (int i0 = threadIdx.x; i0 < nvec; i0 += 4096) {
      u16x8 v[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) v[u] = *reinterpret_cast<const u16x8*>(p + (int64_t)min(i0 + 1024 * u, nvec - 1) * 8);
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int i = i0 + 1024 * u;
        if (i < nvec) {
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            const float f = T::to_float(v[u][e]);
            if (arg_better(f, i * 8 + e, best, bi)) { best = f; bi = i * 8 + e; }
          }
        }
      }
    }
    for (int i = nvec * 8 + threadIdx.x; i < n; i += 1024) {
      const float f = T::to_float(p[i]);
      if (arg_better(f, i, best, bi)) { best = f; bi = i; }
    }
  } else {
    for (int i = threadIdx.x; i < n; i += 1024) {
      const float f = T::to_float(p[i]);
      if (arg_better(f, i, best, bi)) { best = f; bi = i; }
    }
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    const float ov = __shfl_xor(best, off, 64);
    const int oi = __shfl_xor(bi, off, 64);
    if (arg_better(ov, oi, best, bi)) { best = ov; bi = oi; }
  }
  if ((threadIdx.x & 63) == 0) { s_v[threadIdx.x >> 6] = best; s_i[threadIdx.x >> 6] = bi; }
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int k = 1; k < 16; ++k)
      if (arg_better(s_v[k], s_i[k], best, bi)) { best = s_v[k]; bi = s_i[k]; }
    out[row] = bi;
  }
}

// ---------------------------------------------------------------------------
// RoPE.  Work item = (token, head (q heads then kv heads), pair index).
// One thread rotates PV pairs: x' = x*c - y*s ; y' = x*s + y*c with every
// operation rounded to T (rope.cu:22-27).  fp contraction is disabled so the
// fp32 path also matches an unfused CPU evaluation bit for bit.
// ---------------------------------------------------------------------------
template <typename T>
__device__ __forceinline__ void rotate_pair(float x, float y, float c, float s, float& xo,
                                            float& yo) {
#pragma clang fp contract(off)
  const float xc = round_to<T>(x * c);
  const float ys = round_to<T>(y * s);
  const float xs = round_to<T>(x * s);
  const float yc = round_to<T>(y * c);
  xo = round_to<T>(xc - ys);
  yo = round_to<T>(xs + yc);
}

template <typename T>
__global__ __launch_bounds__(256) void rope_kernel(
    typename T::storage* __restrict__ q, typename T::storage* __restrict__ k,
    const int32_t* __restrict__ positions, const typename T::storage* __restrict__ cos_sin,
    int32_t n_heads, int32_t n_kv_heads, int32_t head_dim, int32_t rotary_dim, int64_t q_stride,
    int64_t k_stride, int32_t interleaved) {
  const int token = blockIdx.x;
  const int half = rotary_dim >> 1;
  const int total = (n_heads + n_kv_heads) * half;
  const typename T::storage* cs = cos_sin + (int64_t)positions[token] * rotary_dim;
  for (int i = threadIdx.x; i < total; i += 256) {
    const int h = i / half;
    const int r = i - h * half;
    typename T::storage* base = (h < n_heads) ? (q + token * q_stride + (int64_t)h * head_dim)
                                              : (k + token * k_stride +
                                                 (int64_t)(h - n_heads) * head_dim);
    const int xi = interleaved ? 2 * r : r;
    const int yi = interleaved ? 2 * r + 1 : r + half;
    const float c = T::to_float(cs[r]);
    const float s = T::to_float(cs[half + r]);
    float xo, yo;
    rotate_pair<T>(T::to_float(base[xi]), T::to_float(base[yi]), c, s, xo, yo);
    base[xi] = T::from_float(xo);
    base[yi] = T::from_float(yo);
  }
}

// Vectorised NeoX (non-interleaved) variant: each thread owns N consecutive pairs:
// 16-byte loads of x[r..r+N), y[r..r+N), cos[r..), sin[r..).  Requires half % N == 0.
template <typename T>
__global__ __launch_bounds__(256) void rope_neox_vec_kernel(
    typename T::storage* __restrict__ q, typename T::storage* __restrict__ k,
    const int32_t* __restrict__ positions, const typename T::storage* __restrict__ cos_sin,
    int32_t n_heads, int32_t n_kv_heads, int32_t head_dim, int32_t rotary_dim, int64_t q_stride,
    int64_t k_stride) {
  typedef typename VecOf<T>::type V;
  constexpr int N = VecOf<T>::N;
  const int token = blockIdx.x;
  const int half = rotary_dim >> 1;
  const int vph = half / N;  // vectors per head-half
  const int total = (n_heads + n_kv_heads) * vph;
  const typename T::storage* cs = cos_sin + (int64_t)positions[token] * rotary_dim;
  for (int i = threadIdx.x; i < total; i += 256) {
    const int h = i / vph;
    const int rv = i - h * vph;
    typename T::storage* base = (h < n_heads) ? (q + token * q_stride + (int64_t)h * head_dim)
                                              : (k + token * k_stride +
                                                 (int64_t)(h - n_heads) * head_dim);
    V* xp = reinterpret_cast<V*>(base + rv * N);
    V* yp = reinterpret_cast<V*>(base + half + rv * N);
    const V cv = *reinterpret_cast<const V*>(cs + rv * N);
    const V sv = *reinterpret_cast<const V*>(cs + half + rv * N);
    V xv = *xp, yv = *yp, xo, yo;
#pragma unroll
    for (int e = 0; e < N; ++e) {
      float a, b;
      rotate_pair<T>(T::to_float(xv[e]), T::to_float(yv[e]), T::to_float(cv[e]),
                     T::to_float(sv[e]), a, b);
      xo[e] = T::from_float(a);
      yo[e] = T::from_float(b);
    }
    *xp = xo;
    *yp = yo;
  }
}


// Fused RoPE + set_kv_cache: rotates q and k in place exactly like
// rope_neox_vec_kernel, and in the same pass writes the rotated k row and the v row of the
// token into the paged cache at slot = block*block_size + offset.  One launch replaces two.
template <typename T>
__global__ __launch_bounds__(256) void rope_cache_neox_vec_kernel(
    typename T::storage* __restrict__ q, typename T::storage* __restrict__ k,
    const typename T::storage* __restrict__ v, const int32_t* __restrict__ positions,
    const typename T::storage* __restrict__ cos_sin, const int32_t* __restrict__ slot_ids,
    typename T::storage* __restrict__ key_cache, typename T::storage* __restrict__ value_cache,
    int32_t n_heads, int32_t n_kv_heads, int32_t head_dim, int32_t rotary_dim, int64_t q_stride,
    int64_t k_stride, int64_t v_stride, int64_t kc_block_stride, int64_t vc_block_stride,
    int32_t block_size) {
  typedef typename VecOf<T>::type V;
  constexpr int N = VecOf<T>::N;
  const int token = blockIdx.x;
  const int half = rotary_dim >> 1;
  const int vph = half / N;
  const int total = (n_heads + n_kv_heads) * vph;
  const typename T::storage* cs = cos_sin + (int64_t)positions[token] * rotary_dim;
  const int slot = slot_ids[token];
  const int64_t row_elems = (int64_t)n_kv_heads * head_dim;
  const int64_t blk = slot / block_size, off = slot % block_size;
  typename T::storage* kc_row = key_cache + blk * kc_block_stride + off * row_elems;
  typename T::storage* vc_row = value_cache + blk * vc_block_stride + off * row_elems;
  for (int i = threadIdx.x; i < total; i += 256) {
    const int h = i / vph;
    const int rv = i - h * vph;
    const bool is_k = h >= n_heads;
    typename T::storage* base = !is_k ? (q + token * q_stride + (int64_t)h * head_dim)
                                      : (k + token * k_stride + (int64_t)(h - n_heads) * head_dim);
    V* xp = reinterpret_cast<V*>(base + rv * N);
    V* yp = reinterpret_cast<V*>(base + half + rv * N);
    const V cv = *reinterpret_cast<const V*>(cs + rv * N);
    const V sv = *reinterpret_cast<const V*>(cs + half + rv * N);
    V xv = *xp, yv = *yp, xo, yo;
#pragma unroll
    for (int e = 0; e < N; ++e) {
      float a, b;
      rotate_pair<T>(T::to_float(xv[e]), T::to_float(yv[e]), T::to_float(cv[e]),
                     T::to_float(sv[e]), a, b);
      xo[e] = T::from_float(a);
      yo[e] = T::from_float(b);
    }
    *xp = xo;
    *yp = yo;
    if (is_k && slot >= 0) {
      typename T::storage* crow = kc_row + (int64_t)(h - n_heads) * head_dim;
      *reinterpret_cast<V*>(crow + rv * N) = xo;
      *reinterpret_cast<V*>(crow + half + rv * N) = yo;
    }
  }
  if (slot >= 0) {
    // pass-through dims of k (rotary_dim < head_dim) and the whole v row
    const int tail_v = (head_dim - rotary_dim) / N;
    for (int i = threadIdx.x; i < n_kv_heads * tail_v; i += 256) {
      const int h = i / tail_v, t = i - h * tail_v;
      const int64_t o = (int64_t)h * head_dim + rotary_dim + t * N;
      *reinterpret_cast<V*>(kc_row + o) = *reinterpret_cast<const V*>(k + token * k_stride + o);
    }
    const int vvec = (int)(row_elems / N);
    const V* vs = reinterpret_cast<const V*>(v + token * v_stride);
    V* vd = reinterpret_cast<V*>(vc_row);
    for (int i = threadIdx.x; i < vvec; i += 256) vd[i] = vs[i];
  }
}

template <typename T>
int launch_rope(void* q, void* k, const int32_t* positions, const void* cos_sin,
                int64_t n_tokens, int64_t n_heads, int64_t n_kv_heads, int64_t head_dim,
                int64_t rotary_dim, int64_t q_stride, int64_t k_stride, int interleaved,
                hipStream_t stream) {
  typedef typename T::storage S;
  constexpr int N = VecOf<T>::N;
  if (n_tokens == 0) return HX_OK;
  const int64_t half = rotary_dim / 2;
  const int64_t es = sizeof(S);
  const bool vec = !interleaved && (half % N == 0) && (head_dim % N == 0) &&
                   (q_stride * es % 16 == 0) && (k_stride * es % 16 == 0) && aligned16(q) &&
                   aligned16(k) && aligned16(cos_sin);
  dim3 grid((unsigned)n_tokens);
  if (vec) {
    hx::launcher(rope_neox_vec_kernel<T>, grid, 256, 0, stream)((S*)q, (S*)k, positions, (const S*)cos_sin,
                                                     (int)n_heads, (int)n_kv_heads,
                                                     (int)head_dim, (int)rotary_dim, q_stride,
                                                     k_stride);
  } else {
    hx::launcher(rope_kernel<T>, grid, 256, 0, stream)((S*)q, (S*)k, positions, (const S*)cos_sin,
                                            (int)n_heads, (int)n_kv_heads, (int)head_dim,
                                            (int)rotary_dim, q_stride, k_stride, interleaved);
  }
  return check_launch();
}

// ---------------------------------------------------------------------------
// SiLU: (T)(x / (1 + exp(-x))) in fp32 (activation.cu:16-19).  MUL: times `up`
// in T arithmetic (model_forward.py:36).
// ---------------------------------------------------------------------------
__device__ __forceinline__ float silu_f32(float x) { return x / (1.0f + __expf(-x)); }

template <typename T, bool MUL>
__global__ __launch_bounds__(256) void silu_vec_kernel(
    typename T::storage* __restrict__ out, const typename T::storage* __restrict__ gate,
    const typename T::storage* __restrict__ up, int32_t nvec, int64_t gate_stride,
    int64_t up_stride, int64_t out_stride) {
  typedef typename VecOf<T>::type V;
  constexpr int N = VecOf<T>::N;
  const int64_t row = blockIdx.y;
  const V* g = reinterpret_cast<const V*>(gate + row * gate_stride);
  const V* u = MUL ? reinterpret_cast<const V*>(up + row * up_stride) : nullptr;
  V* o = reinterpret_cast<V*>(out + row * out_stride);
  for (int i = blockIdx.x * 256 + threadIdx.x; i < nvec; i += gridDim.x * 256) {
    V gv = g[i], r;
    V uv;
    if (MUL) uv = u[i];
#pragma unroll
    for (int e = 0; e < N; ++e) {
      float a = silu_f32(T::to_float(gv[e]));
      if (MUL) a = round_to<T>(a) * T::to_float(uv[e]);
      r[e] = T::from_float(a);
    }
    o[i] = r;
  }
}

template <typename T, bool MUL>
__global__ __launch_bounds__(256) void silu_elem_kernel(
    typename T::storage* __restrict__ out, const typename T::storage* __restrict__ gate,
    const typename T::storage* __restrict__ up, int64_t n, int64_t gate_stride, int64_t up_stride,
    int64_t out_stride) {
  const int64_t row = blockIdx.y;
  for (int64_t i = blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    float a = silu_f32(T::to_float(gate[row * gate_stride + i]));
    if (MUL) a = round_to<T>(a) * T::to_float(up[row * up_stride + i]);
    out[row * out_stride + i] = T::from_float(a);
  }
}


// ---------------------------------------------------------------------------
// Split-K slab consumers (SURVEY.md §8f-2 + csrc/gemm_skinny.hip): the decode GEMM leaves
// fp32 partial sums partial[s][row][col]; these kernels add the splits in a fixed order,
// round ONCE to T (= the GEMM's output rounding) and continue exactly like their plain
// counterparts — so results are bit-identical to "reduce, then op".
// ---------------------------------------------------------------------------
template <typename T>
__device__ __forceinline__ void slab_sum8(const float* __restrict__ p, int n_splits,
                                          int64_t slab_stride, float (&acc)[8]) {
  // the slab pieces are L2 / HBM round trips: all loads of a batch of 6 splits are issued before
  // the first add (a loop of load-add pairs made the 11-split down projection pay 11 dependent
  // round trips); the adds stay in split order, so the sum is bit-identical
  constexpr int kB = 6;
  f32x4 a = {0.f, 0.f, 0.f, 0.f}, b = {0.f, 0.f, 0.f, 0.f};
  for (int s0 = 0; s0 < n_splits; s0 += kB) {
    f32x4 pa[kB], pb[kB];
#pragma unroll
    for (int k = 0; k < kB; ++k) {
      const int s = min(s0 + k, n_splits - 1);
      pa[k] = *reinterpret_cast<const f32x4*>(p + s * slab_stride);
      pb[k] = *reinterpret_cast<const f32x4*>(p + s * slab_stride + 4);
    }
#pragma unroll
    for (int k = 0; k < kB; ++k) {
      if (s0 + k == 0) { a = pa[0]; b = pb[0]; }
      else if (s0 + k < n_splits) { a += pa[k]; b += pb[k]; }
    }
  }
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    acc[e] = round_to<T>(a[e]);
    acc[4 + e] = round_to<T>(b[e]);
  }
}

template <typename T, int MAXV, int NT>
__global__ __launch_bounds__(NT) void add_rms_norm_slab_kernel(
    u16* __restrict__ out, u16* __restrict__ residual, const float* __restrict__ partial,
    int n_splits, int64_t slab_stride, const u16* __restrict__ weight, float eps, int32_t hidden,
    int frag_mb) {
  __shared__ float red[NT / 64];
  const int64_t row = blockIdx.x;
  const int nvec = hidden / 8;
  u16x8* res_v = reinterpret_cast<u16x8*>(residual + row * hidden);
  // frag_mb > 0: `out` is written in MFMA-B-fragment order for the activations-in-registers GEMM
  // (hydra_hip.h "fragment-major activations"): vector i of row r goes to piece
  // ((i / 4) * frag_mb + r / 16) * 64 + (i % 4) * 16 + r % 16
  u16x8* out_v = frag_mb ? reinterpret_cast<u16x8*>(out) + (row >> 4) * 64 + (row & 15)
                         : reinterpret_cast<u16x8*>(out + row * hidden);
  const u16x8* w_v = reinterpret_cast<const u16x8*>(weight);
  float x[MAXV][8];
  float ss = 0.f;
  // the residual and the norm weight are requested BEFORE the slab sums: the kernel is three
  // dependent round trips otherwise (slabs -> residual -> weight), and it is latency-bound
  u16x8 rr[MAXV], ww[MAXV];
#pragma unroll
  for (int j = 0; j < MAXV; ++j) {
    const int i = min(threadIdx.x + j * NT, nvec - 1);
    rr[j] = res_v[i];
    ww[j] = w_v[i];
  }
#pragma unroll
  for (int j = 0; j < MAXV; ++j) {
    const int i = threadIdx.x + j * NT;
    if (i < nvec) {
      float a[8];
      slab_sum8<T>(partial + row * hidden + i * 8, n_splits, slab_stride, a);
      const u16x8 r = rr[j];
      u16x8 h;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float sum = round_to<T>(a[e] + T::to_float(r[e]));
        x[j][e] = sum;
        h[e] = T::from_float(sum);
        ss += sum * sum;
      }
      res_v[i] = h;
    }
  }
  // block reduction over NT threads
  float tsum = wave_sum(ss);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = tsum;
  __syncthreads();
  float total = 0.f;
#pragma unroll
  for (int k = 0; k < NT / 64; ++k) total += red[k];
  const float inv = rsqrtf(total / (float)hidden + eps);
#pragma unroll
  for (int j = 0; j < MAXV; ++j) {
    const int i = threadIdx.x + j * NT;
    if (i < nvec) {
      const u16x8 w = ww[j];
      u16x8 o;
#pragma unroll
      for (int e = 0; e < 8; ++e)
        o[e] = T::from_float(round_to<T>(x[j][e] * inv) * T::to_float(w[e]));
      out_v[frag_mb ? (i >> 2) * frag_mb * 64 + (i & 3) * 16 : i] = o;
    }
  }
}

// out[row][i] = (T)silu(gate) * up, gate = col i, up = col inter + i of the [rows, 2*inter] slabs
template <typename T>
__global__ __launch_bounds__(256) void silu_mul_slab_kernel(u16* __restrict__ out,
                                                            const float* __restrict__ partial,
                                                            int n_splits, int64_t slab_stride,
                                                            int32_t inter, int frag_mb) {
  const int64_t row = blockIdx.y;
  const int nvec = inter / 8;
  const float* base = partial + row * 2 * inter;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < nvec; i += gridDim.x * 256) {
    float gte[8], up[8];
    slab_sum8<T>(base + i * 8, n_splits, slab_stride, gte);
    slab_sum8<T>(base + inter + i * 8, n_splits, slab_stride, up);
    u16x8 r;
#pragma unroll
    for (int e = 0; e < 8; ++e) r[e] = T::from_float(round_to<T>(silu_f32(gte[e])) * up[e]);
    if (frag_mb) reinterpret_cast<u16x8*>(out)[((i >> 2) * frag_mb + (row >> 4)) * 64 + (i & 3) * 16 + (row & 15)] = r;
    else *reinterpret_cast<u16x8*>(out + row * inter + i * 8) = r;
  }
}

template <typename T, bool MUL>
int launch_silu(void* out, const void* gate, const void* up, int64_t rows, int64_t n,
                int64_t gate_stride, int64_t up_stride, hipStream_t stream) {
  typedef typename T::storage S;
  constexpr int N = VecOf<T>::N;
  if (rows == 0 || n == 0) return HX_OK;
  const int64_t es = sizeof(S);
  const bool vec = (n % N == 0) && (gate_stride * es % 16 == 0) && aligned16(out) &&
                   aligned16(gate) && (!MUL || ((up_stride * es % 16 == 0) && aligned16(up)));
  for (int64_t r0 = 0; r0 < rows; r0 += 65535) {
    const int64_t nr = rows - r0 < 65535 ? rows - r0 : 65535;
    S* o = (S*)out + r0 * n;
    const S* g = (const S*)gate + r0 * gate_stride;
    const S* u = MUL ? (const S*)up + r0 * up_stride : nullptr;
    if (vec) {
      const int nvec = (int)(n / N);
      int gx = (nvec + 255) / 256;
      if (gx > 64) gx = 64;
      hx::launcher(silu_vec_kernel<T, MUL>, dim3(gx, (unsigned)nr), 256, 0, stream)(o, g, u, nvec,
                                                                         gate_stride, up_stride, n);
    } else {
      int64_t gx = (n + 255) / 256;
      if (gx > 64) gx = 64;
      hx::launcher(silu_elem_kernel<T, MUL>, dim3((unsigned)gx, (unsigned)nr), 256, 0, stream)(
          o, g, u, n, gate_stride, up_stride, n);
    }
    int rc = check_launch();
    if (rc) return rc;
  }
  return HX_OK;
}

}  // namespace

extern "C" int hx_rms_norm(void* out, const void* input, const void* weight, float epsilon,
                           int64_t rows, int64_t hidden, int dtype, hx_stream stream) {
  if (rows < 0 || hidden <= 0) return HX_ERR_SHAPE;
  if (rows == 0) return HX_OK;
  if (!out || !input || !weight) return HX_ERR_NULL;
  hipStream_t s = (hipStream_t)stream;
  switch (dtype) {
    case HX_F32: return launch_rms<F32, false>(out, nullptr, input, weight, epsilon, rows, hidden, s);
    case HX_F16: return launch_rms<F16, false>(out, nullptr, input, weight, epsilon, rows, hidden, s);
    case HX_BF16: return launch_rms<BF16, false>(out, nullptr, input, weight, epsilon, rows, hidden, s);
    default: return HX_ERR_DTYPE;
  }
}

extern "C" int hx_add_rms_norm(void* out, void* residual, const void* x, const void* weight,
                               float epsilon, int64_t rows, int64_t hidden, int dtype,
                               hx_stream stream) {
  if (rows < 0 || hidden <= 0) return HX_ERR_SHAPE;
  if (rows == 0) return HX_OK;
  if (!out || !residual || !x || !weight) return HX_ERR_NULL;
  hipStream_t s = (hipStream_t)stream;
  switch (dtype) {
    case HX_F32: return launch_rms<F32, true>(out, residual, x, weight, epsilon, rows, hidden, s);
    case HX_F16: return launch_rms<F16, true>(out, residual, x, weight, epsilon, rows, hidden, s);
    case HX_BF16: return launch_rms<BF16, true>(out, residual, x, weight, epsilon, rows, hidden, s);
    default: return HX_ERR_DTYPE;
  }
}

extern "C" int hx_add_layer_norm(void* out, void* residual, const void* x, const void* weight, const void* bias,
                                 float epsilon, int64_t rows, int64_t hidden, int dtype, hx_stream stream) {
  if (rows < 0 || hidden <= 0) return HX_ERR_SHAPE;
  if (rows == 0) return HX_OK;
  if (!out || !residual || !weight || !bias) return HX_ERR_NULL;
  hipStream_t s = (hipStream_t)stream;
  // x == NULL: plain layer norm of `residual` (which is left untouched)
#define HX_LN_T(T) (x ? launch_layer_norm<T, true>(out, residual, x, weight, bias, epsilon, rows, hidden, s) \
                      : launch_layer_norm<T, false>(out, residual, nullptr, weight, bias, epsilon, rows, hidden, s))
  switch (dtype) {
    case HX_F32: return HX_LN_T(F32);
    case HX_F16: return HX_LN_T(F16);
    case HX_BF16: return HX_LN_T(BF16);
    default: return HX_ERR_DTYPE;
  }
#undef HX_LN_T
}

extern "C" int hx_quick_gelu(void* out, const void* input, int64_t rows, int64_t n, int64_t in_stride, int dtype,
                             hx_stream stream) {
  if (rows < 0 || n < 0) return HX_ERR_SHAPE;
  if (rows == 0 || n == 0) return HX_OK;
  if (!out || !input) return HX_ERR_NULL;
  if (in_stride < n) return HX_ERR_STRIDE;
  hipStream_t s = (hipStream_t)stream;
  switch (dtype) {
    case HX_F32: return launch_quick_gelu<F32>(out, input, rows, n, in_stride, s);
    case HX_F16: return launch_quick_gelu<F16>(out, input, rows, n, in_stride, s);
    case HX_BF16: return launch_quick_gelu<BF16>(out, input, rows, n, in_stride, s);
    default: return HX_ERR_DTYPE;
  }
}

extern "C" int hx_apply_rotary_pos_emb(void* query, void* key, const int32_t* positions,
                                       const void* cos_sin, int64_t n_tokens, int64_t n_heads,
                                       int64_t n_kv_heads, int64_t head_dim, int64_t rotary_dim,
                                       int64_t q_stride, int64_t k_stride, int interleaved,
                                       int dtype, hx_stream stream) {
  if (n_tokens < 0 || n_heads < 0 || n_kv_heads < 0 || head_dim <= 0) return HX_ERR_SHAPE;
  if (rotary_dim <= 0 || rotary_dim > head_dim || (rotary_dim & 1)) return HX_ERR_SHAPE;
  if (n_tokens == 0) return HX_OK;
  if (!query || !key || !positions || !cos_sin) return HX_ERR_NULL;
  if (q_stride < n_heads * head_dim || k_stride < n_kv_heads * head_dim) return HX_ERR_STRIDE;
  hipStream_t s = (hipStream_t)stream;
  switch (dtype) {
    case HX_F32:
      return launch_rope<F32>(query, key, positions, cos_sin, n_tokens, n_heads, n_kv_heads,
                              head_dim, rotary_dim, q_stride, k_stride, interleaved, s);
    case HX_F16:
      return launch_rope<F16>(query, key, positions, cos_sin, n_tokens, n_heads, n_kv_heads,
                              head_dim, rotary_dim, q_stride, k_stride, interleaved, s);
    case HX_BF16:
      return launch_rope<BF16>(query, key, positions, cos_sin, n_tokens, n_heads, n_kv_heads,
                               head_dim, rotary_dim, q_stride, k_stride, interleaved, s);
    default: return HX_ERR_DTYPE;
  }
}

extern "C" int hx_silu(void* out, const void* input, int64_t rows, int64_t n, int64_t in_stride,
                       int dtype, hx_stream stream) {
  if (rows < 0 || n < 0) return HX_ERR_SHAPE;
  if (rows == 0 || n == 0) return HX_OK;
  if (!out || !input) return HX_ERR_NULL;
  if (in_stride < n) return HX_ERR_STRIDE;
  hipStream_t s = (hipStream_t)stream;
  switch (dtype) {
    case HX_F32: return launch_silu<F32, false>(out, input, nullptr, rows, n, in_stride, 0, s);
    case HX_F16: return launch_silu<F16, false>(out, input, nullptr, rows, n, in_stride, 0, s);
    case HX_BF16: return launch_silu<BF16, false>(out, input, nullptr, rows, n, in_stride, 0, s);
    default: return HX_ERR_DTYPE;
  }
}

extern "C" int hx_silu_and_mul(void* out, const void* gate, const void* up, int64_t rows,
                               int64_t n, int64_t gate_stride, int64_t up_stride, int dtype,
                               hx_stream stream) {
  if (rows < 0 || n < 0) return HX_ERR_SHAPE;
  if (rows == 0 || n == 0) return HX_OK;
  if (!out || !gate || !up) return HX_ERR_NULL;
  if (gate_stride < n || up_stride < n) return HX_ERR_STRIDE;
  hipStream_t s = (hipStream_t)stream;
  switch (dtype) {
    case HX_F32: return launch_silu<F32, true>(out, gate, up, rows, n, gate_stride, up_stride, s);
    case HX_F16: return launch_silu<F16, true>(out, gate, up, rows, n, gate_stride, up_stride, s);
    case HX_BF16: return launch_silu<BF16, true>(out, gate, up, rows, n, gate_stride, up_stride, s);
    default: return HX_ERR_DTYPE;
  }
}

extern "C" int hx_rope_set_kv_cache(void* query, void* key, const void* value,
                                    const int32_t* positions, const void* cos_sin,
                                    const int32_t* slot_ids, void* key_cache, void* value_cache,
                                    int64_t n_tokens, int64_t n_heads, int64_t n_kv_heads,
                                    int64_t head_dim, int64_t rotary_dim, int64_t q_stride,
                                    int64_t k_stride, int64_t v_stride, int64_t block_size,
                                    int64_t kcache_block_stride, int64_t vcache_block_stride,
                                    int dtype, hx_stream stream) {
  if (n_tokens < 0 || n_heads < 0 || n_kv_heads <= 0 || head_dim <= 0 || block_size <= 0)
    return HX_ERR_SHAPE;
  if (rotary_dim <= 0 || rotary_dim > head_dim || (rotary_dim & 1)) return HX_ERR_SHAPE;
  if (n_tokens == 0) return HX_OK;
  if (!query || !key || !value || !positions || !cos_sin || !slot_ids || !key_cache || !value_cache)
    return HX_ERR_NULL;
  const int64_t es = dtype_size(dtype);
  if (es == 0) return HX_ERR_DTYPE;
  const int64_t N = 16 / es;
  // vector path only: this op exists for the LLaVA decode/prefill shapes
  if ((rotary_dim / 2) % N || head_dim % N || (head_dim - rotary_dim) % N) return HX_ERR_SHAPE;
  if (q_stride % N || k_stride % N || v_stride % N || kcache_block_stride % N ||
      vcache_block_stride % N)
    return HX_ERR_STRIDE;
  if (!aligned16(query) || !aligned16(key) || !aligned16(value) || !aligned16(cos_sin) ||
      !aligned16(key_cache) || !aligned16(value_cache))
    return HX_ERR_STRIDE;
  if (q_stride < n_heads * head_dim || k_stride < n_kv_heads * head_dim ||
      v_stride < n_kv_heads * head_dim)
    return HX_ERR_STRIDE;
  hipStream_t s = (hipStream_t)stream;
  dim3 grid((unsigned)n_tokens);
#define HX_RC(TT)                                                                              \
  hx::launcher(rope_cache_neox_vec_kernel<TT>, grid, 256, 0, s)(                                         \
      (TT::storage*)query, (TT::storage*)key, (const TT::storage*)value, positions,            \
      (const TT::storage*)cos_sin, slot_ids, (TT::storage*)key_cache,                          \
      (TT::storage*)value_cache, (int)n_heads, (int)n_kv_heads, (int)head_dim, (int)rotary_dim, \
      q_stride, k_stride, v_stride, kcache_block_stride, vcache_block_stride, (int)block_size)
  switch (dtype) {
    case HX_F32: HX_RC(F32); break;
    case HX_F16: HX_RC(F16); break;
    case HX_BF16: HX_RC(BF16); break;
    default: return HX_ERR_DTYPE;
  }
#undef HX_RC
  return check_launch();
}

extern "C" int hx_add_rms_norm_slabs(void* out, void* residual, const float* partial,
                                     int32_t n_splits, const void* weight, float epsilon,
                                     int64_t rows, int64_t hidden, int dtype, hx_stream stream) {
  return hx_add_rms_norm_slabs_ex(out, residual, partial, n_splits, weight, epsilon, rows, hidden, dtype, 0, stream);
}

extern "C" int hx_add_rms_norm_slabs_ex(void* out, void* residual, const float* partial,
                                        int32_t n_splits, const void* weight, float epsilon,
                                        int64_t rows, int64_t hidden, int dtype, int out_fragment_major,
                                        hx_stream stream) {
  if (rows < 0 || hidden <= 0 || n_splits < 1) return HX_ERR_SHAPE;
  if (out_fragment_major && hidden % 32) return HX_ERR_SHAPE;
  const int frag_mb = out_fragment_major ? (int)((rows + 15) / 16) : 0;
  if (rows == 0) return HX_OK;
  if (!out || !residual || !partial || !weight) return HX_ERR_NULL;
  if (hidden % 8 || hidden / 8 > 2048) return HX_ERR_SHAPE;
  if (!aligned16(out) || !aligned16(residual) || !aligned16(partial) || !aligned16(weight))
    return HX_ERR_STRIDE;
  hipStream_t s = (hipStream_t)stream;
  const int64_t stride = rows * hidden;
  dim3 grid((unsigned)rows);
  // 512 threads: one 8-element vector (and its n_splits fp32 slab pieces) per thread up to
  // hidden = 4096 — the kernel is latency-bound on 32 rows, so more loads in flight per row win
#define HX_L(TT, MV)                                                                             \
  hx::launcher(add_rms_norm_slab_kernel<TT, MV, 512>, grid, 512, 0, s)((u16*)out, (u16*)residual, partial, \
                                                            n_splits, stride, (const u16*)weight, \
                                                            epsilon, (int)hidden, frag_mb)
  const int mv = (int)((hidden / 8 + 511) / 512);
  if (dtype == HX_F16) {
    if (mv <= 1) HX_L(F16, 1); else if (mv <= 2) HX_L(F16, 2); else HX_L(F16, 4);
  } else if (dtype == HX_BF16) {
    if (mv <= 1) HX_L(BF16, 1); else if (mv <= 2) HX_L(BF16, 2); else HX_L(BF16, 4);
  } else {
    return HX_ERR_DTYPE;
  }
#undef HX_L
  return check_launch();
}

extern "C" int hx_silu_and_mul_slabs(void* out, const float* partial, int32_t n_splits,
                                     int64_t rows, int64_t inter, int dtype, hx_stream stream) {
  return hx_silu_and_mul_slabs_ex(out, partial, n_splits, rows, inter, dtype, 0, stream);
}

extern "C" int hx_silu_and_mul_slabs_ex(void* out, const float* partial, int32_t n_splits,
                                        int64_t rows, int64_t inter, int dtype, int out_fragment_major,
                                        hx_stream stream) {
  if (rows < 0 || inter <= 0 || n_splits < 1) return HX_ERR_SHAPE;
  if (out_fragment_major && inter % 32) return HX_ERR_SHAPE;
  const int frag_mb = out_fragment_major ? (int)((rows + 15) / 16) : 0;
  if (rows == 0) return HX_OK;
  if (!out || !partial) return HX_ERR_NULL;
  if (inter % 8 || rows > 65535) return HX_ERR_SHAPE;
  if (!aligned16(out) || !aligned16(partial)) return HX_ERR_STRIDE;
  hipStream_t s = (hipStream_t)stream;
  int gx = (int)((inter / 8 + 255) / 256);
  if (gx > 64) gx = 64;
  dim3 grid(gx, (unsigned)rows);
  const int64_t stride = rows * 2 * inter;
  if (dtype == HX_F16)
    hx::launcher(silu_mul_slab_kernel<F16>, grid, 256, 0, s)((u16*)out, partial, n_splits, stride, (int)inter, frag_mb);
  else if (dtype == HX_BF16)
    hx::launcher(silu_mul_slab_kernel<BF16>, grid, 256, 0, s)((u16*)out, partial, n_splits, stride, (int)inter, frag_mb);
  else
    return HX_ERR_DTYPE;
  return check_launch();
}

extern "C" int hx_embed_rms_norm(void* h_out, void* x_out, const void* ids, int ids_are_int64, const void* table,
                                 const void* weight, float epsilon, int64_t rows, int64_t hidden, int64_t vocab,
                                 int dtype, hx_stream stream) {
  if (rows < 0 || hidden <= 0 || vocab <= 0) return HX_ERR_SHAPE;
  if (rows == 0) return HX_OK;
  if (!h_out || !x_out || !ids || !table || !weight) return HX_ERR_NULL;
  if (dtype != HX_F16 && dtype != HX_BF16) return HX_ERR_DTYPE;
  if (hidden % 8 || hidden / 8 > 1024) return HX_ERR_SHAPE;
  if (!aligned16(h_out) || !aligned16(x_out) || !aligned16(table) || !aligned16(weight)) return HX_ERR_STRIDE;
  hipStream_t s = (hipStream_t)stream;
  const int mv = (int)((hidden / 8 + 255) / 256);
#define HX_L(TT, MV)                                                                                      \
  hx::launcher(embed_rms_norm_kernel<TT, MV>, (unsigned)rows, 256, 0, s)((u16*)h_out, (u16*)x_out, ids, ids_are_int64, \
                                                               (const u16*)table, (const u16*)weight, epsilon,  \
                                                               (int)hidden, vocab)
  if (dtype == HX_F16) {
    if (mv <= 1) HX_L(F16, 1); else if (mv <= 2) HX_L(F16, 2); else HX_L(F16, 4);
  } else {
    if (mv <= 1) HX_L(BF16, 1); else if (mv <= 2) HX_L(BF16, 2); else HX_L(BF16, 4);
  }
#undef HX_L
  return check_launch();
}

extern "C" int hx_decode_step_head(const hx_step_head_args* a, hx_stream stream) {
  if (!a) return HX_ERR_NULL;
  if (a->rows <= 0 || a->hidden <= 0 || a->vocab <= 0 || a->zero_bytes < 0 || a->batch < 0) return HX_ERR_SHAPE;
  if (!a->h_out || !a->x_out || !a->ids || !a->table || !a->weight) return HX_ERR_NULL;
  if (a->dtype != HX_F16 && a->dtype != HX_BF16) return HX_ERR_DTYPE;
  if (a->hidden % 8 || a->hidden / 8 > 1024) return HX_ERR_SHAPE;
  if (!aligned16(a->h_out) || !aligned16(a->x_out) || !aligned16(a->table) || !aligned16(a->weight)) return HX_ERR_STRIDE;
  if ((a->feed_src == nullptr) != (a->feed_prev == nullptr)) return HX_ERR_NULL;
  if (a->zero_bytes > 0 && (!a->zero_ptr || (a->zero_bytes & 3) || !aligned16(a->zero_ptr))) return a->zero_ptr ? HX_ERR_STRIDE : HX_ERR_NULL;
  if (a->batch > 0) {
    if (!a->positions || !a->kv_lens || !a->cu_seqlens_k || !a->new_cache_slots || !a->block_table || !a->cu_block_lens)
      return HX_ERR_NULL;
    if (a->block_size <= 0 || a->stride < 1) return HX_ERR_SHAPE;
  }
  StepHeadParams p;
  p.h_out = a->h_out; p.x_out = a->x_out; p.ids = a->ids; p.feed_src = a->feed_src; p.feed_prev = a->feed_prev;
  p.fed_out = a->fed_out; p.table = a->table; p.weight = a->weight; p.zero_ptr = (uint32_t*)a->zero_ptr;
  p.positions = a->positions; p.kv_lens = a->kv_lens; p.cu_seqlens_k = a->cu_seqlens_k; p.new_cache_slots = a->new_cache_slots;
  p.block_table = a->block_table; p.cu_block_lens = a->cu_block_lens; p.rank_desc = a->batch > 0 ? a->rank_desc : nullptr;
  p.vocab = a->vocab; p.zero_words = a->zero_bytes >> 2; p.eps = a->epsilon; p.ids_i64 = a->ids_are_int64 ? 1 : 0;
  p.rows = (int32_t)a->rows; p.hidden = (int32_t)a->hidden; p.n_zero = (int32_t)((p.zero_words + 1023) / 1024);
  p.batch = a->batch; p.block_size = a->block_size; p.stride = a->stride;
  const unsigned grid = (unsigned)(p.rows + p.n_zero + (p.batch > 0 ? 1 : 0));
  hipStream_t s = (hipStream_t)stream;
  const int mv = (int)((a->hidden / 8 + 255) / 256);
#define HX_L(TT, MV) hx::launcher(decode_step_head_kernel<TT, MV>, grid, 256, 0, s)(p)
  if (a->dtype == HX_F16) {
    if (mv <= 1) HX_L(F16, 1); else if (mv <= 2) HX_L(F16, 2); else HX_L(F16, 4);
  } else {
    if (mv <= 1) HX_L(BF16, 1); else if (mv <= 2) HX_L(BF16, 2); else HX_L(BF16, 4);
  }
#undef HX_L
  return check_launch();
}

extern "C" int hx_argmax_rows(int64_t* out, const void* logits, int64_t rows, int64_t n, int64_t ld, int dtype,
                              hx_stream stream) {
  if (rows < 0 || n <= 0 || ld < n || n > 0x7ffffff0) return HX_ERR_SHAPE;
  if (rows == 0) return HX_OK;
  if (!out || !logits) return HX_ERR_NULL;
  hipStream_t s = (hipStream_t)stream;
  if (dtype == HX_F16) hx::launcher(argmax_rows_kernel<F16>, (unsigned)rows, 1024, 0, s)(out, (const u16*)logits, (int)n, ld);
  else if (dtype == HX_BF16) hx::launcher(argmax_rows_kernel<BF16>, (unsigned)rows, 1024, 0, s)(out, (const u16*)logits, (int)n, ld);
  else return HX_ERR_DTYPE;
  return check_launch();
}
