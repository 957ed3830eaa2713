// attn_decode_gqa.hip — paged decode attention for grouped-query models (n_heads > n_kv_heads,
// one new token per sequence).  Same entry as attn_decode.hip (mha_varlen_fwd with q_len == 1,
// flash_api.cpp:216-355 of the reference); chosen by hx_api when group > 1.
//
// Why a second kernel: attn_decode.hip gives every QUERY head its own workgroup, so a KV head is
// streamed `group` times (measured on MI355X, B=32 H=32 ctx 832: 73 us whether HK is 32 or 8 —
// 1.5 TB/s of unique bytes at group 4, 0.8 TB/s at group 7).  Here a workgroup owns one
// (sequence, KV head, key split) and the `group` query heads ride along as the 16 columns of the
// MFMA B operand, so every K / V byte is read once.
//
//   * grid = (n_kv_heads, batch, splits), 4 waves; wave w takes the 32-key tiles w, w+4, ... of
//     the split.  A tile goes HBM -> registers (whole 2D-byte rows, 4 rows per instruction; the
//     next tile's loads are in flight under the current tile's math; page ids are fetched one
//     tile ahead) -> a wave-private LDS image (row stride 2D+32 B) — no workgroup barrier in
//     the loop.
//   * S^T[key][head] = K . Q^T : A = K fragments from the image, B = Q^T of the group's heads
//     (columns >= group are zero).  Softmax per column as in attn_fwd.hip; P^T is the B operand
//     of O^T[dim][head] += V^T . P^T with V^T read by ds_read_b64_tr_b16.  P is rounded to T
//     before P.V like the prefill kernel (flash_fwd_kernel.h:878).
//   * the four waves' (m, l, O) states are merged through LDS; with splits > 1 the workgroup
//     writes per-head partials in the layout attn_decode_combine_kernel reads.
#include "attn_common.h"

namespace hx {
int launch_decode_combine(const AttnParams& p, int batch, int head_dim, int dtype, hipStream_t stream);
}

namespace {

using namespace hx;

constexpr int NW = 4;

template <typename T, int D>
__global__ __launch_bounds__(NW * 64) void attn_decode_gqa_kernel(const AttnParams p) {
  constexpr int NS = D / 32;       // QK k-steps
  constexpr int NDB = D / 16;      // 16-dim output blocks
  constexpr int RS = 2 * D + 32;   // LDS row stride in bytes
  constexpr int LPR = D / 8;       // 16-byte chunks per key row
  constexpr int RPI = 64 / LPR;    // rows covered by one load instruction of the wave
  constexpr int NL = 32 / RPI;     // load instructions per tile (per K and per V)
  constexpr int IMG = 32 * RS;     // one image
  extern __shared__ __attribute__((aligned(16))) char smem[];   // per wave: K image | V image

  const int hk = blockIdx.x, b = blockIdx.y, split = blockIdx.z;
  const int lane = threadIdx.x & 63;
  const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int g = lane >> 4, c = lane & 15;
  const int G = p.group;

  const int kv_len = p.cu_k[b + 1] - p.cu_k[b];
  const int q_row = p.cu_q[b];
  const int n_tiles = (kv_len + 31) >> 5;
  const int per_split = (n_tiles + p.n_splits - 1) / p.n_splits;
  const int t_begin = split * per_split;
  const int t_end = min(n_tiles, t_begin + per_split);
  const int32_t* bt = p.block_table + p.cu_block_lens[b];
  const u16* kbase = reinterpret_cast<const u16*>(p.k) + (int64_t)hk * p.k_head_stride;
  const u16* vbase = reinterpret_cast<const u16*>(p.v) + (int64_t)hk * p.v_head_stride;
  char* kimg = smem + w * (2 * IMG);
  char* vimg = kimg + IMG;

  // tile staging: instruction j covers rows j*RPI + lane/LPR, chunk lane%LPR
  const int lrow = lane / LPR, lchunk = lane % LPR;
  int page_next[NL];
  auto lookup_pages = [&](int t) {
#pragma unroll
    for (int j = 0; j < NL; ++j)
      page_next[j] = bt[page_slot(min(t * 32 + j * RPI + lrow, kv_len - 1), p.block_size, p.block_shift)];
  };
  u16x8 kreg[NL], vreg[NL];
  auto load_tile = [&](int t) {
#pragma unroll
    for (int j = 0; j < NL; ++j) {
      const int row = page_row(min(t * 32 + j * RPI + lrow, kv_len - 1), p.block_size, p.block_shift);
      kreg[j] = __builtin_nontemporal_load(reinterpret_cast<const u16x8*>(
          kbase + (int64_t)page_next[j] * p.k_block_stride + (int64_t)row * p.k_row_stride + 8 * lchunk));
      vreg[j] = __builtin_nontemporal_load(reinterpret_cast<const u16x8*>(
          vbase + (int64_t)page_next[j] * p.v_block_stride + (int64_t)row * p.v_row_stride + 8 * lchunk));
    }
  };
  auto store_tile = [&]() {
#pragma unroll
    for (int j = 0; j < NL; ++j) {
      const int off = (j * RPI + lrow) * RS + lchunk * 16;
      *reinterpret_cast<u16x8*>(kimg + off) = kreg[j];
      *reinterpret_cast<u16x8*>(vimg + off) = vreg[j];
    }
  };

  int t = t_begin + w;
  if (t < t_end) {            // start the HBM stream before touching q
    lookup_pages(t);
    load_tile(t);
    if (t + NW < t_end) lookup_pages(t + NW);
  }

  // Q^T fragments (B operand): lane (c,g) holds Q[head hk*G + c][32s + 8g + j]; columns >= G are zero
  u16x8 qf[NS];
  {
    const u16* qp = reinterpret_cast<const u16*>(p.q) + (int64_t)q_row * p.q_row_stride +
                    (int64_t)(hk * G + min(c, G - 1)) * D + 8 * g;
#pragma unroll
    for (int s = 0; s < NS; ++s) {
      const u16x8 v = *reinterpret_cast<const u16x8*>(qp + 32 * s);
      qf[s] = (c < G) ? v : u16x8{0, 0, 0, 0, 0, 0, 0, 0};
    }
  }

  f32x4 acc[NDB];
#pragma unroll
  for (int i = 0; i < NDB; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  float m = HX_NEG_BIG, l = 0.f;
  const int q4 = c >> 2, p4 = c & 3;

  for (; t < t_end; t += NW) {
    store_tile();                                  // tile t: registers -> this wave's images
    __builtin_amdgcn_wave_barrier();
    if (t + NW < t_end) {
      load_tile(t + NW);                           // next tile in flight under this tile's math
      if (t + 2 * NW < t_end) lookup_pages(t + 2 * NW);
    }
    // ---- S^T = K . Q^T for the two 16-key sub-tiles
    f32x4 s[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      f32x4 a = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int st = 0; st < NS; ++st) {
        const u16x8 kf = *reinterpret_cast<const u16x8*>(kimg + (16 * u + c) * RS + 64 * st + 16 * g);
        a = Mfma<T>::mma(kf, qf[st], a);
      }
      s[u] = a;
    }
    // ---- mask + online softmax per head column (state replicated over g)
    float x[8];
    float mx = HX_NEG_BIG;
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int key = t * 32 + u * 16 + 4 * g + i;
        const float v = (key < kv_len) ? s[u][i] * p.scale_log2 : -INFINITY;
        x[u * 4 + i] = v;
        mx = fmaxf(mx, v);
      }
    mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    const float m_new = fmaxf(m, mx);
    const float alpha = fast_exp2(m - m_new);
    m = m_new;
    u16x8 pf;
    float ps = 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float e = fast_exp2(x[j] - m_new);
      ps += e;
      pf[j] = T::from_float(e);
    }
    l = l * alpha + ps;
    if (__builtin_amdgcn_ballot_w64(alpha != 1.0f)) {
#pragma unroll
      for (int i = 0; i < NDB; ++i) acc[i] *= alpha;
    }
    // ---- O^T += V^T . P^T
    const char* vrd = vimg + (4 * g + q4) * RS + p4 * 8;
#pragma unroll
    for (int db = 0; db < NDB; ++db) {
      const u16x4 lo = lds_tr_read(vrd + db * 32);
      const u16x4 hi = lds_tr_read(vrd + 16 * RS + db * 32);
      u16x8 vf;
      vf[0] = lo[0]; vf[1] = lo[1]; vf[2] = lo[2]; vf[3] = lo[3];
      vf[4] = hi[0]; vf[5] = hi[1]; vf[6] = hi[2]; vf[7] = hi[3];
      acc[db] = Mfma<T>::mma(vf, pf, acc[db]);
    }
    __builtin_amdgcn_wave_barrier();               // image reads done before the next store
  }

  // ---- merge the four waves: state of wave w, head column c at s_m[w][c], s_l[w][c], s_o[w][dim][c]
  l += __shfl_xor(l, 16, 64);
  l += __shfl_xor(l, 32, 64);
  __syncthreads();                                 // every wave is done with its images
  float* s_m = reinterpret_cast<float*>(smem);                 // [NW][16]
  float* s_l = s_m + NW * 16;                                  // [NW][16]
  float* s_o = s_l + NW * 16;                                  // [NW][D][16]
  if (g == 0) {
    s_m[w * 16 + c] = m;
    s_l[w * 16 + c] = l;
  }
#pragma unroll
  for (int db = 0; db < NDB; ++db)
#pragma unroll
    for (int i = 0; i < 4; ++i) s_o[(w * D + 16 * db + 4 * g + i) * 16 + c] = acc[db][i];
  __syncthreads();

  // thread -> (head column hc, dims d0..): 256 threads cover 16 columns x D dims
  const int hc = threadIdx.x & 15;
  if (hc < G) {
    float M = HX_NEG_BIG;
#pragma unroll
    for (int k = 0; k < NW; ++k) M = fmaxf(M, s_m[k * 16 + hc]);
    float wgt[NW], L = 0.f;
#pragma unroll
    for (int k = 0; k < NW; ++k) {
      wgt[k] = fast_exp2(s_m[k * 16 + hc] - M);
      L = fmaf(s_l[k * 16 + hc], wgt[k], L);
    }
    const int h = hk * G + hc;
    const int64_t idx = ((int64_t)b * p.n_heads + h) * p.n_splits + split;
    for (int d = threadIdx.x >> 4; d < D; d += (NW * 64) >> 4) {
      float O = 0.f;
#pragma unroll
      for (int k = 0; k < NW; ++k) O = fmaf(s_o[(k * D + d) * 16 + hc], wgt[k], O);
      if (p.n_splits == 1) {
        reinterpret_cast<u16*>(p.out)[(int64_t)q_row * p.o_row_stride + (int64_t)h * D + d] =
            T::from_float((L > 0.f) ? O / L : 0.f);
      } else {
        p.ws_o[idx * D + d] = O;
      }
    }
    if (p.n_splits > 1 && threadIdx.x < 16) {
      p.ws_ml[idx * 2 + 0] = M;
      p.ws_ml[idx * 2 + 1] = L;
    }
  }
}

template <typename T, int D>
int launch_gqa(const AttnParams& p, int batch, int dtype, hipStream_t stream) {
  constexpr int RS = 2 * D + 32;
  const size_t images = (size_t)NW * 2 * 32 * RS;
  const size_t merge = (size_t)(2 * NW * 16 + NW * D * 16) * sizeof(float);
  const size_t lds = images > merge ? images : merge;
  if (lds > 48 * 1024) {
    hipError_t e = hipFuncSetAttribute((const void*)attn_decode_gqa_kernel<T, D>,
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return hip_rc(e);
  }
  dim3 grid(p.n_heads / p.group, batch, p.n_splits);
  hx::launcher(attn_decode_gqa_kernel<T, D>, grid, NW * 64, lds, stream)(p);
  int rc = check_launch();
  if (rc || p.n_splits == 1) return rc;
  return launch_decode_combine(p, batch, D, dtype, stream);
}

}  // namespace

namespace hx {

bool decode_gqa_supported(int head_dim, int group) {
  return group > 1 && group <= 16 && (head_dim == 64 || head_dim == 128 || head_dim == 256);
}

// splits towards ~512 workgroups (8 waves per CU), but only while every wave keeps at least four
// 32-key tiles: below that the second launch (combine) costs more than the parallelism gains
// (measured, tools/bench_attn_decode_gqa.py: B=32 HK=8 ctx 832 one split 24 us, two 26 us)
int decode_gqa_pick_splits(int batch, int n_kv_heads, int max_seqlen_k, int requested) {
  if (requested >= 1) return requested > 128 ? 128 : requested;
  const int64_t base = (int64_t)batch * n_kv_heads;
  const int n_tiles = (max_seqlen_k + 31) / 32;
  int64_t want = (512 + base - 1) / base;
  int64_t cap = n_tiles / (4 * NW);
  if (cap < 1) cap = 1;
  int64_t s = want < cap ? want : cap;
  if (s > 64) s = 64;
  return (int)s;
}

int launch_attn_decode_gqa(const AttnParams& p, int batch, int head_dim, int dtype, hipStream_t stream) {
#define HX_GQA_CASE(TT, DD) case DD: return launch_gqa<TT, DD>(p, batch, dtype, stream);
  if (dtype == HX_F16) {
    switch (head_dim) { HX_GQA_CASE(F16, 64) HX_GQA_CASE(F16, 128) HX_GQA_CASE(F16, 256) }
  } else if (dtype == HX_BF16) {
    switch (head_dim) { HX_GQA_CASE(BF16, 64) HX_GQA_CASE(BF16, 128) HX_GQA_CASE(BF16, 256) }
  } else {
    return HX_ERR_DTYPE;
  }
#undef HX_GQA_CASE
  return HX_ERR_SHAPE;
}

}  // namespace hx
