// attn_decode4.hip — paged decode attention (q_len == 1) for multi-head models with head_dim 128,
// FOUR query heads per workgroup.  Same operator as attn_decode.hip (replaces the reference's split-KV
// flash kernel run with one query row, csrc/kernel/flash_attn/src/flash_fwd_kernel.h:435-1019), other
// read organisation:
//
//   attn_decode.hip gives every (sequence, head) its own workgroup, so a wave instruction reads 4 key rows
//   x 256 B and the 32 heads of a sequence cover a page from 32 different CUs.  The pattern-only
//   microbenchmark (tools/bench_attn_ceiling.py: the exact addresses, no arithmetic) shows that this
//   organisation tops out at 5.9 TB/s and that the kernel sits on that ceiling; with 4 heads per
//   workgroup — a wave instruction = ONE key row x 1 KiB contiguous (4 heads x 256 B) — the same bytes
//   stream at 6.3-6.4 TB/s.
//
//   * grid = (n_heads / 4, sequence); 8 waves; wave w owns the 8-key tiles w, w + 8, ...
//   * lane = (hq = lane >> 4: head 4*blockIdx.x + hq,  c = lane & 15: dims 8c .. 8c+7 of that head).
//     A tile is 8 K loads + 8 V loads of 16 B per lane (row r of the tile, this lane's 8 dims of its
//     head): every byte is used by exactly the lane that loaded it — no LDS, no MFMA, no transpose.
//   * scores: per row an 8-dim partial dot product per lane (packed 2-element dot instructions) and a
//     4-step DPP butterfly over the 16 lanes of the head; online softmax per head (exp2 domain); P.V is
//     fp32 FMA on the V rows the lane already holds.  P stays fp32 (as in attn_decode.hip).
//   * two tiles (32 KiB per wave) in flight: register double buffer, two waves per SIMD.
//   * FUSE: q and the new token's k / v arrive un-rotated (tensors or the fp32 slabs of the qkv GEMM);
//     RoPE is an exchange with lane c ^ 8 (dims d and d + 64 of one head sit 8 lanes apart), same T
//     rounding as apply_rotary_pos_emb; the new key / value replace the not-yet-written cache row in
//     their tile and are appended to the cache by wave 0.
//   * the four waves' states merge through LDS in wave order.
// STATUS: correct (tests/test_gpu_attention.py::test_decode_four_heads_per_workgroup_experiment, plain and fused
// forms) and OFF (hx_debug_set_option("decode_hpw4", 1) routes eligible launches here).  History of the
// experiment, batch 32 x 32 heads x 720 keys, against 64.8-65.8 us for attn_decode.hip:
//   v1  4 waves x 16-key tiles (two 32 KiB tiles = 256 registers, one wave per SIMD): 64.8-66.6 us — the read
//       pattern alone would give 59.5 us, but the per-tile arithmetic sits on the critical path of the only
//       wave of a SIMD (PMC: VALU active 29 %, waiting 36 % of the wave cycles); halving the instruction count
//       changed nothing;
//   v2  8 waves x 8-key tiles (two waves per SIMD, 170-194 registers), packed fp32 FMAs: 62.8-63.0 us standalone
//       (-4 %), but +-0 inside the decode step (4.876 vs 4.868 ms): the launch is one 512-thread workgroup per
//       CU and opens later than 1024 small workgroups do.
// Both packed dot builtins (v_dot2_f32_f16, v_dot2c_f32_bf16) returned wrong sums in this kernel; plain FMAs.
// Eligible launches: multi-head, head_dim 128, no KV split, grid >= 80 % of the CUs.
#include <cstring>
#include "attn_common.h"

namespace {

using namespace hx;

constexpr int D4 = 128;
constexpr int kDppXor1 = 0xB1;     // quad_perm [1,0,3,2]
constexpr int kDppXor2 = 0x4E;     // quad_perm [2,3,0,1]
constexpr int kDppHalfMirror = 0x141, kDppMirror = 0x140, kDppRor8 = 0x128;

template <int CTRL>
__device__ __forceinline__ float dpp_f(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
}
template <int CTRL>
__device__ __forceinline__ uint32_t dpp_u(uint32_t v) {
  return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, 0xF, 0xF, true);
}
// sum over the 16 lanes of a row (= one head); every lane ends with the total
__device__ __forceinline__ float row16_sum(float v) {
  v += dpp_f<kDppXor1>(v);
  v += dpp_f<kDppXor2>(v);
  v += dpp_f<kDppHalfMirror>(v);
  v += dpp_f<kDppMirror>(v);
  return v;
}

// 8-element partial dot product in fp32: plain FMAs (both packed dot builtins, v_dot2_f32_f16 and
// v_dot2c_f32_bf16, returned wrong sums in this kernel — not pursued).
template <typename T>
struct Dot2 {
  static __device__ __forceinline__ float dot8(u16x8 a, u16x8 b) {
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) s = fmaf(T::to_float(a[i]), T::to_float(b[i]), s);
    return s;
  }
};
constexpr int TR = 8;    // key rows per tile (v2: 8-key tiles, 16 KiB per wave and tile: two waves per SIMD fit)
constexpr int NW4 = 8;   // waves per workgroup
struct Tile4 {
  u16x8 k[TR], v[TR];
};

template <bool NT>
__device__ __forceinline__ u16x8 ld16(const u16* p) {
  if (NT) return __builtin_nontemporal_load(reinterpret_cast<const u16x8*>(p));
  return *reinterpret_cast<const u16x8*>(p);
}

// rows past the sequence's last key are clamped to it (valid >= 1): always in-bounds loads, masked later
__device__ __forceinline__ void load_tile4(Tile4& t, const AttnParams& p, const u16* kb, const u16* vb, int page, int row0,
                                           int valid) {
  const u16* kp = kb + (int64_t)page * p.k_block_stride + (int64_t)row0 * p.k_row_stride;
  const u16* vp = vb + (int64_t)page * p.v_block_stride + (int64_t)row0 * p.v_row_stride;
#pragma unroll
  for (int r = 0; r < TR; ++r) t.k[r] = ld16<true>(kp + (int64_t)min(r, valid - 1) * p.k_row_stride);
#pragma unroll
  for (int r = 0; r < TR; ++r) t.v[r] = ld16<true>(vp + (int64_t)min(r, valid - 1) * p.v_row_stride);
}

// identical rounding to norm_rope_act.hip::rotate_pair / attn_decode.hip::rope_pair
template <typename T>
__device__ __forceinline__ void rope_pair4(float x, float y, float c, float s, float& xo, float& yo) {
#pragma clang fp contract(off)
  const float xc = round_to<T>(x * c), ys = round_to<T>(y * s);
  const float xs = round_to<T>(x * s), yc = round_to<T>(y * c);
  xo = round_to<T>(xc - ys);
  yo = round_to<T>(xs + yc);
}

// NeoX rotation of this lane's 8 dims (8c .. 8c+7 of one head): the partner dims d +- 64 are lane c ^ 8's
template <typename T>
__device__ __forceinline__ u16x8 rope8(u16x8 own, const u16* cs, int c) {
  const u32x4 o32 = __builtin_bit_cast(u32x4, own);
  u32x4 p32;
#pragma unroll
  for (int i = 0; i < 4; ++i) p32[i] = dpp_u<kDppRor8>(o32[i]);
  const u16x8 par = __builtin_bit_cast(u16x8, p32);
  const u16x8 cv = *reinterpret_cast<const u16x8*>(cs + 8 * (c & 7));
  const u16x8 sv = *reinterpret_cast<const u16x8*>(cs + D4 / 2 + 8 * (c & 7));
  const bool low = c < 8;    // this lane holds x (dims < 64); the partner holds y
  u16x8 r;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const float x = T::to_float(low ? own[j] : par[j]), y = T::to_float(low ? par[j] : own[j]);
    float xo, yo;
    rope_pair4<T>(x, y, T::to_float(cv[j]), T::to_float(sv[j]), xo, yo);
    r[j] = T::from_float(low ? xo : yo);
  }
  return r;
}

template <typename T>
__device__ __forceinline__ void compute_tile4(const Tile4& t, u16x8 q, int valid, float scale_log2, float& m, float& l,
                                              float (&o)[8]) {
  float s[TR];
  float mx = m;
#pragma unroll
  for (int r = 0; r < TR; ++r) {
    const float d = row16_sum(Dot2<T>::dot8(t.k[r], q));
    s[r] = r < valid ? d * scale_log2 : -INFINITY;
    mx = fmaxf(mx, s[r]);
  }
  const float alpha = fast_exp2(m - mx);
  float ps = 0.f;
  typedef float f32x2 __attribute__((ext_vector_type(2)));
  f32x2 o2[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) o2[e] = f32x2{o[2 * e] * alpha, o[2 * e + 1] * alpha};
#pragma unroll
  for (int r = 0; r < TR; ++r) {
    const float pr = fast_exp2(s[r] - mx);
    ps += pr;
    const f32x2 p2 = {pr, pr};
#pragma unroll
    for (int e = 0; e < 4; ++e)   // <2 x float> FMAs: v_pk_fma_f32
      o2[e] = __builtin_elementwise_fma(p2, f32x2{T::to_float(t.v[r][2 * e]), T::to_float(t.v[r][2 * e + 1])}, o2[e]);
  }
#pragma unroll
  for (int e = 0; e < 4; ++e) { o[2 * e] = o2[e][0]; o[2 * e + 1] = o2[e][1]; }
  m = mx;
  l = l * alpha + ps;
}

template <typename T, bool FUSE>
__global__ __launch_bounds__(NW4 * 64) void attn_decode4_kernel(const AttnParams p) {
  __shared__ float s_m[NW4][4], s_l[NW4][4];
  __shared__ float s_o[NW4][4][D4];
  const int b = blockIdx.y;
  const int lane = threadIdx.x & 63;
  const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int hq = lane >> 4, c = lane & 15;
  const int h = 4 * blockIdx.x + hq;              // query head == kv head (group 1)

  // FUSE: the new token is the LAST key; it is not read from the cache (its row may not be visible yet) but
  // enters the softmax from registers after the loop, so the loop covers the kv_len - 1 cached keys
  const int kv_len = (p.cu_k[b + 1] - p.cu_k[b]) - (FUSE ? 1 : 0);
  const int q_row = p.cu_q[b];
  const int n_tiles = (kv_len + TR - 1) / TR;
  const int tpp = p.block_size / TR;
  const int32_t* bt = p.block_table + p.cu_block_lens[b];
  // this lane's 16 bytes of a key / value row: head h, dims 8c..
  const u16* kb = reinterpret_cast<const u16*>(p.k) + (int64_t)h * p.k_head_stride + 8 * c;
  const u16* vb = reinterpret_cast<const u16*>(p.v) + (int64_t)h * p.v_head_stride + 8 * c;

  // start the HBM stream first: page ids of this wave's first chunk of 64 tiles, then its first tile
  Tile4 bufA, bufB;
  int my_page = 0, n_my = 0;
  int chunk0 = w;
  auto begin_chunk = [&]() {
    const int tj = chunk0 + NW4 * lane;
    my_page = (tj < n_tiles) ? bt[tj / tpp] : 0;
    n_my = min(64, (n_tiles - chunk0 + NW4 - 1) / NW4);   // wave-uniform
    load_tile4(bufA, p, kb, vb, __builtin_amdgcn_readlane(my_page, 0), (chunk0 % tpp) * TR, kv_len - chunk0 * TR);
  };
  if (chunk0 < n_tiles) begin_chunk();

  // q (and, FUSE, the new token's k / v): this lane's 8 dims of its head
  u16x8 q, kn, vn;
  if (FUSE) {
    // every wave needs q; only wave 0 needs the new token's k / v (it appends them and adds their term)
    kn = vn = u16x8{0, 0, 0, 0, 0, 0, 0, 0};
    auto slab8 = [&](int64_t col) {   // 8 columns of this sequence's qkv row: splits added in order, rounded once to T
      const float* src = p.qkv_partial + (int64_t)b * p.qkv_row + col;
      f32x4 a0 = *reinterpret_cast<const f32x4*>(src), a1 = *reinterpret_cast<const f32x4*>(src + 4);
      for (int s = 1; s < p.qkv_splits; ++s) {
        a0 += *reinterpret_cast<const f32x4*>(src + s * p.qkv_slab_stride);
        a1 += *reinterpret_cast<const f32x4*>(src + 4 + s * p.qkv_slab_stride);
      }
      u16x8 r;
#pragma unroll
      for (int e = 0; e < 4; ++e) { r[e] = T::from_float(a0[e]); r[4 + e] = T::from_float(a1[e]); }
      return r;
    };
    const int64_t hc = (int64_t)h * D4 + 8 * c;
    if (p.qkv_partial) {
      q = slab8(hc);
      if (w == 0) { kn = slab8((int64_t)p.n_heads * D4 + hc); vn = slab8((int64_t)2 * p.n_heads * D4 + hc); }
    } else {
      q = *reinterpret_cast<const u16x8*>(reinterpret_cast<const u16*>(p.q) + (int64_t)q_row * p.q_row_stride + hc);
      if (w == 0) {
        kn = *reinterpret_cast<const u16x8*>(reinterpret_cast<const u16*>(p.k_new) + (int64_t)b * p.kn_row_stride + hc);
        vn = *reinterpret_cast<const u16x8*>(reinterpret_cast<const u16*>(p.v_new) + (int64_t)b * p.vn_row_stride + hc);
      }
    }
    const u16* cs = reinterpret_cast<const u16*>(p.cos_sin) + (int64_t)p.positions[b] * D4;
    q = rope8<T>(q, cs, c);
    if (w == 0) {   // one writer per head: wave 0 rotates the new key and appends the new token to the cache
      kn = rope8<T>(kn, cs, c);
      const int slot = p.new_slots[b];
      const int64_t blk = slot / p.block_size, off = slot % p.block_size;
      *reinterpret_cast<u16x8*>(const_cast<u16*>(kb) + blk * p.k_block_stride + off * p.k_row_stride) = kn;
      *reinterpret_cast<u16x8*>(const_cast<u16*>(vb) + blk * p.v_block_stride + off * p.v_row_stride) = vn;
    }
  } else {
    q = *reinterpret_cast<const u16x8*>(reinterpret_cast<const u16*>(p.q) + (int64_t)q_row * p.q_row_stride + (int64_t)h * D4 + 8 * c);
  }
  float m = HX_NEG_BIG, l = 0.f;
  float o[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) o[e] = 0.f;

  for (bool first = true; chunk0 < n_tiles; chunk0 += NW4 * 64, first = false) {
    if (!first) begin_chunk();
    int j = 0;
    while (j < n_my) {
      if (j + 1 < n_my) {
        const int t = chunk0 + NW4 * (j + 1);
        load_tile4(bufB, p, kb, vb, __builtin_amdgcn_readlane(my_page, j + 1), (t % tpp) * TR, kv_len - t * TR);
      }
      compute_tile4<T>(bufA, q, kv_len - (chunk0 + NW4 * j) * TR, p.scale_log2, m, l, o);
      ++j;
      if (j >= n_my) break;
      if (j + 1 < n_my) {
        const int t = chunk0 + NW4 * (j + 1);
        load_tile4(bufA, p, kb, vb, __builtin_amdgcn_readlane(my_page, j + 1), (t % tpp) * TR, kv_len - t * TR);
      }
      compute_tile4<T>(bufB, q, kv_len - (chunk0 + NW4 * j) * TR, p.scale_log2, m, l, o);
      ++j;
    }
  }

  if (FUSE && w == 0) {   // the new token's key / value, from registers
    const float sn = row16_sum(Dot2<T>::dot8(kn, q)) * p.scale_log2;
    const float mx = fmaxf(m, sn);
    const float alpha = fast_exp2(m - mx), pr = fast_exp2(sn - mx);
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = fmaf(pr, T::to_float(vn[e]), o[e] * alpha);
    l = l * alpha + pr;
    m = mx;
  }
  // merge the waves' states per head (wave order), normalise, store
  if (c == 0) { s_m[w][hq] = m; s_l[w][hq] = l; }
#pragma unroll
  for (int e = 0; e < 8; ++e) s_o[w][hq][8 * c + e] = o[e];
  __syncthreads();
  if (w == 0) {
    float M = HX_NEG_BIG;
#pragma unroll
    for (int k = 0; k < NW4; ++k) M = fmaxf(M, s_m[k][hq]);
    float L = 0.f;
    float O[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) O[e] = 0.f;
#pragma unroll
    for (int k = 0; k < NW4; ++k) {
      const float wgt = fast_exp2(s_m[k][hq] - M);
      L = fmaf(s_l[k][hq], wgt, L);
#pragma unroll
      for (int e = 0; e < 8; ++e) O[e] = fmaf(s_o[k][hq][8 * c + e], wgt, O[e]);
    }
    u16x8 r;
#pragma unroll
    for (int e = 0; e < 8; ++e) r[e] = T::from_float(L > 0.f ? O[e] / L : 0.f);
    *reinterpret_cast<u16x8*>(reinterpret_cast<u16*>(p.out) + (int64_t)q_row * p.o_row_stride + (int64_t)h * D4 + 8 * c) = r;
  }
}

int g_decode4 = 0;   // EXPERIMENT, off: hx_debug_set_option("decode_hpw4", 1) routes eligible launches here (see below)

}  // namespace

namespace hx {

int decode4_set_option(const char* name, int value) {
  if (!strcmp(name, "decode_hpw4")) { g_decode4 = value ? 1 : 0; return HX_OK; }
  return HX_ERR_UNSUPPORTED;
}

// multi-head (group 1), head_dim 128, one KV split, rows of 4 heads contiguous, and a grid that gives every
// CU a workgroup (below that the per-head kernel's 4x finer grid wins)
bool decode4_applies(const AttnParams& p, int batch, int head_dim, int n_cus) {
  return g_decode4 && head_dim == 128 && p.group == 1 && p.n_splits == 1 && p.n_heads % 4 == 0 &&
         p.k_head_stride == 128 && p.v_head_stride == 128 && p.k_row_stride % 8 == 0 && p.v_row_stride % 8 == 0 &&
         (int64_t)batch * (p.n_heads / 4) * 5 >= (int64_t)n_cus * 4;
}

int launch_attn_decode4(const AttnParams& p, int batch, int dtype, hipStream_t stream) {
  const dim3 grid(p.n_heads / 4, batch);
  const bool fuse = p.k_new || p.qkv_partial;
  if (dtype == HX_F16) {
    if (fuse) hx::launcher(attn_decode4_kernel<F16, true>, grid, NW4 * 64, 0, stream)(p);
    else hx::launcher(attn_decode4_kernel<F16, false>, grid, NW4 * 64, 0, stream)(p);
  } else if (dtype == HX_BF16) {
    if (fuse) hx::launcher(attn_decode4_kernel<BF16, true>, grid, NW4 * 64, 0, stream)(p);
    else hx::launcher(attn_decode4_kernel<BF16, false>, grid, NW4 * 64, 0, stream)(p);
  } else {
    return HX_ERR_DTYPE;
  }
  return check_launch();
}

}  // namespace hx
