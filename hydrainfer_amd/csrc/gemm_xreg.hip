// gemm_xreg.hip — decode-batch (M <= 32) weight-streaming GEMM with the ACTIVATIONS IN REGISTERS:
//     partial[s][m][n] = sum_{k in split s} x[m][k] * W[n][k]         (fp32 slabs, usually ONE)
// the nn.Linear products of a decoder layer (hydrainfer/model/llama.py:24-27,48-50) at decode batch
// sizes.  gemm_skinny.hip keeps a 1024-wide slice of x in LDS, so K = 4096 needs 4 K-splits whose
// fp32 partial slabs (25 MB per 7B layer) cost 8.7 us of a 159 us layer in stores alone
// (tools/gemm_ablate.py) and force a reduce kernel behind every product.  Here:
//   * a workgroup is 4 waves and spans the WHOLE K of its split: wave w owns KW consecutive
//     k-steps (of 32) and keeps the x fragments of exactly those k-steps in registers (B operands:
//     KW * MB * 4 registers, 256 for K = 4096 at batch 32; one wave per SIMD, 512 registers each);
//   * the workgroup walks row groups rg = b, b + nb, b + 2 nb, ...; per row group a wave streams its
//     KW KiB of packed weight fragments (1 KiB contiguous per wave instruction, non-temporal)
//     through ceil(KW / 8) register buffers that are refilled for the NEXT row group as soon as they
//     are consumed: up to 32 KiB in flight per wave, 128 KiB per CU, no LDS and no barrier in the
//     loop;
//   * the four waves' partial tiles go to LDS; ONE barrier at the end of the workgroup's life, then
//     the tiles are summed in wave order (deterministic) and stored;
//   * waves start their k-steps at a workgroup-dependent rotation so that the fragments the chip
//     reads at one instant are spread over all memory channels (in lock-step they would sit
//     128 KiB apart).
// K <= 5120 needs no K-split at all (one slab); K = 11008 takes four (was 11), chosen per (N, K) so
// that every CU gets a workgroup (xreg_plan).
// Two extensions of the same kernel finish the operators around the product:
//   EPI = 1   gate|up + silu*mul: the packed gate and up row groups are interleaved, a workgroup owns
//             both halves of its columns and writes act = silu(gate) * up (hx_gate_up_silu_xreg);
//   NORM = 1  the add + RMSNorm that PRODUCES x runs inside the launch: the first M workgroups
//             compute one row each, publish x write-through and count themselves in; all workgroups
//             prefetch weights meanwhile, wait on their XCD's flag line, then load x.  Rows are
//             owned by whoever claims their state word first, and a workgroup that has waited 30 us
//             claims unstarted rows itself: progress needs ONE resident workgroup (hx_norm_*_xreg).
// Activations between these launches are FRAGMENT-MAJOR (the order the B operands are loaded in;
// include/hydra_hip.h): row-major x makes every x load touch 16 cache lines for 16 bytes each.
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <type_traits>
#include "attn_common.h"

namespace {

using namespace hx;

struct XregParams {
  const void* x;
  const void* w;        // packed: fragment (split s, row group rg, k-step j of the split) at KiB index
                        //         s*P*n_rg + rg*nks(s) + j,   P = 4*KW,  nks(s) = min(P, total_ks - s*P)
  float* partial;       // [S][M][N]
  int64_t ldx;
  int32_t M, N, K;
  int32_t stagger;      // bit 0: rotate the k-steps per workgroup; bit 1: test hook, NORM rows only via the rescue path
  int32_t x_packed;     // x is fragment-major: piece ((ks*MB + mb)*64 + lane) = x[16mb + (lane&15)][32ks + 8(lane>>4) ..+8]
  // NORM = 1: x is produced INSIDE the launch — workgroup r < M first computes row r of
  //   residual += (T) sum of the nm_splits slabs;  x = rms_norm(residual) * nm_weight   (hx_add_rms_norm_slabs)
  // and writes it fragment-major to `x` (write-through), the others prefetch weights meanwhile
  const float* nm_partial;   // [nm_splits][M][K]
  void* nm_residual;         // [M][K], in place
  const void* nm_weight;     // [K]
  uint32_t* sync;            // HX_XREG_SYNC_WORDS zeroed words: rows done, error word, one flag line per XCD, row states
  int32_t nm_splits;
  float nm_eps;
  void* act;            // EPI = 1: silu(gate)*up, fragment-major [inter/32][MB][64 lanes][8]
  int32_t interleaved;  // EPI = 0 over a gate|up packing (row groups gate, up, gate, ...): slab columns in [gate | up] order
  int32_t pk_P;         // wide kernel: k-steps per split of the PACKING (the launch's own splits are halves of those)
};

// KERNARG PRELOADING (round 5; Makefile: -mllvm -amdgpu-kernarg-preload-count=16).  A by-value struct parameter is read
// from the kernarg segment by s_loads at the kernel's start — a cold ~0.4 us round trip in front of the first address a
// wave can form, paid at every launch boundary (the math-free null layer got 0.39 us per launch faster with preloaded
// arguments, profiles/r5_kernarg_preload.md).  gfx950 can deliver the first 14 argument dwords in SGPRs WITH the wave,
// but only for scalar / pointer parameters: the kernels therefore take what their first loads need as leading
// scalars and the struct after them; the copy below lets the rest of the code keep reading `p.field`.
// The activations-in-registers kernels go further: NOTHING in front of their first loads comes from the struct.  Five pointer slots
// whose meaning depends on NORM (a norm-fused launch needs the producers' pointers first, x / partial / act late; a plain
// launch the other way round) and four packed integers — 14 dwords:
//   NORM = 0: p0 = w, p1 = x, p2 = partial, p3 = act            NORM = 1: p0 = w, p1 = nm_partial, p2 = nm_residual,
//                                                                          p3 = nm_weight, p4 = sync
//   a = N / 16 | grid.x << 16     b = K / 32 | M << 16 | x_packed << 24 | interleaved << 25
//   c = stagger | nm_splits << 8 | pk_P << 16 (wide kernel)       ldx (elements, < 2^31; NORM = 1: the bits of nm_eps instead)
struct XregHot {
  const void *p0, *p1, *p2, *p3, *p4;
  int32_t a, b, c, ldx;
};
template <int NORM>
inline bool xreg_hot(const XregParams& p, unsigned nb, XregHot* h) {
  if ((p.N >> 4) > 0xffff || nb > 0xffff || (p.K >> 5) > 0xffff || p.M > 255 || p.nm_splits > 255 || p.stagger > 255 ||
      p.pk_P > 0xffff || p.pk_P < 0 || p.ldx > 0x7fffffff || p.ldx < 0)
    return false;
  h->p0 = p.w;
  if (NORM) { h->p1 = p.nm_partial; h->p2 = p.nm_residual; h->p3 = p.nm_weight; h->p4 = p.sync; }
  else { h->p1 = p.x; h->p2 = p.partial; h->p3 = p.act; h->p4 = nullptr; }
  h->a = (int32_t)((uint32_t)(p.N >> 4) | (nb << 16));
  h->b = (int32_t)((uint32_t)(p.K >> 5) | ((uint32_t)p.M << 16) | ((uint32_t)(p.x_packed ? 1 : 0) << 24) | ((uint32_t)(p.interleaved ? 1 : 0) << 25));
  h->c = (int32_t)((uint32_t)p.stagger | ((uint32_t)p.nm_splits << 8) | ((uint32_t)p.pk_P << 16));
  h->ldx = NORM ? __builtin_bit_cast(int32_t, p.nm_eps) : (int32_t)p.ldx;      // (a norm-fused launch reads fragment-major x: no ldx)
  return true;
}

#define HX_XREG_HOT_SIG                                                                                             \
  const void* __restrict__ h_p0, const void* __restrict__ h_p1, const void* __restrict__ h_p2,                        \
      const void* __restrict__ h_p3, const void* __restrict__ h_p4, const int32_t h_a, const int32_t h_b,            \
      const int32_t h_c, const int32_t h_ldx
#define HX_XREG_UNPACK_HOT(NORM_)                                                                                   \
  XregParams p = p_in;                                                                                              \
  p.w = h_p0;                                                                                                       \
  if (NORM_) {                                                                                                      \
    p.nm_partial = reinterpret_cast<const float*>(h_p1); p.nm_residual = const_cast<void*>(h_p2); p.nm_weight = h_p3; \
    p.sync = reinterpret_cast<uint32_t*>(const_cast<void*>(h_p4));                                                  \
  } else {                                                                                                          \
    p.x = h_p1; p.partial = reinterpret_cast<float*>(const_cast<void*>(h_p2)); p.act = const_cast<void*>(h_p3);     \
  }                                                                                                                 \
  p.N = (h_a & 0xffff) << 4; p.K = (h_b & 0xffff) << 5; p.M = (h_b >> 16) & 0xff; p.x_packed = (h_b >> 24) & 1;     \
  p.interleaved = (h_b >> 25) & 1; p.stagger = h_c & 0xff; p.nm_splits = (h_c >> 8) & 0xff;                         \
  p.pk_P = (int32_t)((uint32_t)h_c >> 16);                                                                          \
  if (NORM_) p.nm_eps = __builtin_bit_cast(float, h_ldx);                                                           \
  else p.ldx = h_ldx;                                                                                               \
  const int h_nb = (int)((uint32_t)h_a >> 16);      /* gridDim.x (from the hidden arguments it would be one more s_load) */

__device__ __forceinline__ float silu_f32(float x) { return x / (1.0f + __expf(-x)); }

__device__ __attribute__((aligned(128))) u16 g_zero_line[64] = {0};

static_assert(HX_XREG_SYNC_WORDS >= 320 + 32, "sync area: word 0 arrivals, word 1 error, word 32*(1+xcc) the flag line of that XCD");
constexpr int kXAux = 16;     // cache policy of the x loads behind the hand-over: 16 = sc1 (past the XCD's L2); 0 = plain (L2 hits after
                              // an XCD's first reader) was measured again in round 5: first row group, kernel end and step unchanged — the
                              // 256 KiB per CU are bound by the CU's own 64 B/clk, not by where they come from
constexpr int kStateWord = 320;        // sync area: one ownership word per row (<= 32)
constexpr uint64_t kRescueTicks = 3000;   // 30 us of the 100 MHz clock (a normal wait is ~3 us)
constexpr int kMaxG = 16;

typedef __amdgpu_buffer_rsrc_t rsrc_t;
typedef unsigned int bu32x4 __attribute__((__vector_size__(16)));
__device__ __forceinline__ rsrc_t make_rsrc(const void* p) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, 0x7fffffff, 0x00020000);
}

// hx_add_rms_norm_slabs for ONE row by 256 threads, bit-identical to add_rms_norm_slab_kernel<T, MAXV, 512>:
// thread u plays its virtual threads u and u + 256 (same per-thread sums, same wave sums, same order of
// the eight wave sums).  x goes out fragment-major with sc1 (write-through) stores.
// OWNERSHIP: a row is computed by whoever first exchanges its state word 0 -> 1 (thread 0; the answer
// travels with the reduction barrier, so claiming costs no extra round trip); everybody else returns
// false without having stored anything — the residual is updated exactly once.
template <typename T, int MAXV, int MB, int KB = 6>      // KB: slab pieces requested per round trip (the wide kernel's down hands over 8)
__device__ __forceinline__ bool norm_row_256(const float* __restrict__ partial, int n_splits, int64_t slab_stride,
                                             u16* __restrict__ residual, const u16* __restrict__ weight, float eps,
                                             int hidden, int row, void* x_frag, uint32_t* state, float* red,
                                             int mb_layout = MB) {   // 16-row blocks of the fragment-major x: ceil(M / 16)
  const int tid = threadIdx.x;
  const int nvec = hidden / 8;
  uint32_t claimed = 1;
  if (tid == 0) claimed = __hip_atomic_exchange(state + row, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  u16x8* res_v = reinterpret_cast<u16x8*>(residual + (int64_t)row * hidden);
  const u16x8* w_v = reinterpret_cast<const u16x8*>(weight);
  float x[2 * MAXV][8];
  float ss[2] = {0.f, 0.f};
  u16x8 rr[2 * MAXV], ww[2 * MAXV], hh[2 * MAXV];
#pragma unroll
  for (int v = 0; v < 2; ++v)
#pragma unroll
    for (int j = 0; j < MAXV; ++j) {
      const int i = min(tid + 256 * v + 512 * j, nvec - 1);
      rr[v * MAXV + j] = res_v[i];
      ww[v * MAXV + j] = w_v[i];
    }
#pragma unroll
  for (int v = 0; v < 2; ++v) {
#pragma unroll
    for (int j = 0; j < MAXV; ++j) {
      const int i = tid + 256 * v + 512 * j;
      if (i < nvec) {
        // slab pieces: all loads of a batch of 6 splits before the first add, adds in split order
        // (norm_rope_act.hip slab_sum8)
        const float* pp = partial + (int64_t)row * hidden + i * 8;
        constexpr int kB = KB;
        f32x4 a = {0.f, 0.f, 0.f, 0.f}, b = {0.f, 0.f, 0.f, 0.f};
        for (int s0 = 0; s0 < n_splits; s0 += kB) {
          f32x4 pa[kB], pb[kB];
#pragma unroll
          for (int k = 0; k < kB; ++k) {
            const int sp = min(s0 + k, n_splits - 1);
            pa[k] = *reinterpret_cast<const f32x4*>(pp + sp * slab_stride);
            pb[k] = *reinterpret_cast<const f32x4*>(pp + sp * slab_stride + 4);
          }
#pragma unroll
          for (int k = 0; k < kB; ++k) {
            if (s0 + k == 0) { a = pa[0]; b = pb[0]; }
            else if (s0 + k < n_splits) { a += pa[k]; b += pb[k]; }
          }
        }
        const u16x8 r = rr[v * MAXV + j];
        u16x8 h;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const float sum = round_to<T>(round_to<T>(e < 4 ? a[e] : b[e - 4]) + T::to_float(r[e]));
          x[v * MAXV + j][e] = sum;
          h[e] = T::from_float(sum);
          ss[v] += sum * sum;
        }
        hh[v * MAXV + j] = h;
      }
    }
  }
  const float t0 = wave_sum(ss[0]), t1 = wave_sum(ss[1]);
  if ((tid & 63) == 0) {
    red[tid >> 6] = t0;
    red[4 + (tid >> 6)] = t1;
  }
  if (tid == 0) red[8] = claimed == 0u ? 1.f : 0.f;
  __syncthreads();
  const bool own = red[8] != 0.f;
  float total = 0.f;
#pragma unroll
  for (int k = 0; k < 8; ++k) total += red[k];
  const float inv = rsqrtf(total / (float)hidden + eps);
  if (own) {
    const rsrc_t xrs = make_rsrc(x_frag);      // (built here: x's pointer is not among the preloaded arguments)
#pragma unroll
    for (int v = 0; v < 2; ++v) {
#pragma unroll
      for (int j = 0; j < MAXV; ++j) {
        const int i = tid + 256 * v + 512 * j;
        if (i < nvec) {
          res_v[i] = hh[v * MAXV + j];
          const u16x8 w = ww[v * MAXV + j];
          u16x8 o;
#pragma unroll
          for (int e = 0; e < 8; ++e)
            o[e] = T::from_float(round_to<T>(x[v * MAXV + j][e] * inv) * T::to_float(w[e]));
          // fragment-major piece ((i / 4) * MB + row / 16) * 64 + (i % 4) * 16 + row % 16
          const uint32_t piece = (uint32_t)((i >> 2) * mb_layout + (row >> 4)) * 64 + (i & 3) * 16 + (row & 15);
          __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(bu32x4, o), xrs, piece * 16, 0, 16);
        }
      }
    }
  }
  __syncthreads();   // red[] may be reused by the caller / the next row
  return own;
}

// EPI = 0: fp32 slabs.  EPI = 1 (one split only): the weight is a gate|up projection packed with its
// 16-row groups interleaved (group 2j = gate rows 16j.., group 2j+1 = up rows 16j..); a workgroup
// takes whole pairs and writes act = silu(gate) * up with the rounding of hx_silu_and_mul_slabs on the
// one-slab result (sum -> T, silu -> T, product -> T), fragment-major for the down projection.
template <typename T, int MB, int KW, int EPI = 0, int DBG = 0, int NORM = 0, int RM = 0>
__global__ __launch_bounds__(256) void gemm_xreg_kernel(HX_XREG_HOT_SIG, const XregParams p_in) {
  HX_XREG_UNPACK_HOT(NORM)
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int NBUF = (KW + 7) / 8;
  constexpr int P = 4 * KW;
  const int lane = threadIdx.x & 63;
  const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int g = lane >> 4, c = lane & 15;
  // Work is numbered from the END of the grid: row groups (pairs) are dealt round-robin, so the workgroups that go
  // short in the last, partial round are the FIRST ones of the grid — in a NORM launch the producers of x, which start
  // their own row groups ~3 us after everybody else (tools/xreg_timeline.py: with a full share they were the launch's
  // tail, 32.3 us against ~28 for the others in the 7B gate|up launch).  The same numbering in every form of the
  // kernel: the k-step rotation (stagger) follows it, and with it the summation order of a row group.
  const int split = blockIdx.y, nb = h_nb, b = nb - 1 - (int)blockIdx.x;
  const int total_ks = p.K >> 5;
  const int n_rg = p.N >> 4;
  const int ks0 = split * P;
  const int nks = min(P, total_ks - ks0);
  const int kw = max(0, min(KW, nks - w * KW));            // this wave's real k-steps
  // this workgroup's row groups (>= 1): rg_of(0), rg_of(1), ...
  const int G = EPI ? 2 * (((n_rg >> 1) - b + nb - 1) / nb) : (n_rg - b + nb - 1) / nb;
  auto rg_of = [&](int i) { return EPI ? 2 * (b + (i >> 1) * nb) + (i & 1) : b + i * nb; };
  const int j0 = (p.stagger & 1) ? (int)(((unsigned)b * 7u + (unsigned)w * 3u) % (unsigned)KW) : 0;
  // bit 3 of `stagger` (hx_debug_set_option("xreg_timeline", 1); NORM launches only): the first and the last workgroup
  // of the grid write 100 MHz time stamps of their phases into words 384.. of the sync area (tools/xreg_timeline.py)
  const int flat_id = blockIdx.y * nb + blockIdx.x;
  auto stamp = [&](int k) {
    if (NORM && (p.stagger & 8) && threadIdx.x == 0 && (flat_id == 0 || flat_id == (int)(nb * (NORM ? 1u : gridDim.y)) - 1))
      reinterpret_cast<unsigned long long*>(p.sync + 384 + (flat_id ? 32 : 0))[k] = __builtin_amdgcn_s_memrealtime();
  };
  stamp(0);

  // k-step of slot t: rot(t) = (j0 + t) mod KW; slots whose k-step is past the wave's range are
  // padding: x fragment zero, weight address clamped to a valid fragment
  auto rot = [&](int t) { const int r = j0 + t; return r >= KW ? r - KW : r; };
  // RM = 1: the weight is the ROW-MAJOR [N][K] tensor itself (the one the library prefill GEMMs read: an E/P/EPD node
  // then holds ONE copy).  The fragment of (row group, k-step) is then 16 rows x 64 bytes, lane (c, g) at row c, bytes
  // 16 g: the same elements in the same lanes as the packed fragment, so the sums are bit-identical.
  constexpr int64_t jstride = RM ? 32 : 512;
  const u16* wbase = RM ? reinterpret_cast<const u16*>(p.w) + (int64_t)c * p.K + 8 * g + (int64_t)ks0 * 32
                        : reinterpret_cast<const u16*>(p.w) + 8 * lane + ((int64_t)ks0 * n_rg) * 512;
  auto rg_off = [&](int rg) -> int64_t {
    if constexpr (!RM) return (int64_t)rg * nks * 512;
    else if (EPI || p.interleaved) return ((int64_t)(rg & 1) * (p.N >> 1) + (int64_t)(rg >> 1) * 16) * p.K;
    else return (int64_t)rg * 16 * p.K;
  };
  const int wave_k0 = min(w * KW, max(nks - 1, 0));
  auto frag_ptr = [&](int rg, int t) {
    const int r = rot(t);
    const int j = wave_k0 + (r < kw ? r : 0);
    return wbase + rg_off(rg) + j * jstride;
  };

  u16x8 buf[NBUF][8];
  auto load_buf = [&](int rg, int q) {
#pragma unroll
    for (int j = 0; j < 8; ++j)
      if (8 * q + j < KW) buf[q][j] = __builtin_nontemporal_load(reinterpret_cast<const u16x8*>(frag_ptr(rg, 8 * q + j)));
  };
  // x fragments of this wave's k-steps, in slot order.  Prologue order: for each block of 8 slots,
  // its x fragments, then its weight buffer — the first MFMAs need only the first block, so the
  // x broadcast (256 KiB per CU out of the XCD's L2, ~2 us chip-wide) overlaps the arrival of the
  // first weights instead of preceding it.  No use of a loaded value before the first MFMA
  // (padding slots read a zero line instead of being masked: a select on the loaded value made
  // hipcc wait for every earlier load after each of these).
  u16x8 xb[KW][MB];
  const u16* xp = reinterpret_cast<const u16*>(p.x) + 8 * g;
  const u16* zp = reinterpret_cast<const u16*>(g_zero_line) + 8 * g;
  auto load_x_block = [&](int q) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int t = 8 * q + j;
      if (t < KW) {
        const int r = rot(t);
        const bool ok = r < kw;
        const int ks = min(ks0 + w * KW + r, total_ks - 1);
#pragma unroll
        for (int mb = 0; mb < MB; ++mb) {
          const int m = min(mb * 16 + c, p.M - 1);
          const u16* src = ok ? (p.x_packed ? reinterpret_cast<const u16*>(p.x) + ((int64_t)(ks * MB + mb) * 64 + lane) * 8
                                            : xp + (int64_t)m * p.ldx + (int64_t)ks * 32)
                              : zp;
          if (DBG & 1) xb[t][mb] = u16x8{1, 2, 3, 4, 5, 6, (u16)t, (u16)lane};   // ablation: no x loads
          else xb[t][mb] = *reinterpret_cast<const u16x8*>(src);
        }
      }
    }
  };
  if (NORM) {
    // Sync area (zeroed by the caller): word 0 rows done, word 1 error, word 32*(1+xcc) the flag line of
    // that XCD, words kStateWord.. one ownership word per row.
    uint32_t* st = p.sync + kStateWord;
    float* red = reinterpret_cast<float*>(smem);
    int* cmd = reinterpret_cast<int*>(smem) + 16;
    auto produce = [&](int row) {      // whole workgroup; true if this workgroup computed the row
      const bool own = norm_row_256<T, (KW + 31) / 32, MB>(p.nm_partial, p.nm_splits, (int64_t)p.M * p.K,
                                                           reinterpret_cast<u16*>(p.nm_residual),
                                                           reinterpret_cast<const u16*>(p.nm_weight), p.nm_eps, p.K, row,
                                                           const_cast<void*>(p.x), st, red);
      if (own) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // x is written through: drained = visible
        __syncthreads();
        if (threadIdx.x == 0) {
          const uint32_t old = __hip_atomic_fetch_add(p.sync, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          if (old + 1 == (uint32_t)p.M) {   // last row: one flag line per XCD (a single polled line stalls its channel)
#pragma unroll
            for (int cpy = 0; cpy < 8; ++cpy)
              __hip_atomic_store(p.sync + 32 * (1 + cpy), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          }
        }
      }
      return own;
    };
    // 1. producers: workgroup r < M computes row r of x FIRST (nothing of its own in flight yet: the row's
    //    loads and the store drain are not queued behind weight loads — prefetching first cost the whole gain)
    const int flat = flat_id;
    const int n_wg = nb;      // (a norm-fused launch has ONE split: gridDim.y == 1)
    if (flat < p.M && !(p.stagger & 6))   // (bit 1 of `stagger`: test hook — nobody produces up front, every row is rescued)
      for (int row = flat; row < p.M; row += n_wg) produce(row);   // several rows only when N is tiny
    stamp(1);
    // 2. everyone: weight prefetch (independent of x).  (Round 3, tools/xreg_timeline.py: a producer sees the flag 3 us
    //    later than the others — its first poll returns behind its own prefetch, loads return in order — but letting it
    //    poll first and request its weights afterwards only moves the wait: its first row group ends at 15 us either way.)
#pragma unroll
    for (int q = 0; q < NBUF; ++q) load_buf(rg_of(0), q);
    __builtin_amdgcn_sched_barrier(0);
    // 3. wait for the last row.  Wave 0 polls its XCD's flag line.  RESCUE: a producer workgroup that has
    //    not been dispatched yet cannot be waited for if every CU it could get is held by waiters (two
    //    processes sharing the GPU, each with such a launch: their workgroups interleave per XCD) — so a
    //    workgroup that has waited kRescueTicks looks for a row nobody has claimed and computes it itself.
    //    Progress therefore needs ONE resident workgroup, not the first M.
    for (;;) {
      if (threadIdx.x < 64) {
        const uint32_t* fl = p.sync + 32 * (1 + (__builtin_amdgcn_s_getreg(20 | (3 << 11)) & 7));
        const uint64_t t0 = __builtin_amdgcn_s_memrealtime();
        int c = -1;   // -1: flag seen; >= 0: row to rescue; -2: gave up
        while (!__builtin_amdgcn_readfirstlane(__hip_atomic_load(fl, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))) {
          __builtin_amdgcn_s_sleep(8);
          const uint64_t waited = __builtin_amdgcn_s_memrealtime() - t0;
          if (waited > kRescueTicks && !(p.stagger & 4)) {
            const uint32_t sv = lane < p.M ? __hip_atomic_load(st + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 1u;
            uint64_t free_rows = __ballot(sv == 0u);
            if (free_rows) {
              // rescuers spread over the unclaimed rows (workgroup f takes the (f mod n)-th of them): K missing
              // producers are then rescued side by side, not one after the other through the same lowest row
              for (int skip = flat % __builtin_popcountll(free_rows); skip > 0; --skip) free_rows &= free_rows - 1;
              c = __builtin_ctzll(free_rows);
              break;
            }
          }
          // 1 s at 100 MHz: report, never hang the GPU (bit 2 of `stagger`: test hook — no producers, no rescue,
          // 2 ms bound: every norm-fused launch gives up and must be reported by its caller)
          if (waited > ((p.stagger & 4) ? 200000ull : 100000000ull)) {
            if (threadIdx.x == 0) __hip_atomic_fetch_or(p.sync + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            c = -2;
            break;
          }
        }
        if (threadIdx.x == 0) *cmd = c;
      }
      __syncthreads();
      const int c = *cmd;
      __syncthreads();
      if (c < 0) break;
      produce(c);
    }
    stamp(2);
    asm volatile("" ::: "memory");
    // 4. x: sc1 loads (served past this CU's L1 and the XCD's L2; the bytes were written through by other CUs).
    //    Round 3 measured what that costs — 256 CUs x 256 KiB = 67 MB over the XCD links: with sc0 loads (L2 hits; sound
    //    only by the argument that a kernel's start invalidates the L2) or with no x loads at all the first row group
    //    ends ~1 us earlier (12.7 -> 11.5 us, tools/xreg_timeline.py) and the decode step does not move: not taken.
    {
      const rsrc_t xrs = make_rsrc(p.x), zrs = make_rsrc(g_zero_line);
#pragma unroll
      for (int t = 0; t < KW; ++t) {
        const int r = rot(t);
        const bool ok = r < kw;
        const int ks = min(ks0 + w * KW + r, total_ks - 1);
#pragma unroll
        for (int mb = 0; mb < MB; ++mb) {
          // padding slots (K rounded up to the wave's k-step count) read the zero line
          xb[t][mb] = __builtin_bit_cast(u16x8, __builtin_amdgcn_raw_buffer_load_b128(
              ok ? xrs : zrs, ok ? (uint32_t)((ks * MB + mb) * 64 + lane) * 16 : (uint32_t)(lane & 7) * 16, 0, kXAux));
        }
      }
    }
    __builtin_amdgcn_sched_barrier(0);
  } else {
#pragma unroll
    for (int q = 0; q < NBUF; ++q) {
      load_x_block(q);
      load_buf(rg_of(0), q);
      __builtin_amdgcn_sched_barrier(0);
    }
  }

  f32x4 acc[MB];
#pragma unroll
  for (int mb = 0; mb < MB; ++mb) acc[mb] = f32x4{0.f, 0.f, 0.f, 0.f};
  f32x4* slot = reinterpret_cast<f32x4*>(smem) + (w * MB) * 64 + lane;   // + i*4*MB*64 + mb*64

  auto row_group = [&](int i, auto refill_tag) {
    constexpr bool REFILL = decltype(refill_tag)::value;
    const int rg_next = rg_of(i + 1);
#pragma unroll
    for (int q = 0; q < NBUF; ++q) {
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        if (8 * q + j < KW) {
#pragma unroll
          for (int mb = 0; mb < MB; ++mb) acc[mb] = Mfma<T>::mma(buf[q][j], xb[8 * q + j][mb], acc[mb]);
        }
      }
      // refill behind this buffer's MFMAs (hoisted loads would need fresh registers: spills)
      __builtin_amdgcn_sched_barrier(0);
      if (REFILL) load_buf(rg_next, q);
      __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int mb = 0; mb < MB; ++mb) {
      slot[(i * 4 * MB + mb) * 64] = acc[mb];
      acc[mb] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
  };
  for (int i = 0; i < G - 1; ++i) { row_group(i, std::true_type{}); if (i < 6) stamp(3 + i); }
  row_group(G - 1, std::false_type{});
  stamp(10);
  if ((DBG & 2) && acc[0][0] != 123.25f) return;   // ablation: no reduction, no stores
  __syncthreads();

  // tile (i, mb) is summed by wave (i*MB + mb) % 4 over the four waves in order
  const f32x4* tiles = reinterpret_cast<const f32x4*>(smem) + lane;
  auto tile_sum = [&](int i, int mb) {
    f32x4 s = tiles[((i * 4 + 0) * MB + mb) * 64];
#pragma unroll
    for (int ww = 1; ww < 4; ++ww) s += tiles[((i * 4 + ww) * MB + mb) * 64];
    return s;
  };
  if (EPI == 0) {
    for (int pr = w; pr < G * MB; pr += 4) {
      const int i = pr / MB, mb = pr - i * MB;
      const f32x4 s = tile_sum(i, mb);
      const int m = mb * 16 + c;
      const int rg = rg_of(i);
      const int col = p.interleaved ? ((rg & 1) ? (p.N >> 1) : 0) + ((rg >> 1) << 4) : (rg << 4);
      if (m < p.M) *reinterpret_cast<f32x4*>(p.partial + ((int64_t)split * p.M + m) * p.N + col + 4 * g) = s;
    }
  } else {
    for (int pr = w; pr < (G >> 1) * MB; pr += 4) {
      const int ip = pr / MB, mb = pr - ip * MB;
      const f32x4 gt = tile_sum(2 * ip, mb), up = tile_sum(2 * ip + 1, mb);
      u16x4 r;
#pragma unroll
      for (int e = 0; e < 4; ++e)
        r[e] = T::from_float(round_to<T>(silu_f32(round_to<T>(gt[e]))) * round_to<T>(up[e]));
      // act[m = 16mb + c][k = 16j + 4g + e], j = pair index: piece ((k/32)*MB + mb)*64 + ((k%32)/8)*16 + c, element k%8
      const int k = 16 * (b + ip * nb) + 4 * g;
      u16* dst = reinterpret_cast<u16*>(p.act) + ((((int64_t)(k >> 5) * MB + mb) * 64 + ((k & 31) >> 3) * 16 + c) << 3) + (k & 7);
      *reinterpret_cast<u16x4*>(dst) = r;
    }
  }
  stamp(11);
}

// ------------------------------------------------------------------------------------------------------------
// The same kernel for decode batches of 33 .. 64 rows ("wide": MB = 4).  x for 64 rows is twice the registers, so a
// workgroup spans HALF of a packed K split (KW = half the packing's k-steps per wave: the SAME packed weights serve
// both batch ranges — no second copy) and writes twice the slabs; to keep 32 KiB per wave in flight a work unit is a
// PAIR of 16-row groups (two A fragments per k-step share the four x fragments).  The four waves' tiles are summed
// through LDS after every unit (one barrier per unit, two tile sets used alternately) — a share of up to 11 units
// would not fit LDS at 16 KiB each.  EPI = 0 only (silu*mul needs the whole K: hx_silu_and_mul_slabs behind it);
// NORM as in the 32-row kernel, with up to 64 producers.  Same k-step rotation and summation order per split.
// Round 5: (a) a launch split is HALF of a packing split in the general sense — split s = (packing split s / 2, half
// s % 2), the first half takes P = 4 KW k-steps, the second one what is left of the packing split (LLaVA-1.5-13B's down
// projection: 27 k-steps per wave in the packing -> halves of 14 and 13); (b) RG = row groups per work unit: 2 keeps
// 2 KW KiB per wave in flight; RG = 1 is for KW = 20 (13B's K = 5120: x for 64 rows is 320 registers, a second weight
// buffer set does not fit beside it).
// EPI = 1 (round 5; RG = 2, NORM = 1, a gate|up packing whose split is two halves): BOTH K halves in one workgroup, one
// after the other — x of the first half, all of the workgroup's units (their reduced tiles wait in LDS, 8 KiB per unit), x of
// the second half, the units again, and act = silu(gate) * up leaves from the epilogue with hx_silu_and_mul_slabs' order of
// summation and roundings.  No slabs, no silu*mul launch, nothing between workgroups; the weight stream runs on across
// the seam (the last unit of the first half refills with the first unit of the second).  gridDim.y == 1.
template <typename T, int KW, int NORM, int RG = 2, int EPI = 0>
__global__ __launch_bounds__(256) void gemm_xreg_wide_kernel(HX_XREG_HOT_SIG, const XregParams p_in) {
  HX_XREG_UNPACK_HOT(NORM)
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int MB = 4, NBUF = (KW + 7) / 8, P = 4 * KW;
  static_assert(!EPI || (RG == 2 && NORM == 1), "the two-pass form: (gate, up) pairs, norm-fused");
  const int lane = threadIdx.x & 63;
  const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int g = lane >> 4, c = lane & 15;
  const int split = EPI ? 0 : (int)blockIdx.y, nb = h_nb, b = nb - 1 - (int)blockIdx.x;
  const int total_ks = p.K >> 5;
  const int n_rg = p.N >> 4, n_un = n_rg / RG;
  // where the launch's split sits inside the packing
  const int sp = split >> 1, off_in = (split & 1) * P;
  const int nks_p = min(p.pk_P, total_ks - sp * p.pk_P);
  const int ks0 = sp * p.pk_P + off_in;
  const int nks = max(0, min(P, nks_p - off_in));
  const int kw = max(0, min(KW, nks - w * KW));
  const int G = (n_un - b + nb - 1) / nb;                  // this workgroup's units (>= 1): b, b + nb, ...
  const int mbl = (p.M + 15) >> 4;                         // 16-row blocks of the fragment-major x (3 for 33 .. 48 rows): the
                                                           // fourth MFMA column block then re-reads the third (rows >= M are never stored)
  const int j0 = (p.stagger & 1) ? (int)(((unsigned)b * 7u + (unsigned)w * 3u) % (unsigned)KW) : 0;
  const int flat_id = blockIdx.y * nb + blockIdx.x;
  auto stamp = [&](int k) {      // (xreg_timeline, as in the 32-row kernel: first and last workgroup of a norm-fused launch)
    if (NORM && (p.stagger & 8) && threadIdx.x == 0 && (flat_id == 0 || flat_id == (int)(nb * gridDim.y) - 1))
      reinterpret_cast<unsigned long long*>(p.sync + 384 + (flat_id ? 32 : 0))[k] = __builtin_amdgcn_s_memrealtime();
  };
  stamp(0);
  auto rot = [&](int t) { const int r = j0 + t; return r >= KW ? r - KW : r; };
  const u16* wbase = reinterpret_cast<const u16*>(p.w) + 8 * lane + ((int64_t)sp * p.pk_P * n_rg) * 512;
  const int wave_k0 = off_in + min(w * KW, max(nks - 1, 0));
  // what depends on the K half: first k-step, this wave's k-step count, its first k-step inside the packing split
  struct PassK { int ks0, kw, wk0; };
  const PassK pa{ks0, kw, wave_k0};
  const int nks_b = max(0, min(P, nks_p - P));                       // (EPI: the second half)
  const PassK pb{ks0 + P, max(0, min(KW, nks_b - w * KW)), P + min(w * KW, max(nks_b - 1, 0))};
  auto frag_ptr = [&](int rg, int t, const PassK& ps) {
    const int r = rot(t);
    return wbase + ((int64_t)rg * nks_p + ps.wk0 + (r < ps.kw ? r : 0)) * 512;
  };
  u16x8 buf[RG][NBUF][8];
  auto load_buf = [&](int unit, int q, const PassK& ps) {
#pragma unroll
    for (int h = 0; h < RG; ++h)
#pragma unroll
      for (int j = 0; j < 8; ++j)
        if (8 * q + j < KW)
          buf[h][q][j] = __builtin_nontemporal_load(reinterpret_cast<const u16x8*>(frag_ptr(RG * unit + h, 8 * q + j, ps)));
  };
  auto unit_of = [&](int i) { return b + i * nb; };
  u16x8 xb[KW][MB];
  const u16* xp = reinterpret_cast<const u16*>(p.x) + 8 * g;
  const u16* zp = reinterpret_cast<const u16*>(g_zero_line) + 8 * g;
  auto load_x_block = [&](int q) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int t = 8 * q + j;
      if (t < KW) {
        const int r = rot(t);
        const bool ok = r < kw;
        const int ks = min(ks0 + w * KW + r, total_ks - 1);
#pragma unroll
        for (int mb = 0; mb < MB; ++mb) {
          const int m = min(mb * 16 + c, p.M - 1);
          const u16* src = ok ? (p.x_packed ? reinterpret_cast<const u16*>(p.x) + ((int64_t)(ks * mbl + min(mb, mbl - 1)) * 64 + lane) * 8
                                            : xp + (int64_t)m * p.ldx + (int64_t)ks * 32)
                              : zp;
          xb[t][mb] = *reinterpret_cast<const u16x8*>(src);
        }
      }
    }
  };
  if (NORM) {
    uint32_t* st = p.sync + kStateWord;
    float* red = reinterpret_cast<float*>(smem);
    int* cmd = reinterpret_cast<int*>(smem) + 16;
    auto produce = [&](int row) {
      const bool own = norm_row_256<T, (2 * KW + 31) / 32, MB, 8>(p.nm_partial, p.nm_splits, (int64_t)p.M * p.K,
                                                               reinterpret_cast<u16*>(p.nm_residual),
                                                               reinterpret_cast<const u16*>(p.nm_weight), p.nm_eps, p.K, row,
                                                               const_cast<void*>(p.x), st, red, mbl);
      if (own) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (threadIdx.x == 0) {
          const uint32_t old = __hip_atomic_fetch_add(p.sync, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          if (old + 1 == (uint32_t)p.M) {
#pragma unroll
            for (int cpy = 0; cpy < 8; ++cpy)
              __hip_atomic_store(p.sync + 32 * (1 + cpy), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          }
        }
      }
      return own;
    };
    const int n_wg = nb * (int)gridDim.y;
    if (flat_id < p.M && !(p.stagger & 6))
      for (int row = flat_id; row < p.M; row += n_wg) produce(row);
#pragma unroll
    for (int q = 0; q < NBUF; ++q) load_buf(unit_of(0), q, pa);
    __builtin_amdgcn_sched_barrier(0);
    stamp(1);
    for (;;) {
      if (threadIdx.x < 64) {
        const uint32_t* fl = p.sync + 32 * (1 + (__builtin_amdgcn_s_getreg(20 | (3 << 11)) & 7));
        const uint64_t t0 = __builtin_amdgcn_s_memrealtime();
        int cc = -1;
        while (!__builtin_amdgcn_readfirstlane(__hip_atomic_load(fl, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))) {
          __builtin_amdgcn_s_sleep(8);
          const uint64_t waited = __builtin_amdgcn_s_memrealtime() - t0;
          if (waited > kRescueTicks && !(p.stagger & 4)) {
            const uint32_t sv = lane < p.M ? __hip_atomic_load(st + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 1u;
            uint64_t free_rows = __ballot(sv == 0u);
            if (free_rows) {
              for (int skip = flat_id % __builtin_popcountll(free_rows); skip > 0; --skip) free_rows &= free_rows - 1;
              cc = __builtin_ctzll(free_rows);
              break;
            }
          }
          if (waited > ((p.stagger & 4) ? 200000ull : 100000000ull)) {
            if (threadIdx.x == 0) __hip_atomic_fetch_or(p.sync + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            cc = -2;
            break;
          }
        }
        if (threadIdx.x == 0) *cmd = cc;
      }
      __syncthreads();
      const int cc = *cmd;
      __syncthreads();
      if (cc < 0) break;
      produce(cc);
    }
    stamp(2);
    asm volatile("" ::: "memory");
  }
  // the produced x of one K half (behind the hand-over: kXAux loads)
  auto load_x_norm = [&](const PassK& ps) {
    const rsrc_t xrs = make_rsrc(p.x), zrs = make_rsrc(g_zero_line);
#pragma unroll
    for (int t = 0; t < KW; ++t) {
      const int r = rot(t);
      const bool ok = r < ps.kw;
      const int ks = min(ps.ks0 + w * KW + r, total_ks - 1);
#pragma unroll
      for (int mb = 0; mb < MB; ++mb)
        xb[t][mb] = __builtin_bit_cast(u16x8, __builtin_amdgcn_raw_buffer_load_b128(
            ok ? xrs : zrs, ok ? (uint32_t)((ks * mbl + min(mb, mbl - 1)) * 64 + lane) * 16 : (uint32_t)(lane & 7) * 16, 0, kXAux));
    }
  };
  if (NORM) {
    load_x_norm(pa);
    __builtin_amdgcn_sched_barrier(0);
  } else {
#pragma unroll
    for (int q = 0; q < NBUF; ++q) {
      load_x_block(q);
      load_buf(unit_of(0), q, pa);
      __builtin_amdgcn_sched_barrier(0);
    }
  }

  f32x4 acc[RG][MB];
#pragma unroll
  for (int h = 0; h < RG; ++h)
#pragma unroll
    for (int mb = 0; mb < MB; ++mb) acc[h][mb] = f32x4{0.f, 0.f, 0.f, 0.f};
  // tile (set, row group h of the unit, wave, mb): 1 KiB each
  f32x4* tiles = reinterpret_cast<f32x4*>(smem) + lane;
  // one unit's MFMAs; behind every block of eight k-steps its buffers are requested again for (u_next, ps_next)
  auto mfma_unit = [&](auto refill_tag, int u_next, const PassK& ps_next) {
    constexpr bool REFILL = decltype(refill_tag)::value;
#pragma unroll
    for (int q = 0; q < NBUF; ++q) {
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        if (8 * q + j < KW) {
#pragma unroll
          for (int h = 0; h < RG; ++h)
#pragma unroll
            for (int mb = 0; mb < MB; ++mb) acc[h][mb] = Mfma<T>::mma(buf[h][q][j], xb[8 * q + j][mb], acc[h][mb]);
        }
      }
      __builtin_amdgcn_sched_barrier(0);
      if (REFILL) load_buf(u_next, q, ps_next);
      __builtin_amdgcn_sched_barrier(0);
    }
  };
  // the four waves' tiles of a unit summed in wave order: wave w ends up with tile (h, mb) = ((w + 4 k) >> 2, (w + 4 k) & 3)
  // in sum[k]; tile sets alternate, so one barrier per unit is enough (plain loads in flight survive it)
  auto reduce_unit = [&](int set, f32x4 (&sum)[RG]) {
#pragma unroll
    for (int h = 0; h < RG; ++h)
#pragma unroll
      for (int mb = 0; mb < MB; ++mb) {
        tiles[(((set * RG + h) * 4 + w) * MB + mb) * 64] = acc[h][mb];
        acc[h][mb] = f32x4{0.f, 0.f, 0.f, 0.f};
      }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < RG; ++k) {
      const int tt = w + 4 * k, h = tt >> 2, mb = tt & 3;
      sum[k] = tiles[(((set * RG + h) * 4 + 0) * MB + mb) * 64];
#pragma unroll
      for (int ww = 1; ww < 4; ++ww) sum[k] += tiles[(((set * RG + h) * 4 + ww) * MB + mb) * 64];
    }
  };
  if constexpr (!EPI) {
    auto unit = [&](int i, auto refill_tag) {
      mfma_unit(refill_tag, unit_of(i + 1), pa);
      f32x4 sum[RG];
      reduce_unit(i & 1, sum);
#pragma unroll
      for (int k = 0; k < RG; ++k) {
        const int tt = w + 4 * k, h = tt >> 2, mb = tt & 3;
        const int rg = RG * unit_of(i) + h;
        const int col = p.interleaved ? ((rg & 1) ? (p.N >> 1) : 0) + ((rg >> 1) << 4) : (rg << 4);
        const int m = mb * 16 + c;
        if (m < p.M) *reinterpret_cast<f32x4*>(p.partial + ((int64_t)split * p.M + m) * p.N + col + 4 * g) = sum[k];
      }
    };
    for (int i = 0; i < G - 1; ++i) { unit(i, std::true_type{}); if (i < 6) stamp(3 + i); }
    unit(G - 1, std::false_type{});
  } else {
    // first K half: the units' reduced tiles go to LDS (wave-private pieces: the same wave adds the second half to them)
    f32x4* part = reinterpret_cast<f32x4*>(smem + 2 * RG * 4 * MB * 1024) + lane;
    int n_red = 0;
    for (int i = 0; i < G; ++i) {
      const bool seam = i + 1 == G;
      mfma_unit(std::true_type{}, seam ? unit_of(0) : unit_of(i + 1), seam ? pb : pa);
      f32x4 sum[RG];
      reduce_unit(n_red++ & 1, sum);
#pragma unroll
      for (int k = 0; k < RG; ++k) part[((i * RG + k) * 4 + w) * 64] = sum[k];
      if (i < 3) stamp(3 + i);
    }
    load_x_norm(pb);
    __builtin_amdgcn_sched_barrier(0);
    // second K half, then act[m][16 u + 4 g ..+4] = T(T(silu(T(g0 + g1))) * T(u0 + u1)); wave w holds row block w of both
    // the gate tile (k = 0) and the up tile (k = 1)
    auto finish = [&](int i, auto refill_tag) {
      mfma_unit(refill_tag, unit_of(i + 1), pb);
      f32x4 sum[RG];
      reduce_unit(n_red++ & 1, sum);
      f32x4 gt = part[((i * RG + 0) * 4 + w) * 64], up = part[((i * RG + 1) * 4 + w) * 64];
      gt += sum[0];
      up += sum[1];
      const int m = w * 16 + c;
      if (m < p.M) {
        u16x4 r;
#pragma unroll
        for (int e = 0; e < 4; ++e)
          r[e] = T::from_float(round_to<T>(silu_f32(round_to<T>(gt[e]))) * round_to<T>(up[e]));
        // piece ((k / 32) * mbl + m / 16) * 64 + ((k % 32) / 8) * 16 + m % 16, element k % 8
        const int k = (unit_of(i) << 4) + 4 * g;
        u16* dst = reinterpret_cast<u16*>(p.act) + ((((int64_t)(k >> 5) * mbl + w) * 64 + ((k & 31) >> 3) * 16 + c) << 3) + (k & 7);
        *reinterpret_cast<u16x4*>(dst) = r;
      }
    };
    for (int i = 0; i < G - 1; ++i) { finish(i, std::true_type{}); if (i < 3) stamp(6 + i); }
    finish(G - 1, std::false_type{});
  }
  stamp(11);
}

// output piece i (16 bytes) of the packed tensor <- its source in the row-major weight
__global__ __launch_bounds__(256) void pack_xreg_kernel(u16* __restrict__ packed, const u16* __restrict__ w,
                                                        int64_t n_pieces, int total_ks, int n_rg, int64_t ldw, int P,
                                                        int interleave) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n_pieces) return;
  const int lane = (int)(i & 63);
  const int64_t blk = i >> 6;                                   // KiB index
  const int64_t per_split = (int64_t)P * n_rg;
  const int split = (int)(blk / per_split);
  const int ks0 = split * P;
  const int nks = min(P, total_ks - ks0);
  const int64_t rem = blk - (int64_t)ks0 * n_rg;
  const int64_t rg = rem / nks;
  const int s = ks0 + (int)(rem % nks);
  // interleave: packed group 2j = source group j (gate), 2j+1 = source group n_rg/2 + j (up)
  const int64_t srg = interleave ? (rg >> 1) + ((rg & 1) ? (n_rg >> 1) : 0) : rg;
  const u16* src = w + (16 * srg + (lane & 15)) * ldw + 32 * s + 8 * (lane >> 4);
  *reinterpret_cast<u16x8*>(packed + i * 8) = *reinterpret_cast<const u16x8*>(src);
}

int g_stagger = 1;
int g_no_producers = 0;   // test hook (xreg_no_producers): 1 = the norm-fused launches rely on the rescue path alone; 2 = no producers and no rescue: every such launch gives up (error word) after 2 ms
int g_dbg = 0;
int g_row_major = 0;   // EXPERIMENTS builds (xreg_row_major): every 32-row launch reads its weight argument as the row-major [N][K] tensor
int g_timeline = 0;      // diagnostic (xreg_timeline): phase time stamps of the NORM launches into their sync areas
int g_force_wgs = 0;     // tuning: cap on workgroups per launch (0 = the CU count)

constexpr int kKwSet[] = {4, 8, 16, 20, 22, 27, 29, 32, 40};

int round_kw(int kw) {
  for (int v : kKwSet) if (v >= kw) return v;
  return 0;
}

int n_cus() {
  static int n = [] {
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return 256;
    return prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
  }();
  return n;
}

// (N, K) -> (splits, k-steps per wave).  The fewest splits whose per-wave share fits the register
// budget (KW <= 32), or one more when fewer than 80 % of the CUs would get a workgroup and the extra
// split fills more (one workgroup per CU: K = 11008 with N = 4096 runs 192 workgroups at 3 splits,
// 256 at 4) without padding more than 4 % of the k-steps.
// HX_XREG_S="K:S;K:S" overrides the split count per K (tuning).  Pack and launch both come here.
void xreg_plan(int64_t N, int64_t K, int* S, int* KW, bool fewest_splits = false) {
  const int total_ks = (int)(K >> 5);
  const int n_rg = (int)(N >> 4);
  const int s0 = (total_ks + 159) / 160;   // KW <= 40
  auto kw_of = [&](int s) { return round_kw(((total_ks + s - 1) / s + 3) / 4); };
  auto wgs_of = [&](int s) {
    int nb = n_cus() / s;
    if (nb < 1) nb = 1;
    if (nb > n_rg) nb = n_rg;
    const int G = (n_rg + nb - 1) / nb;
    return s * ((n_rg + G - 1) / G);
  };
  int s = s0;
  if (!fewest_splits) {   // the fused gate|up epilogue needs ONE split: it never takes the extra one
    // padded k-steps (per cent over the real ones) of a split count: what the padding waves re-read
    auto waste_pct = [&](int sc) {
      const int kw = kw_of(sc);
      if (kw <= 0) return 1000;
      const int used = (total_ks + 4 * kw - 1) / (4 * kw);
      return used != sc ? 1000 : (int)(((int64_t)used * 4 * kw * 100) / total_ks) - 100;
    };
    const int w0 = waste_pct(s0), w1 = waste_pct(s0 + 1);
    if (w1 <= 4 && (w0 > 4 ||                                                   // K = 13824: 3 splits pad 11 %, 4 are exact
                    (wgs_of(s0) * 5 < n_cus() * 4 && wgs_of(s0 + 1) > wgs_of(s0))))   // K = 11008, N = 4096: 192 -> 256 workgroups
      s = s0 + 1;
  }
  static const char* env = getenv("HX_XREG_S");
  if (env) {
    const char* q = env;
    while (*q) {
      long k = 0, sv = 0;
      if (sscanf(q, "%ld:%ld", &k, &sv) == 2 && k == K && sv >= s0 && sv <= 16) s = (int)sv;
      const char* semi = strchr(q, ';');
      if (!semi) break;
      q = semi + 1;
    }
  }
  *KW = kw_of(s);
  // the split size is 4*KW k-steps; the number of splits that actually hold data
  *S = *KW > 0 ? (total_ks + 4 * *KW - 1) / (4 * *KW) : 0;
}

template <typename T, int MB, int KW, int EPI, int NORM = 0>
int launch_kw(const XregParams& p, int S, hipStream_t stream) {
  const int n_units = EPI ? (p.N >> 5) : (p.N >> 4);     // row groups, or gate/up pairs of them
  const int per_unit = EPI ? 2 : 1;
  const int cap = g_force_wgs > 0 ? g_force_wgs : n_cus();
  int nb = cap / S;
  if (nb < 1) nb = 1;
  if (nb > n_units) nb = n_units;
  int G = (n_units + nb - 1) / nb;
  if (G * per_unit > kMaxG) {
    // the kernel derives its share from gridDim.x (ceil((n_units - b) / nb) units) and keeps one LDS tile set per
    // row group: a share past kMaxG row groups would index LDS beyond the allocation, so the grid grows instead
    // (more workgroups than CUs; they run in rounds)
    G = kMaxG / per_unit;
    nb = (n_units + G - 1) / G;
  }
  if (((n_units + nb - 1) / nb) * per_unit > kMaxG) return HX_ERR_SHAPE;   // cannot happen: nb >= n_units / G
  // (one workgroup per CU even when the last round is partial: its short shares go to the first workgroups of the
  // grid, which are the late starters of a NORM launch — see the kernel)
  const size_t lds = (size_t)G * per_unit * 4 * MB * 1024;
  if (lds > 48 * 1024) {
    hipError_t e = hipFuncSetAttribute((const void*)gemm_xreg_kernel<T, MB, KW, EPI, 0, NORM>,
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return hip_rc(e);
  }
  const dim3 grid((unsigned)nb, (unsigned)S);
  XregHot h;
  if (!xreg_hot<NORM>(p, (unsigned)nb, &h)) return HX_ERR_SHAPE;
  if (NORM && S != 1) return HX_ERR_SHAPE;
#define HX_XREG_LAUNCH(...) hx::launcher(gemm_xreg_kernel<__VA_ARGS__>, grid, 256, lds, stream)(h.p0, h.p1, h.p2, h.p3, h.p4, h.a, h.b, h.c, h.ldx, p)
  if constexpr (EPI == 0 && NORM == 0 && MB == 2 && (KW == 32 || KW == 29)) if (g_dbg) {   // ablation variants (tools/bench_gemm_xreg.py OPTS=xreg_dbg=..)
    if (g_dbg == 1) HX_XREG_LAUNCH(T, MB, KW, 0, 1);
    else if (g_dbg == 2) HX_XREG_LAUNCH(T, MB, KW, 0, 2);
    else HX_XREG_LAUNCH(T, MB, KW, 0, 3);
    return check_launch();
  }
#if HX_EXPERIMENTS
  // measured and not taken (round 6, profiles/rejected.md): the same kernel over the ROW-MAJOR tensor — one weight copy
  // on a collocated node — streams 16 x 64-byte pieces per wave instruction and is 10-30 % slower (7B, 32 rows)
  if constexpr (MB == 2 && (KW == 32 || KW == 22 || KW == 29)) if (p.stagger & 16) {
    HX_XREG_LAUNCH(T, MB, KW, EPI, 0, NORM, 1);
    return check_launch();
  }
#endif
  HX_XREG_LAUNCH(T, MB, KW, EPI, 0, NORM);
#undef HX_XREG_LAUNCH
  return check_launch();
}

template <typename T, int MB, int EPI, int NORM = 0>
int launch_mb(const XregParams& p, int S, int KW, hipStream_t stream) {
  switch (KW) {
    case 4: return launch_kw<T, MB, 4, EPI, NORM>(p, S, stream);
    case 8: return launch_kw<T, MB, 8, EPI, NORM>(p, S, stream);
    case 16: return launch_kw<T, MB, 16, EPI, NORM>(p, S, stream);
    case 20: if constexpr (EPI == 0 && NORM == 0) return launch_kw<T, MB, 20, 0>(p, S, stream); else return HX_ERR_SHAPE;
    case 22: if constexpr (EPI == 0 && NORM == 0) return launch_kw<T, MB, 22, 0>(p, S, stream); else return HX_ERR_SHAPE;
    case 27: if constexpr (EPI == 0 && NORM == 0) return launch_kw<T, MB, 27, 0>(p, S, stream); else return HX_ERR_SHAPE;
    case 29: if constexpr (EPI == 0 && NORM == 0) return launch_kw<T, MB, 29, 0>(p, S, stream); else return HX_ERR_SHAPE;
    case 32: return launch_kw<T, MB, 32, EPI, NORM>(p, S, stream);
    case 40: return launch_kw<T, MB, 40, EPI, NORM>(p, S, stream);
    default: return HX_ERR_SHAPE;
  }
}

template <int EPI, int NORM = 0>
int launch_any(const XregParams& p, int S, int KW, int dtype, hipStream_t stream) {
  const int MB = (p.M + 15) / 16;
  if (dtype == HX_F16) return MB == 1 ? launch_mb<F16, 1, EPI, NORM>(p, S, KW, stream) : launch_mb<F16, 2, EPI, NORM>(p, S, KW, stream);
  return MB == 1 ? launch_mb<BF16, 1, EPI, NORM>(p, S, KW, stream) : launch_mb<BF16, 2, EPI, NORM>(p, S, KW, stream);
}

// ---- 33 .. 64 rows: the wide kernel over the SAME packing (its splits are halves of the packing's) ----------
// (N, K, how the weight was packed) -> the launch's (splits, k-steps per wave, k-steps per packed split); KW = 0: no
bool wide_plan(int64_t N, int64_t K, bool gate_up_packing, int* S, int* KW, int* pkP, int* S_packed) {
  if (N <= 0 || K <= 0 || N % 32 || K % 32) return false;
  int sp, kwp;
  xreg_plan(N, K, &sp, &kwp, gate_up_packing);
  if (kwp <= 0) return false;
  const int kw = (kwp + 1) / 2;      // an odd packing count halves unevenly: 27 -> 14 + 13 (LLaVA-1.5-13B's down projection)
  if (kw != 2 && kw != 4 && kw != 8 && kw != 10 && kw != 11 && kw != 14 && kw != 16 && kw != 20) return false;   // the built instantiations
  const int total_ks = (int)(K >> 5);
  // launch split s = (packing split s / 2, half s % 2); only the last packing split can be partial, so the splits that
  // hold k-steps are a prefix of that numbering
  int n = 0;
  for (int q = 0; q < sp; ++q) {
    const int nks_p = std::min(4 * kwp, total_ks - q * 4 * kwp);
    n += nks_p > 4 * kw ? 2 : 1;
  }
  *KW = kw; *pkP = 4 * kwp; *S = n; *S_packed = sp;
  return true;
}

constexpr int wide_rg(int kw) { return kw == 20 ? 1 : 2; }      // row groups per work unit (the kernel's RG)

template <typename T, int KW, int NORM, int EPI = 0>
int launch_wide_kw(const XregParams& p, int S, hipStream_t stream) {
  constexpr int RG = wide_rg(KW);
  const int n_units = (int)(p.N >> 4) / RG;
  if (EPI) S = 1;                                    // both K halves in one workgroup
  int nb = n_cus() / S;
  if (nb < 1) nb = 1;
  if (nb > n_units) nb = n_units;
  size_t lds = 2 * RG * 4 * 4 * 1024;                // two tile sets x RG row groups x four waves x MB tiles of 1 KiB
  if (EPI) lds += (size_t)((n_units + nb - 1) / nb) * RG * 4 * 1024;      // + a unit's reduced tiles (first K half) per unit of a workgroup
  if (lds > 150 * 1024) return HX_ERR_SHAPE;
  {   // up to 64 KiB (EPI: + 8 KiB per unit) of dynamic LDS: above the default limit (per device, so not cached in a static)
    hipError_t e = hipFuncSetAttribute((const void*)gemm_xreg_wide_kernel<T, KW, NORM, RG, EPI>,
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return hip_rc(e);
  }
  XregHot h;
  if (!xreg_hot<NORM>(p, (unsigned)nb, &h)) return HX_ERR_SHAPE;
  hx::launcher(gemm_xreg_wide_kernel<T, KW, NORM, RG, EPI>, dim3((unsigned)nb, (unsigned)S), 256, lds, stream)(
      h.p0, h.p1, h.p2, h.p3, h.p4, h.a, h.b, h.c, h.ldx, p);
  return check_launch();
}

template <int NORM>
int launch_wide(const XregParams& p, int S, int KW, int dtype, hipStream_t stream) {
#define HX_W(KWV)                                                                                   \
  case KWV: return dtype == HX_F16 ? launch_wide_kw<F16, KWV, NORM>(p, S, stream) : launch_wide_kw<BF16, KWV, NORM>(p, S, stream)
  switch (KW) {
    HX_W(2); HX_W(4); HX_W(8); HX_W(16); HX_W(20);
    case 14: if constexpr (NORM == 0) return dtype == HX_F16 ? launch_wide_kw<F16, 14, 0>(p, S, stream) : launch_wide_kw<BF16, 14, 0>(p, S, stream); else return HX_ERR_SHAPE;
    case 10: if constexpr (NORM == 0) return dtype == HX_F16 ? launch_wide_kw<F16, 10, 0>(p, S, stream) : launch_wide_kw<BF16, 10, 0>(p, S, stream); else return HX_ERR_SHAPE;
    case 11: if constexpr (NORM == 0) return dtype == HX_F16 ? launch_wide_kw<F16, 11, 0>(p, S, stream) : launch_wide_kw<BF16, 11, 0>(p, S, stream); else return HX_ERR_SHAPE;
    default: return HX_ERR_SHAPE;
  }
#undef HX_W
}

bool xreg_ok(int64_t M, int64_t N, int64_t K) {
  if (M < 1 || M > 64 || N <= 0 || K <= 0 || N % 16 || K % 32) return false;
  int S, KW;
  if (M > 32) {
    int pkP, sp;
    return wide_plan(N, K, false, &S, &KW, &pkP, &sp);
  }
  xreg_plan(N, K, &S, &KW);
  return KW > 0;
}

}  // namespace

namespace hx {
int xreg_set_option(const char* name, int value) {
  if (!strcmp(name, "xreg_stagger")) { g_stagger = value; return HX_OK; }
  if (!strcmp(name, "xreg_timeline")) { g_timeline = value ? 1 : 0; return HX_OK; }
  if (!strcmp(name, "xreg_wgs")) { g_force_wgs = value; return HX_OK; }
  if (!strcmp(name, "xreg_no_producers")) { g_no_producers = value; return HX_OK; }
  if (!strcmp(name, "xreg_dbg")) { g_dbg = value; return HX_OK; }
  if (!strcmp(name, "xreg_row_major")) { g_row_major = value ? 1 : 0; return HX_OK; }
  return HX_ERR_UNSUPPORTED;
}
}  // namespace hx

namespace {
int stagger_bits() {
  return (g_stagger ? 1 : 0) | (g_no_producers == 1 ? 2 : 0) | (g_no_producers == 2 ? 4 : 0) | (g_timeline ? 8 : 0) | (g_row_major ? 16 : 0);
}

// plain product of <= 64 rows over a packing (plain or gate|up-interleaved) with the wide kernel; returns the slab count
int wide_product(float* partial, const void* x, const void* packed, int64_t M, int64_t N, int64_t K, int64_t ldx,
                 int x_fragment_major, bool gate_up_packing, int dtype, hipStream_t stream) {
  int S, KW, pkP, sp;
  if (!wide_plan(N, K, gate_up_packing, &S, &KW, &pkP, &sp)) return HX_ERR_SHAPE;
  XregParams p;
  p.x = x; p.w = packed; p.partial = partial; p.ldx = ldx; p.act = nullptr;
  p.M = (int)M; p.N = (int)N; p.K = (int)K; p.stagger = stagger_bits(); p.x_packed = x_fragment_major ? 1 : 0;
  p.nm_partial = nullptr; p.nm_residual = nullptr; p.nm_weight = nullptr; p.sync = nullptr; p.nm_splits = 0; p.nm_eps = 0.f;
  p.interleaved = gate_up_packing ? 1 : 0; p.pk_P = pkP;
  const int rc = launch_wide<0>(p, S, KW, dtype, stream);
  return rc ? rc : S;
}

// add + RMSNorm fused in front of it (K in ONE packed split)
int wide_norm_product(float* partial, void* residual, const float* slabs_in, int32_t n_splits_in, const void* norm_weight,
                      float epsilon, void* x_frag, const void* packed, int64_t M, int64_t N, int64_t K, void* sync,
                      bool gate_up_packing, int dtype, hipStream_t stream, void* act = nullptr) {
  int S, KW, pkP, sp;
  if (!wide_plan(N, K, gate_up_packing, &S, &KW, &pkP, &sp) || sp != 1 || K % 8 || K / 8 > 1024) return HX_ERR_SHAPE;
  XregParams p;
  p.x = x_frag; p.w = packed; p.partial = partial; p.ldx = K; p.act = act;
  p.M = (int)M; p.N = (int)N; p.K = (int)K; p.stagger = stagger_bits(); p.x_packed = 1;
  p.nm_partial = slabs_in; p.nm_residual = residual; p.nm_weight = norm_weight; p.sync = (uint32_t*)sync;
  p.nm_splits = n_splits_in; p.nm_eps = epsilon; p.interleaved = gate_up_packing ? 1 : 0; p.pk_P = pkP;
  if (act) {      // silu * mul in the launch: both K halves in one workgroup (gemm_xreg_wide_kernel, EPI = 1)
    if (KW != 16 || S != 2 || pkP != 2 * 4 * KW || !gate_up_packing) return HX_ERR_SHAPE;
    return dtype == HX_F16 ? launch_wide_kw<F16, 16, 1, 1>(p, S, stream) : launch_wide_kw<BF16, 16, 1, 1>(p, S, stream);
  }
  const int rc = launch_wide<1>(p, S, KW, dtype, stream);
  return rc ? rc : S;
}
}  // namespace

extern "C" int hx_linear_decode_xreg_supported(int64_t M, int64_t N, int64_t K) { return xreg_ok(M, N, K) ? 1 : 0; }

extern "C" int hx_linear_decode_xreg_splits(int64_t N, int64_t K) {
  if (N <= 0 || N % 16 || K <= 0 || K % 32) return HX_ERR_SHAPE;
  int S, KW;
  xreg_plan(N, K, &S, &KW);
  return KW > 0 ? S : HX_ERR_SHAPE;
}

extern "C" int64_t hx_linear_decode_xreg_workspace_bytes(int64_t M, int64_t N, int64_t K) {
  if (!xreg_ok(M, N, K)) return 0;
  int S, KW;
  if (M > 32) {
    int pkP, sp;
    wide_plan(N, K, false, &S, &KW, &pkP, &sp);
  } else {
    xreg_plan(N, K, &S, &KW);
  }
  return (int64_t)S * M * N * (int64_t)sizeof(float);
}

extern "C" int64_t hx_fragment_major_elems(int64_t rows, int64_t K) {
  if (rows <= 0 || K <= 0 || K % 32) return 0;
  return (rows + 15) / 16 * 16 * K;
}

extern "C" int hx_pack_decode_weight_xreg(void* packed, const void* weight, int64_t N, int64_t K, int64_t ldw,
                                          int interleave_halves, int dtype, hx_stream stream) {
  if (!packed || !weight) return HX_ERR_NULL;
  if (N <= 0 || K <= 0 || N % 16 || K % 32 || ldw % 8) return HX_ERR_SHAPE;
  if (interleave_halves && N % 32) return HX_ERR_SHAPE;
  if (dtype != HX_F16 && dtype != HX_BF16) return HX_ERR_DTYPE;
  if (!aligned16(packed) || !aligned16(weight)) return HX_ERR_STRIDE;
  int S, KW;
  xreg_plan(N, K, &S, &KW, interleave_halves != 0);
  if (KW <= 0) return HX_ERR_SHAPE;
  const int64_t n_pieces = N * K / 8;
  hx::launcher(pack_xreg_kernel, (unsigned)((n_pieces + 255) / 256), 256, 0, (hipStream_t)stream)(
      (u16*)packed, (const u16*)weight, n_pieces, (int)(K >> 5), (int)(N >> 4), ldw, 4 * KW, interleave_halves ? 1 : 0);
  return check_launch();
}

extern "C" int hx_linear_decode_partial_xreg(float* partial, const void* x, const void* packed_weight,
                                             int64_t M, int64_t N, int64_t K, int64_t ldx, int x_fragment_major,
                                             int64_t partial_bytes, int dtype, hx_stream stream) {
  if (!partial || !x || !packed_weight) return HX_ERR_NULL;
  if (!xreg_ok(M, N, K) || (!x_fragment_major && ldx % 8)) return HX_ERR_SHAPE;
  if (dtype != HX_F16 && dtype != HX_BF16) return HX_ERR_DTYPE;
  if (!aligned16(x) || !aligned16(packed_weight) || !aligned16(partial)) return HX_ERR_STRIDE;
  if (partial_bytes < hx_linear_decode_xreg_workspace_bytes(M, N, K)) return HX_ERR_WORKSPACE;
  if (M > 32) return wide_product(partial, x, packed_weight, M, N, K, ldx, x_fragment_major, false, dtype, (hipStream_t)stream);
  int S, KW;
  xreg_plan(N, K, &S, &KW);
  XregParams p;
  p.x = x; p.w = packed_weight; p.partial = partial; p.ldx = ldx; p.act = nullptr;
  p.M = (int)M; p.N = (int)N; p.K = (int)K; p.stagger = stagger_bits(); p.x_packed = x_fragment_major ? 1 : 0;
  p.nm_partial = nullptr; p.nm_residual = nullptr; p.nm_weight = nullptr; p.sync = nullptr; p.nm_splits = 0; p.nm_eps = 0.f; p.interleaved = 0; p.pk_P = 0;
  const int rc = launch_any<0>(p, S, KW, dtype, (hipStream_t)stream);
  return rc ? rc : S;
}

extern "C" int hx_gate_up_silu_xreg_supported(int64_t M, int64_t inter, int64_t K) {
  if (M > 32 || !xreg_ok(M, 2 * inter, K) || inter % 32) return 0;   // (the fused epilogue needs the whole K in the workgroup)
  int S, KW;
  xreg_plan(2 * inter, K, &S, &KW, true);
  return S == 1 && (KW == 4 || KW == 8 || KW == 16 || KW == 32 || KW == 40) ? 1 : 0;   // the fused epilogue is built for these
}

extern "C" int hx_gate_up_silu_xreg(void* act, const void* x, const void* packed_gate_up, int64_t M,
                                    int64_t inter, int64_t K, int64_t ldx, int x_fragment_major, int dtype,
                                    hx_stream stream) {
  if (!act || !x || !packed_gate_up) return HX_ERR_NULL;
  if (!hx_gate_up_silu_xreg_supported(M, inter, K) || (!x_fragment_major && ldx % 8)) return HX_ERR_SHAPE;
  if (dtype != HX_F16 && dtype != HX_BF16) return HX_ERR_DTYPE;
  if (!aligned16(x) || !aligned16(packed_gate_up) || !aligned16(act)) return HX_ERR_STRIDE;
  int S, KW;
  xreg_plan(2 * inter, K, &S, &KW, true);
  XregParams p;
  p.x = x; p.w = packed_gate_up; p.partial = nullptr; p.ldx = ldx; p.act = act;
  p.M = (int)M; p.N = (int)(2 * inter); p.K = (int)K; p.stagger = stagger_bits(); p.x_packed = x_fragment_major ? 1 : 0;
  p.nm_partial = nullptr; p.nm_residual = nullptr; p.nm_weight = nullptr; p.sync = nullptr; p.nm_splits = 0; p.nm_eps = 0.f; p.interleaved = 0; p.pk_P = 0;
  return launch_any<1>(p, 1, KW, dtype, (hipStream_t)stream);
}

// ---- add + RMSNorm fused in front of the product (one launch instead of two) --------------------
static bool norm_kw_ok(int S, int KW) { return S == 1 && (KW == 4 || KW == 8 || KW == 16 || KW == 32 || KW == 40); }

extern "C" int hx_norm_xreg_supported(int64_t M, int64_t N, int64_t K, int gate_up) {
  if (gate_up) return hx_gate_up_silu_xreg_supported(M, N / 2, K);   // same condition: one split, a built k-step count
  if (!xreg_ok(M, N, K)) return 0;
  if (M > 32) {          // wide kernel: K must sit in ONE split of the packing, rows of <= 4096 x 2 elements
    int S, KW, pkP, sp;
    return wide_plan(N, K, false, &S, &KW, &pkP, &sp) && sp == 1 && K % 8 == 0 && K / 8 <= 1024 && KW != 10 && KW != 11 && KW != 14 ? 1 : 0;
  }
  int S, KW;
  xreg_plan(N, K, &S, &KW);
  return norm_kw_ok(S, KW) && K % 8 == 0 && K / 8 <= 2048 ? 1 : 0;
}

static int norm_args_ok(const void* residual, const float* slabs, int32_t n_splits, const void* weight, const void* x_frag,
                        const void* sync) {
  if (!residual || !slabs || !weight || !x_frag || !sync) return HX_ERR_NULL;
  if (n_splits < 1) return HX_ERR_SHAPE;
  if (!aligned16(residual) || !aligned16(slabs) || !aligned16(weight) || !aligned16(x_frag) || !aligned16(sync)) return HX_ERR_STRIDE;
  return HX_OK;
}

extern "C" int hx_norm_linear_decode_xreg(float* partial, void* residual, const float* slabs_in, int32_t n_splits_in,
                                          const void* norm_weight, float epsilon, void* x_frag,
                                          const void* packed_weight, int64_t M, int64_t N, int64_t K, void* sync,
                                          int64_t partial_bytes, int dtype, hx_stream stream) {
  if (!partial || !packed_weight) return HX_ERR_NULL;
  int rc = norm_args_ok(residual, slabs_in, n_splits_in, norm_weight, x_frag, sync);
  if (rc) return rc;
  if (!hx_norm_xreg_supported(M, N, K, 0)) return HX_ERR_SHAPE;
  if (dtype != HX_F16 && dtype != HX_BF16) return HX_ERR_DTYPE;
  if (!aligned16(packed_weight) || !aligned16(partial)) return HX_ERR_STRIDE;
  if (partial_bytes < hx_linear_decode_xreg_workspace_bytes(M, N, K)) return HX_ERR_WORKSPACE;
  if (M > 32) return wide_norm_product(partial, residual, slabs_in, n_splits_in, norm_weight, epsilon, x_frag, packed_weight, M, N, K,
                                       sync, false, dtype, (hipStream_t)stream);
  int S, KW;
  xreg_plan(N, K, &S, &KW);
  XregParams p;
  p.x = x_frag; p.w = packed_weight; p.partial = partial; p.ldx = K; p.act = nullptr;
  p.M = (int)M; p.N = (int)N; p.K = (int)K; p.stagger = stagger_bits(); p.x_packed = 1;
  p.nm_partial = slabs_in; p.nm_residual = residual; p.nm_weight = norm_weight; p.sync = (uint32_t*)sync;
  p.nm_splits = n_splits_in; p.nm_eps = epsilon; p.interleaved = 0; p.pk_P = 0;
  rc = launch_any<0, 1>(p, S, KW, dtype, (hipStream_t)stream);
  return rc ? rc : S;
}

extern "C" int hx_norm_gate_up_silu_xreg(void* act, void* residual, const float* slabs_in, int32_t n_splits_in,
                                         const void* norm_weight, float epsilon, void* x_frag,
                                         const void* packed_gate_up, int64_t M, int64_t inter, int64_t K, void* sync,
                                         int dtype, hx_stream stream) {
  if (!act || !packed_gate_up) return HX_ERR_NULL;
  int rc = norm_args_ok(residual, slabs_in, n_splits_in, norm_weight, x_frag, sync);
  if (rc) return rc;
  if (!hx_norm_xreg_supported(M, 2 * inter, K, 1)) return HX_ERR_SHAPE;
  if (dtype != HX_F16 && dtype != HX_BF16) return HX_ERR_DTYPE;
  if (!aligned16(packed_gate_up) || !aligned16(act)) return HX_ERR_STRIDE;
  int S, KW;
  xreg_plan(2 * inter, K, &S, &KW, true);
  XregParams p;
  p.x = x_frag; p.w = packed_gate_up; p.partial = nullptr; p.ldx = K; p.act = act;
  p.M = (int)M; p.N = (int)(2 * inter); p.K = (int)K; p.stagger = stagger_bits(); p.x_packed = 1;
  p.nm_partial = slabs_in; p.nm_residual = residual; p.nm_weight = norm_weight; p.sync = (uint32_t*)sync;
  p.nm_splits = n_splits_in; p.nm_eps = epsilon; p.interleaved = 0; p.pk_P = 0;
  return launch_any<1, 1>(p, 1, KW, dtype, (hipStream_t)stream);
}

// ---- the gate|up product WITHOUT the fused silu*mul, over the interleaved packing (batches of 33 .. 64 rows: x for
// 64 rows needs two K splits per workgroup, silu*mul the whole K) — slabs in [gate | up] column order for
// hx_silu_and_mul_slabs.  Optionally with the add + RMSNorm in front.
extern "C" int hx_gate_up_xreg_supported(int64_t M, int64_t inter, int64_t K, int with_norm) {
  if (M < 1 || M > 64 || inter <= 0 || inter % 32 || K <= 0 || K % 32) return 0;
  int S, KW, pkP, sp;
  if (!wide_plan(2 * inter, K, true, &S, &KW, &pkP, &sp)) return 0;
  if (with_norm && (sp != 1 || K % 8 || K / 8 > 1024 || KW == 10 || KW == 11 || KW == 14)) return 0;
  return 1;
}

extern "C" int64_t hx_gate_up_xreg_workspace_bytes(int64_t M, int64_t inter, int64_t K) {
  int S, KW, pkP, sp;
  if (M < 1 || M > 64 || !wide_plan(2 * inter, K, true, &S, &KW, &pkP, &sp)) return 0;
  return (int64_t)S * M * 2 * inter * (int64_t)sizeof(float);
}

extern "C" int hx_gate_up_xreg(float* partial, const void* x, const void* packed_gate_up, int64_t M, int64_t inter, int64_t K,
                               int64_t ldx, int x_fragment_major, int64_t partial_bytes, int dtype, hx_stream stream) {
  if (!partial || !x || !packed_gate_up) return HX_ERR_NULL;
  if (!hx_gate_up_xreg_supported(M, inter, K, 0) || (!x_fragment_major && ldx % 8)) return HX_ERR_SHAPE;
  if (dtype != HX_F16 && dtype != HX_BF16) return HX_ERR_DTYPE;
  if (!aligned16(x) || !aligned16(packed_gate_up) || !aligned16(partial)) return HX_ERR_STRIDE;
  if (partial_bytes < hx_gate_up_xreg_workspace_bytes(M, inter, K)) return HX_ERR_WORKSPACE;
  return wide_product(partial, x, packed_gate_up, M, 2 * inter, K, ldx, x_fragment_major, true, dtype, (hipStream_t)stream);
}

extern "C" int hx_gate_up_silu_wide_xreg_supported(int64_t M, int64_t inter, int64_t K) {
  if (M < 33 || M > 64 || !hx_gate_up_xreg_supported(M, inter, K, 1)) return 0;
  int S, KW, pkP, sp;
  if (!wide_plan(2 * inter, K, true, &S, &KW, &pkP, &sp)) return 0;
  if (!(KW == 16 && S == 2 && sp == 1 && pkP == 2 * 4 * KW)) return 0;      // one packing split of two full halves (K = 4096)
  // the launch's LDS (launch_wide_kw, EPI = 1: 64 KiB + 8 KiB per unit of a workgroup) depends on the CU count of the
  // device: on a smaller or partitioned one the units per workgroup grow — answer what the launch itself would (round-5
  // ADVICE: the plan said yes, the launch HX_ERR_SHAPE, and the 33 .. 64-row step had no other path)
  constexpr int RG = wide_rg(16);
  const int n_units = (int)((2 * inter) >> 4) / RG;
  const int nb = std::max(1, std::min(n_cus(), n_units));
  const size_t lds = (size_t)2 * RG * 4 * 4 * 1024 + (size_t)((n_units + nb - 1) / nb) * RG * 4 * 1024;
  return lds <= 150 * 1024 ? 1 : 0;
}

extern "C" int hx_norm_gate_up_silu_wide_xreg(void* act, void* residual, const float* slabs_in, int32_t n_splits_in,
                                              const void* norm_weight, float epsilon, void* x_frag,
                                              const void* packed_gate_up, int64_t M, int64_t inter, int64_t K, void* sync,
                                              int dtype, hx_stream stream) {
  if (!act || !packed_gate_up) return HX_ERR_NULL;
  int rc = norm_args_ok(residual, slabs_in, n_splits_in, norm_weight, x_frag, sync);
  if (rc) return rc;
  if (!hx_gate_up_silu_wide_xreg_supported(M, inter, K)) return HX_ERR_SHAPE;
  if (dtype != HX_F16 && dtype != HX_BF16) return HX_ERR_DTYPE;
  if (!aligned16(packed_gate_up) || !aligned16(act)) return HX_ERR_STRIDE;
  return wide_norm_product(nullptr, residual, slabs_in, n_splits_in, norm_weight, epsilon, x_frag, packed_gate_up, M, 2 * inter, K,
                           sync, true, dtype, (hipStream_t)stream, act);
}

extern "C" int hx_norm_gate_up_xreg(float* partial, void* residual, const float* slabs_in, int32_t n_splits_in,
                                    const void* norm_weight, float epsilon, void* x_frag, const void* packed_gate_up,
                                    int64_t M, int64_t inter, int64_t K, void* sync, int64_t partial_bytes, int dtype,
                                    hx_stream stream) {
  if (!partial || !packed_gate_up) return HX_ERR_NULL;
  int rc = norm_args_ok(residual, slabs_in, n_splits_in, norm_weight, x_frag, sync);
  if (rc) return rc;
  if (!hx_gate_up_xreg_supported(M, inter, K, 1)) return HX_ERR_SHAPE;
  if (dtype != HX_F16 && dtype != HX_BF16) return HX_ERR_DTYPE;
  if (!aligned16(packed_gate_up) || !aligned16(partial)) return HX_ERR_STRIDE;
  if (partial_bytes < hx_gate_up_xreg_workspace_bytes(M, inter, K)) return HX_ERR_WORKSPACE;
  return wide_norm_product(partial, residual, slabs_in, n_splits_in, norm_weight, epsilon, x_frag, packed_gate_up, M, 2 * inter, K,
                           sync, true, dtype, (hipStream_t)stream);
}

// ------------------------------------------------------------------------------------------------------------
// One entry for the decode-batch linear layer (include/hydra_hip.h "hx_decode_weight"): the caller describes the
// weight once, the library chooses the layout (activations-in-registers for <= 32 rows when the shape allows it,
// LDS-slice otherwise), packs it, and hx_linear_decode_ex dispatches on the descriptor.
// ------------------------------------------------------------------------------------------------------------
extern "C" int hx_decode_weight_plan(hx_decode_weight* w, int64_t N, int64_t K, int dtype, int max_rows, int flags) {
  if (!w) return HX_ERR_NULL;
  if (N <= 0 || K <= 0 || max_rows <= 0 || max_rows > 64) return HX_ERR_SHAPE;
  if (dtype != HX_F16 && dtype != HX_BF16) return HX_ERR_DTYPE;
  if (flags & ~(HX_DW_GATE_UP | HX_DW_FORCE_LDS_SLICE)) return HX_ERR_UNSUPPORTED;
  const bool force_lds = (flags & HX_DW_FORCE_LDS_SLICE) != 0;
  flags &= ~HX_DW_FORCE_LDS_SLICE;
  w->packed = nullptr;
  w->N = N; w->K = K; w->dtype = dtype; w->flags = flags; w->max_rows = max_rows;
  // the activations-in-registers layout serves <= 32 rows always, 33 .. 64 rows where the wide kernel can read the same
  // packing (a gate|up weight keeps its halves interleaved then: the plain product un-interleaves its slab columns)
  bool xreg = !force_lds && hx_linear_decode_xreg_supported(max_rows < 32 ? max_rows : 32, N, K) == 1;
  if (xreg && max_rows > 32) {
    int S, KW, pkP, sp;
    const bool gu = (flags & HX_DW_GATE_UP) && N % 32 == 0 && hx_gate_up_silu_xreg_supported(32, N / 2, K) == 1;
    xreg = wide_plan(N, K, gu, &S, &KW, &pkP, &sp);
  }
  if (!xreg && (N % 16 || K % 256)) return HX_ERR_SHAPE;
  if ((flags & HX_DW_GATE_UP) && (!xreg || N % 32)) w->flags &= ~HX_DW_GATE_UP;   // no fused epilogue on this layout: halves stay in order
  w->layout = xreg ? HX_DW_XREG : HX_DW_LDS_SLICE;
  return HX_OK;
}

extern "C" int hx_decode_weight_pack(hx_decode_weight* w, void* packed, const void* weight, int64_t ldw, hx_stream stream) {
  if (!w || !packed || !weight) return HX_ERR_NULL;
  int rc;
  if (w->layout == HX_DW_XREG) {
    const int inter = (w->flags & HX_DW_GATE_UP) && hx_gate_up_silu_xreg_supported(w->max_rows < 32 ? w->max_rows : 32, w->N / 2, w->K) == 1;
    if (!inter) w->flags &= ~HX_DW_GATE_UP;
    rc = hx_pack_decode_weight_xreg(packed, weight, w->N, w->K, ldw, inter, w->dtype, stream);
  } else if (w->layout == HX_DW_LDS_SLICE) {
    rc = hx_pack_decode_weight(packed, weight, w->N, w->K, ldw, w->dtype, stream);
  } else {
    return HX_ERR_UNSUPPORTED;
  }
  if (rc == HX_OK) w->packed = packed;
  return rc;
}

extern "C" int64_t hx_linear_decode_ex_workspace_bytes(const hx_decode_weight* w, int64_t M) {
  if (!w || M <= 0) return 0;
  if (w->layout == HX_DW_XREG && (w->flags & HX_DW_GATE_UP)) return hx_gate_up_xreg_workspace_bytes(M, w->N / 2, w->K);
  return w->layout == HX_DW_XREG ? hx_linear_decode_xreg_workspace_bytes(M, w->N, w->K)
                                 : hx_linear_decode_workspace_bytes(M, w->N, w->K);
}

extern "C" int hx_linear_decode_ex(float* partial, int64_t partial_bytes, const void* x, int64_t ldx,
                                   int x_fragment_major, const hx_decode_weight* w, int64_t M, hx_stream stream) {
  if (!w || !w->packed) return HX_ERR_NULL;
  if (M > w->max_rows) return HX_ERR_SHAPE;
  if (w->layout == HX_DW_XREG) {
    if (w->flags & HX_DW_GATE_UP)     // interleaved halves: the fused hx_gate_up_silu_xreg, or this plain product (slabs [gate | up])
      return hx_gate_up_xreg(partial, x, w->packed, M, w->N / 2, w->K, ldx, x_fragment_major, partial_bytes, w->dtype, stream);
    return hx_linear_decode_partial_xreg(partial, x, w->packed, M, w->N, w->K, ldx, x_fragment_major, partial_bytes,
                                         w->dtype, stream);
  }
  if (x_fragment_major) return HX_ERR_STRIDE;                  // the LDS-slice kernel stages row-major x
  return hx_linear_decode_partial_packed(partial, x, w->packed, M, w->N, w->K, ldx, partial_bytes, w->dtype, stream);
}
