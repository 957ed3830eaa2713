// decode_chain.hip — the dense part of one decoder layer of a decode step as ONE launch:
//     o GEMM -> [slab reduce + residual add + RMSNorm] -> gate|up GEMM -> [slab reduce + silu*mul]
//     -> down GEMM -> [slab reduce + residual add + RMSNorm] -> next layer's qkv GEMM
// i.e. hydrainfer/model/model_forward.py:84-105 (o_proj ... down_proj, the two norms) plus the
// qkv projection of model_forward.py:72-77 for the following layer, for a decode batch (M <= 32).
//
// Why one launch: as seven launches every seam costs a kernel boundary (~2 us with the HBM idle),
// a cold ramp and a tail, and the three small kernels run ~5 us each moving < 1 MB
// (profiles/r2_base_timeline.md: 97 us of a 163 us layer for 405 MB of weights).  Here the seven
// phases are WORK ITEMS of one grid:
//   * a workgroup takes its item by an atomic TICKET, never by blockIdx: items are numbered phase
//     by phase, so every item of phase P is held by a running workgroup before any item of phase
//     P+1 is handed out — a workgroup that waits for phase P can only be waiting for workgroups
//     that are already running.  Deadlock-free with no assumption about dispatch order, residency
//     or workgroup -> XCD placement (no cooperative launch, no grid barrier).
//   * a GEMM item issues its first 32 KiB of weight loads per wave BEFORE it looks at its
//     dependency: weights do not depend on activations, so the HBM stream of phase P+1 starts
//     while phase P drains and while the small phases run.
//   * hand-over inside the launch (MI355X_MICROARCH.md, inter-workgroup visibility): producers
//     store write-through (sc1) whole 128-byte lines per wave instruction, every storing wave
//     drains (s_waitcnt vmcnt(0)), workgroup barrier, ONE agent-scope atomic add on the phase
//     counter; consumers poll that counter with relaxed agent loads (bounded: a timeout sets the
//     error word instead of hanging the GPU) and read the handed-over bytes only with sc1 loads.
//     No address is written twice within a launch (h_in / h_mid / h_out, x_post / x_next are
//     distinct buffers), so no cache can hold an older version of a handed-over line.
// Rounding points are those of the separate kernels (gemm_skinny.hip + the slab consumers of
// norm_rope_act.hip): fp32 split-K partials added in split order and rounded once to T, the
// RMSNorm reduction tree of add_rms_norm_slab_kernel<.., 512> reproduced exactly — the chain is
// bit-identical to the eight-launch path (tests/test_gpu_chain.py).
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include "attn_common.h"

namespace {

using namespace hx;

constexpr int kNW = 4;                  // waves per workgroup
constexpr int kThreads = kNW * 64;
constexpr int kChunk = 16;              // k-steps (of 32) per register buffer: 16 KiB of W per wave
constexpr int kMaxKs = 32;              // k-steps per split (1024 k)
constexpr int kRS = kMaxKs * 64 + 32;   // LDS row stride of the x slice, bytes
constexpr int kXBytes = 32 * kRS;       // x slice, two 16-row blocks
constexpr int kLdsBytes = kXBytes + kNW * 2048 + 64;   // + transpose images + ticket / reduction words

enum { PH_O = 0, PH_NORM1, PH_GU, PH_SILU, PH_DOWN, PH_NORM2, PH_QKV, PH_COUNT };
// word indices in the sync block: every word that is polled or added to sits on a 128-byte line
// of its own (32 words) — pollers of one phase never slow the ticket or another phase's counter
enum { SY_LINE = 32, SY_TICKET = 0, SY_COUNT = 1 * SY_LINE, SY_FLAG = 8 * SY_LINE, SY_ERR = HX_CHAIN_SYNC_ERR };
constexpr uint64_t kTimeoutTicks = 200000000ull;   // 2 s of the 100 MHz s_memrealtime clock

struct CGemm {
  const u16* w;
  const u16* x;
  float* partial;
  int64_t ldw, ldx;
  int32_t N, K, n_splits, gx, R;
  int32_t handoff;   // x was written inside this launch
  int32_t tiled;     // slabs are consumed inside this launch: [split][n/16][Mpad][16] + sc1 stores
};

struct CNorm {
  const float* slabs;     // tiled slabs of the preceding GEMM
  const u16* res_in;
  u16* res_out;
  u16* x_out;
  const u16* weight;
  int32_t n_splits, n_rg, res_handoff, pad;
};

struct CParams {
  CGemm g[4];             // o, gate|up, down, qkv(next)
  CNorm nrm[2];
  const float* gu_slabs;
  u16* act;
  uint32_t* sync;
  int32_t gu_splits, gu_n_rg, silu_chunks;
  int32_t M, Mpad, hidden, inter;
  float eps;
  int32_t end[PH_COUNT];  // cumulative item counts
  unsigned long long* trace;   // debug (chain_trace option): [item][4] = ticket time, dependency-ready time,
                               // end time (100 MHz clock), phase | XCC id << 8
};

typedef __amdgpu_buffer_rsrc_t rsrc_t;
__device__ __forceinline__ rsrc_t make_rsrc(const void* p) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, 0x7fffffff, 0x00020000);
}
typedef unsigned int bu32x4 __attribute__((__vector_size__(16)));
// AUX 16 = sc1 (agent scope: bypasses the per-CU L1 on loads, writes through on stores)
template <int AUX>
__device__ __forceinline__ u16x8 bload_u16x8(rsrc_t r, uint32_t byte_off) {
  return __builtin_bit_cast(u16x8, __builtin_amdgcn_raw_buffer_load_b128(r, byte_off, 0, AUX));
}
template <int AUX>
__device__ __forceinline__ f32x4 bload_f32x4(rsrc_t r, uint32_t byte_off) {
  return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, byte_off, 0, AUX));
}
template <int AUX, typename V>
__device__ __forceinline__ void bstore16(V v, rsrc_t r, uint32_t byte_off) {
  __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(bu32x4, v), r, byte_off, 0, AUX);
}

__device__ __forceinline__ uint32_t load_flag(const uint32_t* sync, int ph) {
  return __hip_atomic_load(sync + SY_FLAG + ph * SY_LINE, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// Workgroup-level wait until phase `ph` is complete.  Wave 0 polls the phase's FLAG word (written
// once, by the workgroup whose counter add was the last of the phase; the counter itself has no
// pollers); `first` is a value of the flag loaded earlier by wave 0, so the common case costs no
// extra round trip.  The other waves wait at the barrier.
__device__ __forceinline__ void wait_phase(uint32_t* sync, int ph, uint32_t first,
                                           unsigned long long* trace_slot) {
  if (threadIdx.x < 64) {
    uint32_t v = __builtin_amdgcn_readfirstlane(first);
    if (!v) {
      const uint64_t t0 = __builtin_amdgcn_s_memrealtime();
      for (;;) {
        __builtin_amdgcn_s_sleep(24);
        v = __builtin_amdgcn_readfirstlane(load_flag(sync, ph));
        if (v) break;
        if (__builtin_amdgcn_s_memrealtime() - t0 > kTimeoutTicks) {
          if (threadIdx.x == 0)
            __hip_atomic_fetch_or(sync + SY_ERR, 1u << ph, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          break;
        }
      }
    }
    if (trace_slot && threadIdx.x == 0) trace_slot[1] = __builtin_amdgcn_s_memrealtime();
  }
  __syncthreads();
  // no instruction: keeps the compiler from moving the hand-over loads above the poll
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  asm volatile("" ::: "memory");
}

// every storing wave drains its write-through stores, then ONE lane signals for the workgroup;
// the workgroup whose add completes the phase raises the phase's flag
__device__ __forceinline__ void publish(uint32_t* sync, int ph, uint32_t n_items) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0) {
    const uint32_t old = __hip_atomic_fetch_add(sync + SY_COUNT + ph * SY_LINE, 1u, __ATOMIC_RELAXED,
                                                __HIP_MEMORY_SCOPE_AGENT);
    if (old + 1 == n_items)
      __hip_atomic_store(sync + SY_FLAG + ph * SY_LINE, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}

__device__ __forceinline__ float silu_f32(float x) { return x / (1.0f + __expf(-x)); }

__device__ __forceinline__ uint32_t phase_items(const CParams& p, int ph) {
  return (uint32_t)(p.end[ph] - (ph ? p.end[ph - 1] : 0));
}

// ---------------------------------------------------------------------------------------------
// GEMM item: the body of gemm_skinny_kernel<T, 2, R, 4> for workgroup (bx, split) of phase `ph`.
// partial[s][m][n] = sum_{k in split s} x[m][k] * W[n][k]
// ---------------------------------------------------------------------------------------------
template <typename T, int R>
__device__ __forceinline__ void gemm_item(const CParams& p, const CGemm& gp, int item, char* smem, int ph,
                                          unsigned long long* tr) {
  constexpr int MB = 2;
  const int lane = threadIdx.x & 63;
  const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int r = lane & 15, g = lane >> 4, c = lane & 15;
  const int bx = item % gp.gx, split = item / gp.gx;
  const int total_ks = gp.K >> 5;
  const int ks0 = split * kMaxKs;
  const int nks = min(kMaxKs, total_ks - ks0);   // multiple of 8
  const int KR = nks << 5;
  const int n_rg_all = gp.N >> 4;
  const int rg0 = bx * (kNW * R) + w;

  // 0. the dependency counter is the OLDEST load of the wave: its value is back long before the
  // weights, and the x loads that need it follow without a second exposed round trip
  uint32_t first = 0;
  if (gp.handoff && w == 0) first = load_flag(p.sync, ph - 1);

  // 1. W prefetch (independent of every activation).  Load layout: instruction j of a chunk covers
  // rows 8*(j&1) + (lane>>3) and the 128-byte column block j>>1; lane&7 selects the 16-byte piece.
  const int lrow = lane >> 3, lpiece = lane & 7;
  const u16* wb = gp.w + (int64_t)ks0 * 32 + 8 * lpiece;
  auto load = [&](u16x8 (&buf)[kChunk], int it) {
    const int rgi = it >> 1, ch = it & 1;
    const int n0 = min(rg0 + rgi * kNW, n_rg_all - 1) << 4;
    const int last_cb = ((nks - ch * kChunk) >> 1) - 1;
    const u16* wp = wb + (int64_t)(n0 + lrow) * gp.ldw + ch * (kChunk * 32);
#pragma unroll
    for (int j = 0; j < kChunk; ++j) {
      const int cb = max(min(j >> 1, last_cb), -ch * (kChunk / 2));
      buf[j] = __builtin_nontemporal_load(
          reinterpret_cast<const u16x8*>(wp + (int64_t)(8 * (j & 1)) * gp.ldw + 64 * cb));
    }
  };
  u16x8 buf[2][kChunk];
  load(buf[0], 0);
  load(buf[1], 1);
  __builtin_amdgcn_sched_barrier(0);

  // 2. dependency
  if (gp.handoff) wait_phase(p.sync, ph - 1, first, tr);

  // 3. x slice -> registers -> LDS ([32 rows][kMaxKs k-steps], zero beyond this split / beyond M)
  constexpr int kCpr = kMaxKs * 4;                         // 16-byte chunks per LDS row
  constexpr int XPT = MB * 16 * kCpr / kThreads;           // chunks per thread
  u16x8 xr[XPT];
  {
    const rsrc_t xrs = make_rsrc(gp.x);
    const uint32_t col0 = (uint32_t)ks0 * 64;              // bytes
    uint32_t off[XPT];
#pragma unroll
    for (int j = 0; j < XPT; ++j) {
      const int i = threadIdx.x + j * kThreads;
      const int row = i / kCpr, ch = i % kCpr;
      const bool ok = row < p.M && ch * 8 < KR;
      off[j] = (uint32_t)(ok ? row : 0) * (uint32_t)(gp.ldx * 2) + col0 + (ok ? ch * 16 : 0);
    }
    if (gp.handoff) {
#pragma unroll
      for (int j = 0; j < XPT; ++j) xr[j] = bload_u16x8<16>(xrs, off[j]);
    } else {
#pragma unroll
      for (int j = 0; j < XPT; ++j) xr[j] = bload_u16x8<0>(xrs, off[j]);
    }
  }
#pragma unroll
  for (int j = 0; j < XPT; ++j) {
    const int i = threadIdx.x + j * kThreads;
    const int row = i / kCpr, ch = i % kCpr;
    const bool ok = row < p.M && ch * 8 < KR;
    *reinterpret_cast<u16x8*>(smem + row * kRS + ch * 16) = ok ? xr[j] : u16x8{0, 0, 0, 0, 0, 0, 0, 0};
  }
  __syncthreads();

  // 4. stream W through the wave-private transpose image into MFMA A fragments
  char* tl = smem + kXBytes + w * 2048;
  const int wr_off0 = lrow * 128 + 16 * (lpiece ^ ((lrow >> 1) & 7));
  const int wr_off1 = (lrow + 8) * 128 + 16 * (lpiece ^ (((lrow + 8) >> 1) & 7));
  const char* xl = smem + c * kRS + g * 16;
  f32x4 acc[MB];
#pragma unroll
  for (int mb = 0; mb < MB; ++mb) acc[mb] = f32x4{0.f, 0.f, 0.f, 0.f};
  const rsrc_t prs = make_rsrc(gp.partial);

#pragma unroll
  for (int it = 0; it < 2 * R; ++it) {
    const int rgi = it >> 1, ch = it & 1;
    const char* xp = xl + ch * (kChunk * 64);
#pragma unroll
    for (int cb = 0; cb < kChunk / 2; ++cb) {
      *reinterpret_cast<u16x8*>(tl + wr_off0) = buf[it & 1][2 * cb];
      *reinterpret_cast<u16x8*>(tl + wr_off1) = buf[it & 1][2 * cb + 1];
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int st = 0; st < 2; ++st) {
        const u16x8 af = *reinterpret_cast<const u16x8*>(tl + r * 128 + 16 * ((4 * st + g) ^ ((r >> 1) & 7)));
#pragma unroll
        for (int mb = 0; mb < MB; ++mb) {
          const u16x8 xf = *reinterpret_cast<const u16x8*>(xp + mb * 16 * kRS + (2 * cb + st) * 64);
          acc[mb] = Mfma<T>::mma(af, xf, acc[mb]);
        }
      }
      __builtin_amdgcn_wave_barrier();
    }
    if (it + 2 < 2 * R) load(buf[it & 1], it + 2);
    if (ch == 1) {
      const int rg = rg0 + rgi * kNW;
      if (rg < n_rg_all) {
#pragma unroll
        for (int mb = 0; mb < MB; ++mb) {
          if (gp.tiled) {
            // lane (g, c): row mb*16 + c, columns 4g..4g+3 of tile rg -> one wave instruction
            // writes 1 KiB contiguous = 8 whole lines (rows >= M are zero: x rows are zero)
            const uint32_t off = ((((uint32_t)split * n_rg_all + rg) * p.Mpad + mb * 16 + c) * 16 + 4 * g) * 4;
            bstore16<16>(acc[mb], prs, off);
          } else {
            const int m = mb * 16 + c;
            if (m < p.M)
              *reinterpret_cast<f32x4*>(gp.partial + ((int64_t)split * p.M + m) * gp.N + (rg << 4) + 4 * g) = acc[mb];
          }
        }
      }
#pragma unroll
      for (int mb = 0; mb < MB; ++mb) acc[mb] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
  }
  if (gp.tiled) publish(p.sync, ph, phase_items(p, ph));
}

// sum of the n_splits tiled slab pieces of 8 consecutive columns (vector i of row `row`),
// added in split order and rounded once to T — slab_sum8 of norm_rope_act.hip on the tile layout
template <typename T>
__device__ __forceinline__ void tile_sum8(rsrc_t rs, int n_splits, int n_rg, int Mpad, int row, int i,
                                          float (&acc)[8]) {
  const uint32_t base = (((uint32_t)(i >> 1) * Mpad + row) * 16 + 8 * (i & 1)) * 4;
  const uint32_t sstride = (uint32_t)n_rg * Mpad * 64;
  // the slab pieces come from L2 / memory (sc1): all loads of a batch of 6 splits are issued
  // before the first add, the adds stay in split order
  constexpr int kB = 6;
  f32x4 a = {0.f, 0.f, 0.f, 0.f}, b = {0.f, 0.f, 0.f, 0.f};
  for (int s0 = 0; s0 < n_splits; s0 += kB) {
    f32x4 pa[kB], pb[kB];
#pragma unroll
    for (int k = 0; k < kB; ++k) {
      const int s = min(s0 + k, n_splits - 1);
      pa[k] = bload_f32x4<16>(rs, base + s * sstride);
      pb[k] = bload_f32x4<16>(rs, base + s * sstride + 16);
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

// ---------------------------------------------------------------------------------------------
// norm item: one row.  add_rms_norm_slab_kernel<T, MAXV, 512> evaluated by 256 threads: thread u
// plays the virtual threads u and u + 256, so per-thread sums, wave sums and the order in which
// the eight wave sums are added are the same — bit-identical results.
// ---------------------------------------------------------------------------------------------
template <typename T, int MAXV>
__device__ __forceinline__ void norm_item(const CParams& p, const CNorm& np, int row, char* smem, int ph,
                                          unsigned long long* tr) {
  float* red = reinterpret_cast<float*>(smem + kXBytes + kNW * 2048 + 16);   // 8 floats
  uint32_t first = 0;
  if (threadIdx.x < 64) first = load_flag(p.sync, ph - 1);
  wait_phase(p.sync, ph - 1, first, tr);

  const int nvec = p.hidden / 8;
  const rsrc_t srs = make_rsrc(np.slabs);
  const rsrc_t rrs = make_rsrc(np.res_in);
  const rsrc_t hrs = make_rsrc(np.res_out);
  const rsrc_t ors = make_rsrc(np.x_out);
  const uint32_t row_off = (uint32_t)row * (uint32_t)p.hidden * 2;
  float x[2 * MAXV][8];
  float ss[2] = {0.f, 0.f};
#pragma unroll
  for (int v = 0; v < 2; ++v) {
#pragma unroll
    for (int j = 0; j < MAXV; ++j) {
      const int i = threadIdx.x + 256 * v + 512 * j;
      if (i < nvec) {
        float a[8];
        tile_sum8<T>(srs, np.n_splits, np.n_rg, p.Mpad, row, i, a);
        const u16x8 rr = np.res_handoff ? bload_u16x8<16>(rrs, row_off + i * 16)
                                        : bload_u16x8<0>(rrs, row_off + i * 16);
        u16x8 h;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const float sum = round_to<T>(a[e] + T::to_float(rr[e]));
          x[v * MAXV + j][e] = sum;
          h[e] = T::from_float(sum);
          ss[v] += sum * sum;
        }
        bstore16<16>(h, hrs, row_off + i * 16);
      }
    }
  }
  const float t0 = wave_sum(ss[0]), t1 = wave_sum(ss[1]);
  if ((threadIdx.x & 63) == 0) {
    red[threadIdx.x >> 6] = t0;
    red[4 + (threadIdx.x >> 6)] = t1;
  }
  __syncthreads();
  float total = 0.f;
#pragma unroll
  for (int k = 0; k < 8; ++k) total += red[k];
  const float inv = rsqrtf(total / (float)p.hidden + p.eps);
#pragma unroll
  for (int v = 0; v < 2; ++v) {
#pragma unroll
    for (int j = 0; j < MAXV; ++j) {
      const int i = threadIdx.x + 256 * v + 512 * j;
      if (i < nvec) {
        const u16x8 wv = *reinterpret_cast<const u16x8*>(np.weight + i * 8);
        u16x8 o;
#pragma unroll
        for (int e = 0; e < 8; ++e)
          o[e] = T::from_float(round_to<T>(x[v * MAXV + j][e] * inv) * T::to_float(wv[e]));
        bstore16<16>(o, ors, row_off + i * 16);
      }
    }
  }
  publish(p.sync, ph, phase_items(p, ph));
}

// silu item: 256 vectors of one row: act = (T)silu(gate) * up from the gate|up slabs
template <typename T>
__device__ __forceinline__ void silu_item(const CParams& p, int item, int ph, unsigned long long* tr) {
  uint32_t first = 0;
  if (threadIdx.x < 64) first = load_flag(p.sync, ph - 1);
  wait_phase(p.sync, ph - 1, first, tr);
  const int row = item / p.silu_chunks, chunk = item % p.silu_chunks;
  const int nvec = p.inter / 8;
  const int i = chunk * kThreads + threadIdx.x;
  if (i < nvec) {
    const rsrc_t srs = make_rsrc(p.gu_slabs);
    float gte[8], up[8];
    tile_sum8<T>(srs, p.gu_splits, p.gu_n_rg, p.Mpad, row, i, gte);
    tile_sum8<T>(srs, p.gu_splits, p.gu_n_rg, p.Mpad, row, nvec + i, up);
    u16x8 rv;
#pragma unroll
    for (int e = 0; e < 8; ++e) rv[e] = T::from_float(round_to<T>(silu_f32(gte[e])) * up[e]);
    bstore16<16>(rv, make_rsrc(p.act), ((uint32_t)row * (uint32_t)p.inter + i * 8) * 2);
  }
  publish(p.sync, ph, phase_items(p, ph));
}

template <typename T>
__global__ __launch_bounds__(kThreads, 2) void decode_chain_kernel(const CParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  int* s_item = reinterpret_cast<int*>(smem + kXBytes + kNW * 2048);
  if (threadIdx.x == 0)
    *s_item = (int)__hip_atomic_fetch_add(p.sync + SY_TICKET, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  __syncthreads();
  const int t = __builtin_amdgcn_readfirstlane(*s_item);
  int ph = 0;
  while (ph < PH_COUNT && t >= p.end[ph]) ++ph;
  if (ph >= PH_COUNT) return;
  const int item = t - (ph ? p.end[ph - 1] : 0);
  unsigned long long* tr = p.trace ? p.trace + 4 * (int64_t)t : nullptr;
  if (tr && threadIdx.x == 0) {
    tr[0] = __builtin_amdgcn_s_memrealtime();
    tr[1] = 0;
    tr[3] = (unsigned long long)ph | ((unsigned long long)(__builtin_amdgcn_s_getreg(20 | (0 << 6) | (3 << 11)) & 15) << 8);
  }
  if (ph == PH_NORM1 || ph == PH_NORM2) {
    const CNorm& np = p.nrm[ph == PH_NORM2];
    if (p.hidden / 8 <= 512) norm_item<T, 1>(p, np, item, smem, ph, tr);
    else norm_item<T, 2>(p, np, item, smem, ph, tr);
  } else if (ph == PH_SILU) {
    silu_item<T>(p, item, ph, tr);
  } else {
    const CGemm& gp = p.g[ph >> 1];   // PH_O, PH_GU, PH_DOWN, PH_QKV = 0, 2, 4, 6
    switch (gp.R) {
      case 1: gemm_item<T, 1>(p, gp, item, smem, ph, tr); break;
      case 2: gemm_item<T, 2>(p, gp, item, smem, ph, tr); break;
      default: gemm_item<T, 3>(p, gp, item, smem, ph, tr); break;
    }
  }
  if (tr) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (threadIdx.x == 0) tr[2] = __builtin_amdgcn_s_memrealtime();
  }
}

int g_chain_trace = 0;   // debug: per-item timestamps behind the slabs in the workspace
int g_chain_r[4] = {1, 2, 2, 1};   // row groups per wave: o, gate|up, down, qkv (tuning: HX_CHAIN_R / chain_r_*)

}  // namespace

namespace hx {

int chain_set_option(const char* name, int value) {
  if (!strcmp(name, "chain_trace")) { g_chain_trace = value ? 1 : 0; return HX_OK; }
  static const char* names[4] = {"chain_r_o", "chain_r_gu", "chain_r_down", "chain_r_qkv"};
  for (int i = 0; i < 4; ++i)
    if (!strcmp(name, names[i])) {
      if (value < 1 || value > 3) return HX_ERR_SHAPE;
      g_chain_r[i] = value;
      return HX_OK;
    }
  return HX_ERR_UNSUPPORTED;
}

}  // namespace hx

using namespace hx;

static int chain_splits(int64_t K) { return (int)(((K >> 5) + kMaxKs - 1) / kMaxKs); }

static bool chain_shape_ok(int64_t M, int64_t hidden, int64_t inter, int64_t q_size, int64_t qkv_n) {
  return M >= 1 && M <= 32 && hidden % 256 == 0 && inter % 256 == 0 && q_size % 256 == 0 &&
         hidden / 8 <= 1024 && (qkv_n == 0 || qkv_n % 16 == 0);
}

extern "C" int64_t hx_decode_chain_workspace_bytes(int64_t M, int64_t hidden, int64_t inter, int64_t q_size) {
  if (!chain_shape_ok(M, hidden, inter, q_size, 0)) return 0;
  const int64_t Mpad = 32;
  const int64_t fl = (int64_t)chain_splits(q_size) * Mpad * hidden + (int64_t)chain_splits(hidden) * Mpad * 2 * inter +
                     (int64_t)chain_splits(inter) * Mpad * hidden;
  return fl * (int64_t)sizeof(float);
}

extern "C" int hx_decode_chain(const hx_chain_args* a, hx_stream stream) {
  if (!a) return HX_ERR_NULL;
  if (!chain_shape_ok(a->M, a->hidden, a->inter, a->q_size, a->qkv_n)) return HX_ERR_SHAPE;
  if (a->dtype != HX_F16 && a->dtype != HX_BF16) return HX_ERR_DTYPE;
  if (!a->attn_out || !a->h_in || !a->w_o || !a->w_gate_up || !a->w_down || !a->norm_post_weight ||
      !a->norm_next_weight || !a->h_mid || !a->h_out || !a->x_post || !a->act || !a->x_next ||
      !a->workspace || !a->sync)
    return HX_ERR_NULL;
  if (a->qkv_n && (!a->w_qkv_next || !a->qkv_partial)) return HX_ERR_NULL;
  if (a->workspace_bytes < hx_decode_chain_workspace_bytes(a->M, a->hidden, a->inter, a->q_size))
    return HX_ERR_WORKSPACE;
  if (a->qkv_n && a->qkv_partial_bytes < (int64_t)chain_splits(a->hidden) * a->M * a->qkv_n * (int64_t)sizeof(float))
    return HX_ERR_WORKSPACE;
  if (a->attn_out_stride % 8 || a->ld_o % 8 || a->ld_gate_up % 8 || a->ld_down % 8 || (a->qkv_n && a->ld_qkv % 8))
    return HX_ERR_STRIDE;
  const void* ptrs[] = {a->attn_out, a->h_in, a->w_o, a->w_gate_up, a->w_down, a->w_qkv_next, a->norm_post_weight,
                        a->norm_next_weight, a->h_mid, a->h_out, a->x_post, a->act, a->x_next, a->qkv_partial,
                        a->workspace};
  for (const void* q : ptrs)
    if (q && !aligned16(q)) return HX_ERR_STRIDE;
  // hand-over buffers must be pairwise distinct: nothing is written twice within the launch
  const void* bufs[] = {a->h_in, a->h_mid, a->h_out, a->x_post, a->x_next, a->act, a->attn_out};
  for (int i = 0; i < 7; ++i)
    for (int j = i + 1; j < 7; ++j)
      if (bufs[i] == bufs[j]) return HX_ERR_SHAPE;

  static const char* renv = getenv("HX_CHAIN_R");
  if (renv) {
    int r[4];
    if (sscanf(renv, "%d,%d,%d,%d", &r[0], &r[1], &r[2], &r[3]) == 4)
      for (int i = 0; i < 4; ++i)
        if (r[i] >= 1 && r[i] <= 3) g_chain_r[i] = r[i];
  }

  CParams p;
  memset(&p, 0, sizeof(p));
  p.M = a->M; p.Mpad = 32; p.hidden = a->hidden; p.inter = a->inter; p.eps = a->eps;
  p.sync = a->sync;
  float* ws = (float*)a->workspace;
  float* slabs_o = ws;
  float* slabs_gu = slabs_o + (int64_t)chain_splits(a->q_size) * p.Mpad * a->hidden;
  float* slabs_dn = slabs_gu + (int64_t)chain_splits(a->hidden) * p.Mpad * 2 * a->inter;
  auto gemm = [&](int idx, const void* w, int64_t ldw, const void* x, int64_t ldx, float* partial, int N, int K,
                  int handoff, int tiled) {
    CGemm& g = p.g[idx];
    g.w = (const u16*)w; g.x = (const u16*)x; g.partial = partial; g.ldw = ldw; g.ldx = ldx;
    g.N = N; g.K = K; g.n_splits = chain_splits(K); g.R = g_chain_r[idx];
    g.gx = ((N >> 4) + kNW * g.R - 1) / (kNW * g.R);
    g.handoff = handoff; g.tiled = tiled;
    return N ? g.gx * g.n_splits : 0;
  };
  int items[PH_COUNT];
  items[PH_O] = gemm(0, a->w_o, a->ld_o, a->attn_out, a->attn_out_stride, slabs_o, a->hidden, a->q_size, 0, 1);
  items[PH_NORM1] = a->M;
  items[PH_GU] = gemm(1, a->w_gate_up, a->ld_gate_up, a->x_post, a->hidden, slabs_gu, 2 * a->inter, a->hidden, 1, 1);
  p.silu_chunks = (a->inter / 8 + kThreads - 1) / kThreads;
  items[PH_SILU] = a->M * p.silu_chunks;
  items[PH_DOWN] = gemm(2, a->w_down, a->ld_down, a->act, a->inter, slabs_dn, a->hidden, a->inter, 1, 1);
  items[PH_NORM2] = a->M;
  items[PH_QKV] = gemm(3, a->w_qkv_next, a->ld_qkv, a->x_next, a->hidden, a->qkv_partial, a->qkv_n, a->hidden, 1, 0);
  int total = 0;
  for (int i = 0; i < PH_COUNT; ++i) { total += items[i]; p.end[i] = total; }

  CNorm& n1 = p.nrm[0];
  n1.slabs = slabs_o; n1.res_in = (const u16*)a->h_in; n1.res_out = (u16*)a->h_mid; n1.x_out = (u16*)a->x_post;
  n1.weight = (const u16*)a->norm_post_weight; n1.n_splits = p.g[0].n_splits; n1.n_rg = a->hidden >> 4; n1.res_handoff = 0;
  CNorm& n2 = p.nrm[1];
  n2.slabs = slabs_dn; n2.res_in = (const u16*)a->h_mid; n2.res_out = (u16*)a->h_out; n2.x_out = (u16*)a->x_next;
  n2.weight = (const u16*)a->norm_next_weight; n2.n_splits = p.g[2].n_splits; n2.n_rg = a->hidden >> 4; n2.res_handoff = 1;
  p.gu_slabs = slabs_gu; p.gu_splits = p.g[1].n_splits; p.gu_n_rg = (2 * a->inter) >> 4; p.act = (u16*)a->act;

  if (g_chain_trace) {
    const int64_t need = hx_decode_chain_workspace_bytes(a->M, a->hidden, a->inter, a->q_size);
    if (a->workspace_bytes >= need + (int64_t)total * 32)
      p.trace = reinterpret_cast<unsigned long long*>((char*)a->workspace + need);
  }
  hipStream_t s = (hipStream_t)stream;
  const void* fn = a->dtype == HX_F16 ? (const void*)decode_chain_kernel<F16> : (const void*)decode_chain_kernel<BF16>;
  static bool attr_set[2] = {false, false};
  if (!attr_set[a->dtype == HX_BF16]) {
    hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, kLdsBytes);
    if (e != hipSuccess) return hip_rc(e);
    attr_set[a->dtype == HX_BF16] = true;
  }
  if (a->dtype == HX_F16) decode_chain_kernel<F16><<<total, kThreads, kLdsBytes, s>>>(p);
  else decode_chain_kernel<BF16><<<total, kThreads, kLdsBytes, s>>>(p);
  int rc = check_launch();
  if (rc) return rc;
  return a->qkv_n ? p.g[3].n_splits : 0;
}
