// decode_chain.hip — the dense part of one decoder layer of a decode step as ONE launch:
//     o GEMM -> [slab reduce + residual add + RMSNorm] -> gate|up GEMM -> [slab reduce + silu*mul]
//     -> down GEMM -> [slab reduce + residual add + RMSNorm] -> next layer's qkv GEMM
// i.e. hydrainfer/model/model_forward.py:84-105 (o_proj ... down_proj, the two norms) plus the
// qkv projection of model_forward.py:72-77 for the following layer, for a decode batch (M <= 32).
//
// Why one launch: as seven launches every seam costs a kernel boundary (~2 us with the HBM idle),
// a cold ramp and a tail, and the three small kernels run ~5 us each moving < 1 MB
// (profiles/r2_base_timeline.md: 97 us of a 163 us layer for 405 MB of weights).  Here the phases
// are WORK ITEMS of one grid of persistent workgroups:
//   * a workgroup takes its items by an atomic TICKET, never by blockIdx: items are numbered in
//     dependency order, so everything an item waits for is held by a workgroup that is already
//     running and that processes its own tickets in increasing order — the item with the smallest
//     unfinished ticket can always run.  Deadlock-free with no assumption about dispatch order,
//     residency or workgroup -> XCD placement (no cooperative launch, no grid barrier).  The next
//     ticket is requested before the current item's stores drain, so its latency is hidden.
//   * a GEMM item issues its first 32 KiB of weight loads per wave BEFORE it looks at its
//     dependency: weights do not depend on activations, so the HBM stream of the next phase runs
//     while the previous one drains and while the row-wise phases (norms) run.
//   * gate|up -> silu*mul -> down is sliced by the K split of the down projection (1024 columns of
//     `act`): the silu items and the down items of slice j wait only for the gate|up items of
//     slice j, so that seam is not a grid-wide wait.  Only the two RMSNorms (row statistics over
//     the whole hidden size) are grid-wide dependencies.
//   * hand-over inside the launch (MI355X_MICROARCH.md, inter-workgroup visibility): producers
//     store write-through (sc1) whole 128-byte lines per wave instruction, every storing wave
//     drains (s_waitcnt vmcnt(0)), workgroup barrier, ONE agent-scope atomic add on the segment's
//     counter; the workgroup whose add completes the count raises the segment's flag; consumers
//     poll that flag with relaxed agent loads (bounded: a timeout sets the error word instead of
//     hanging the GPU) and read the handed-over bytes only with sc1 loads.  Every polled or added
//     word has a 128-byte line of its own.  No address is written twice within a launch (h_in /
//     h_mid / h_out, x_post / x_next are distinct buffers), so no cache can hold an older version
//     of a handed-over line.
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
constexpr int kLdsBytes = kXBytes + 64;   // + ticket / reduction words

// item kinds; a SEGMENT is a run of consecutive tickets of one kind (and one slice)
enum { K_GEMM_O = 0, K_NORM, K_GEMM_GU, K_SILU, K_GEMM_DOWN, K_GEMM_QKV };
// sync block: every word that is polled or added to sits on a 128-byte line of its own (32
// words) — pollers of one flag never slow the ticket counter or another segment's counter
enum {
  SY_LINE = 32,
  // region A (words 0 .. SY_COPY_WORDS): ticket, counters, error word
  L_TICKET = 0,
  L_COUNT = 1,    // + phase slot (o, norm1, down, norm2): 4 lines
  L_ERR = 15,
  L_GCOUNT = 16,  // + slice (gate|up items of a slice)
  L_SCOUNT = 32,  // + slice (silu items of a slice)
  // flag lines, relative to a flag copy (copy c starts at word SY_COPY_WORDS * (1 + c))
  F_PHASE = 0,    // + phase slot
  F_G = 4,        // + slice
  F_S = 20,       // + slice: 36 lines <= 64 per copy
  // every FLAG exists in SY_COPIES copies 8 KiB apart (different memory channels): a waiting
  // workgroup polls the copy of its XCC id, so hundreds of pollers never queue on one line —
  // polling one line from every CU stalls that line's channel and with it every wave that has a
  // weight load on it (measured: the whole launch ran 2x slower)
  SY_COPIES = 8,
  SY_COPY_WORDS = 2048,
};
constexpr int kMaxSlices = 16;
constexpr int kMaxSeg = 40;   // o, norm1, gu + silu per slice, down, norm2, qkv
constexpr uint64_t kTimeoutTicks = 200000000ull;   // 2 s of the 100 MHz s_memrealtime clock

// what an item kind works on (one descriptor per kind: o, norm1, gate|up, silu, down, norm2, qkv);
// fetched with a dynamic index per item, so nothing of it stays in registers across items
struct Desc {
  const u16* w;         // GEMM: weights [N][K]; norm: weight vector
  const u16* x;         // GEMM: activations [M][K]; norm: residual in
  float* partial;       // GEMM: slabs out; norm / silu: slabs in
  u16* out0;            // norm: residual out; silu: act
  u16* out1;            // norm: normalised x out
  int64_t ldw, ldx;
  int32_t N, K, n_splits, gx, R;
  int32_t tiled;        // GEMM: slabs are consumed inside this launch: [split][n/16][Mpad][16] + sc1 stores
  int32_t n_rg;         // norm / silu: 16-column tiles per slab row
  int32_t handoff;      // norm: residual in was written inside this launch
};
enum { D_O = 0, D_NORM1, D_GU, D_SILU, D_DOWN, D_NORM2, D_QKV, D_COUNT };

struct Seg {
  int32_t end;          // cumulative ticket count up to and including this segment
  int32_t kind;
  int32_t desc;         // index into CParams::desc
  int32_t idx;          // slice (gate|up, silu)
  int32_t wait_word;    // flag word (relative to a flag copy) this segment's items wait for, -1 = none;
                        // down items add split * SY_LINE (one silu flag per slice)
  int32_t count_word;   // sync word of the counter its items add to, -1 = none
  int32_t flag_word;    // flag word raised when the counter reaches target
  int32_t target;
  int32_t a, b;         // gate|up / silu: first tile pair of the slice, pairs in the slice
};

struct CParams {
  uint32_t* sync;
  unsigned long long* trace;   // debug (chain_trace option): [ticket][4] = start, dependency seen, end
                               // (100 MHz clock), kind | idx << 4 | XCC id << 8
  int32_t M, Mpad, hidden, inter;
  float eps;
  int32_t n_seg, total, pad;
  Desc desc[D_COUNT];
  Seg seg[kMaxSeg];
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

__device__ __forceinline__ uint32_t load_word(const uint32_t* sync, int word) {
  return __hip_atomic_load(sync + word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// the flag copy of this workgroup's XCC (HW_REG_XCC_ID, bits 3:0)
__device__ __forceinline__ const uint32_t* my_flags(const uint32_t* sync) {
  return sync + (1 + (__builtin_amdgcn_s_getreg(20 | (3 << 11)) & (SY_COPIES - 1))) * SY_COPY_WORDS;
}

// ONE lane adds to the segment's counter; the workgroup whose add completes the count raises every
// copy of the segment's flag.  Called by lane 0 of the LAST wave right after the workgroup's
// barrier behind the drained stores: that wave alone pays the atomic's round trip.
__device__ __forceinline__ void signal(uint32_t* sync, int count_word, int flag_word, int target) {
  const uint32_t old = __hip_atomic_fetch_add(sync + count_word, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  if (old + 1 == (uint32_t)target) {
#pragma unroll
    for (int cpy = 0; cpy < SY_COPIES; ++cpy)
      __hip_atomic_store(sync + (1 + cpy) * SY_COPY_WORDS + flag_word, 1u, __ATOMIC_RELAXED,
                         __HIP_MEMORY_SCOPE_AGENT);
  }
}

// Workgroup-level wait for the flag at sync[word].  Wave 0 polls; `first` is a value of the flag
// it loaded earlier, so the common case costs no extra round trip.  The other waves wait at the
// barrier.
__device__ __forceinline__ void wait_flag(uint32_t* sync, int word, uint32_t first, int tid,
                                          unsigned long long* trace_slot) {
  if (tid < 64) {
    uint32_t v = __builtin_amdgcn_readfirstlane(first);
    if (!v) {
      const uint64_t t0 = __builtin_amdgcn_s_memrealtime();
      const uint32_t* fl = my_flags(sync);
      for (int spin = 0;; ++spin) {
        // back off: 0.2, 0.4, then 0.8 us between polls
        if (spin == 0) __builtin_amdgcn_s_sleep(8);
        else if (spin == 1) __builtin_amdgcn_s_sleep(16);
        else __builtin_amdgcn_s_sleep(32);
        v = __builtin_amdgcn_readfirstlane(load_word(fl, word));
        if (v) break;
        if (__builtin_amdgcn_s_memrealtime() - t0 > kTimeoutTicks) {
          if (tid == 0)
            __hip_atomic_fetch_or(sync + L_ERR * SY_LINE, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          break;
        }
      }
    }
    if (trace_slot && tid == 0) trace_slot[1] = __builtin_amdgcn_s_memrealtime();
  }
  __syncthreads();
  // no instruction: keeps the compiler from moving the hand-over loads above the poll
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  asm volatile("" ::: "memory");
}

__device__ __forceinline__ float silu_f32(float x) { return x / (1.0f + __expf(-x)); }

// ---------------------------------------------------------------------------------------------
// GEMM item: the body of gemm_skinny_kernel<T, 2, R, 4> for one workgroup: K split `split`, wave w
// owns the 16-row weight tiles rg_first + w + 4*i (i < R) below rg_limit.
// partial[s][m][n] = sum_{k in split s} x[m][k] * W[n][k]
// ---------------------------------------------------------------------------------------------
template <typename T, int R>
__device__ __forceinline__ void gemm_item(const CParams& p, const Desc& gp, int split, int rg_first,
                                          int rg_limit, int wait_word, char* smem, int tid,
                                          unsigned long long* tr) {
  constexpr int MB = 2;
  const int lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int g = lane >> 4, c = lane & 15;
  const int total_ks = gp.K >> 5;
  const int ks0 = split * kMaxKs;
  const int nks = min(kMaxKs, total_ks - ks0);   // multiple of 8
  const int KR = nks << 5;
  const int n_rg_all = gp.N >> 4;
  const int rg0 = rg_first + w;
  const bool handoff = wait_word >= 0;

  // 0. the dependency flag is the OLDEST load of wave 0: its value is back long before the
  // weights, and the x loads that need it follow without a second exposed round trip
  uint32_t first = 0;
  if (handoff && w == 0) first = load_word(my_flags(p.sync), wait_word);

  // 1. W prefetch (independent of every activation).  Packed weights (hx_pack_decode_weight): the
  // fragments of (split, rg) are one run of nks KiB at KiB offset ks0 * n_rg + rg * nks; lane l's
  // 16 bytes of a fragment are the MFMA A operand itself
  // buffer loads: the 1 KiB fragment address is wave-uniform (scalar offset), the lane adds 16 l —
  // no per-lane 64-bit pointers to keep alive across the MFMA loop
  const rsrc_t wrs = make_rsrc(gp.w);
  const uint32_t lane_off = lane * 16;
  auto load = [&](u16x8 (&buf)[kChunk], int it) {
    const int rgi = it >> 1, ch = it & 1;
    const int rg = min(rg0 + rgi * kNW, rg_limit - 1);
    const uint32_t base = (uint32_t)(ks0 * n_rg_all + rg * nks) * 1024u;
#pragma unroll
    for (int j = 0; j < kChunk; ++j) {
      const int s = min(ch * kChunk + j, nks - 1);
      buf[j] = __builtin_bit_cast(u16x8, __builtin_amdgcn_raw_buffer_load_b128(wrs, lane_off, base + s * 1024u, 2));
    }
  };
  u16x8 buf[2][kChunk];
  load(buf[0], 0);
  load(buf[1], 1);
  __builtin_amdgcn_sched_barrier(0);

  // 2. dependency
  if (handoff) wait_flag(p.sync, wait_word, first, tid, tr);

  // 3. x slice -> registers -> LDS ([32 rows][kMaxKs k-steps], zero beyond this split / beyond M)
  constexpr int kCpr = kMaxKs * 4;                         // 16-byte chunks per LDS row
  constexpr int XPT = MB * 16 * kCpr / kThreads;           // chunks per thread
  u16x8 xr[XPT];
  {
    const rsrc_t xrs = make_rsrc(gp.x);
    const uint32_t col0 = (uint32_t)ks0 * 64;              // bytes
    auto xoff = [&](int j) {
      const int i = tid + j * kThreads;
      const int row = i / kCpr, ch = i % kCpr;
      const bool ok = row < p.M && ch * 8 < KR;
      return (uint32_t)(ok ? row : 0) * (uint32_t)(gp.ldx * 2) + col0 + (ok ? ch * 16 : 0);
    };
    if (handoff) {
#pragma unroll
      for (int j = 0; j < XPT; ++j) xr[j] = bload_u16x8<16>(xrs, xoff(j));
    } else {
#pragma unroll
      for (int j = 0; j < XPT; ++j) xr[j] = bload_u16x8<0>(xrs, xoff(j));
    }
  }
#pragma unroll
  for (int j = 0; j < XPT; ++j) {
    const int i = tid + j * kThreads;
    const int row = i / kCpr, ch = i % kCpr;
    const bool ok = row < p.M && ch * 8 < KR;
    *reinterpret_cast<u16x8*>(smem + row * kRS + ch * 16) = ok ? xr[j] : u16x8{0, 0, 0, 0, 0, 0, 0, 0};
  }
  __syncthreads();

  // 4. the fragments are the A operands; B = x^T fragments from LDS
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
    for (int j = 0; j < kChunk; ++j) {
#pragma unroll
      for (int mb = 0; mb < MB; ++mb) {
        const u16x8 xf = *reinterpret_cast<const u16x8*>(xp + mb * 16 * kRS + j * 64);
        acc[mb] = Mfma<T>::mma(buf[it & 1][j], xf, acc[mb]);
      }
    }
    __builtin_amdgcn_sched_barrier(0);   // keep the refill behind this chunk's MFMAs (register pressure)
    if (it + 2 < 2 * R) load(buf[it & 1], it + 2);
    __builtin_amdgcn_sched_barrier(0);
    if (ch == 1) {
      const int rg = rg0 + rgi * kNW;
      if (rg < rg_limit) {
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
__device__ __forceinline__ void norm_item(const CParams& p, const Desc& np, int row, int wait_word,
                                          char* smem, int tid, unsigned long long* tr) {
  float* red = reinterpret_cast<float*>(smem + kXBytes + 16);   // 8 floats
  uint32_t first = 0;
  if (tid < 64) first = load_word(my_flags(p.sync), wait_word);
  wait_flag(p.sync, wait_word, first, tid, tr);

  const int nvec = p.hidden / 8;
  const rsrc_t srs = make_rsrc(np.partial);
  const rsrc_t rrs = make_rsrc(np.x);
  const rsrc_t hrs = make_rsrc(np.out0);
  const rsrc_t ors = make_rsrc(np.out1);
  const uint32_t row_off = (uint32_t)row * (uint32_t)p.hidden * 2;
  float x[2 * MAXV][8];
  float ss[2] = {0.f, 0.f};
#pragma unroll
  for (int v = 0; v < 2; ++v) {
#pragma unroll
    for (int j = 0; j < MAXV; ++j) {
      const int i = tid + 256 * v + 512 * j;
      if (i < nvec) {
        float a[8];
        tile_sum8<T>(srs, np.n_splits, np.n_rg, p.Mpad, row, i, a);
        const u16x8 rr = np.handoff ? bload_u16x8<16>(rrs, row_off + i * 16)
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
  if ((tid & 63) == 0) {
    red[tid >> 6] = t0;
    red[4 + (tid >> 6)] = t1;
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
      const int i = tid + 256 * v + 512 * j;
      if (i < nvec) {
        const u16x8 wv = *reinterpret_cast<const u16x8*>(np.w + i * 8);
        u16x8 o;
#pragma unroll
        for (int e = 0; e < 8; ++e)
          o[e] = T::from_float(round_to<T>(x[v * MAXV + j][e] * inv) * T::to_float(wv[e]));
        bstore16<16>(o, ors, row_off + i * 16);
      }
    }
  }
}

// silu item of slice (pair_lo, n_pairs): rows_per_item rows x the slice's vectors:
// act = (T)silu(gate) * up from the gate|up slabs
template <typename T>
__device__ __forceinline__ void silu_item(const CParams& p, const Desc& sp, int item, int pair_lo, int n_pairs,
                                          int wait_word, int tid, unsigned long long* tr) {
  uint32_t first = 0;
  if (tid < 64) first = load_word(my_flags(p.sync), wait_word);
  wait_flag(p.sync, wait_word, first, tid, tr);
  const int vec_n = n_pairs * 2;                       // vectors of 8 columns per row in this slice
  const int rows_per_item = max(1, kThreads / vec_n);
  const int nvec = p.inter / 8;
  for (int v = tid; v < rows_per_item * vec_n; v += kThreads) {
    const int row = item * rows_per_item + v / vec_n;
    const int i = pair_lo * 2 + v % vec_n;
    if (row < p.M) {
      const rsrc_t srs = make_rsrc(sp.partial);
      float gte[8], up[8];
      tile_sum8<T>(srs, sp.n_splits, sp.n_rg, p.Mpad, row, i, gte);
      tile_sum8<T>(srs, sp.n_splits, sp.n_rg, p.Mpad, row, nvec + i, up);
      u16x8 rv;
#pragma unroll
      for (int e = 0; e < 8; ++e) rv[e] = T::from_float(round_to<T>(silu_f32(gte[e])) * up[e]);
      bstore16<16>(rv, make_rsrc(sp.out0), ((uint32_t)row * (uint32_t)p.inter + i * 8) * 2);
    }
  }
}

template <typename T>
__global__ __launch_bounds__(kThreads, 2) void decode_chain_kernel(const CParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  int* s_next = reinterpret_cast<int*>(smem + kXBytes);
  if (threadIdx.x == 0)
    *s_next = (int)__hip_atomic_fetch_add(p.sync + L_TICKET * SY_LINE, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  __syncthreads();
  int t = __builtin_amdgcn_readfirstlane(*s_next);
  int si = 0;
#pragma clang loop unroll(disable)
  while (t < p.total) {
    // the thread id is made opaque once per item: everything derived from it (lane constants,
    // addresses) is recomputed inside the item instead of being hoisted out of this loop and
    // kept in (or spilled from) registers across items
    int tid = threadIdx.x;
    asm volatile("" : "+v"(tid));
    while (t >= p.seg[si].end) ++si;                     // tickets only grow: the search resumes
    const Seg& sg = p.seg[si];
    const Desc& ds = p.desc[sg.desc];
    const int item = t - (si ? p.seg[si - 1].end : 0);
    unsigned long long* tr = p.trace ? p.trace + 4 * (int64_t)t : nullptr;
    if (tr && tid == 0) {
      tr[0] = __builtin_amdgcn_s_memrealtime();
      tr[1] = 0;
      tr[3] = (unsigned long long)sg.kind | ((unsigned long long)((sg.desc == D_NORM2 ? 1 : sg.idx) & 15) << 4) |
              ((unsigned long long)(__builtin_amdgcn_s_getreg(20 | (3 << 11)) & 15) << 8);
    }
    if (sg.kind == K_NORM) {
      if (p.hidden / 8 <= 512) norm_item<T, 1>(p, ds, item, sg.wait_word, smem, tid, tr);
      else norm_item<T, 2>(p, ds, item, sg.wait_word, smem, tid, tr);
    } else if (sg.kind == K_SILU) {
      silu_item<T>(p, ds, item, sg.a, sg.b, sg.wait_word, tid, tr);
    } else {
      int split, rg_first, rg_limit, wait_word = sg.wait_word;
      if (sg.kind == K_GEMM_GU) {
        // slice = tile pairs [a, a + b) of the gate rows and of the up rows; items of the slice:
        // (half, group of 4*R tiles, split) with the split fastest
        const int per = kNW * ds.R, ng = (sg.b + per - 1) / per;
        split = item % ds.n_splits;
        const int gq = item / ds.n_splits, half = gq / ng, grp = gq % ng;
        const int base = half * (p.inter >> 4) + sg.a;
        rg_first = base + grp * per;
        rg_limit = base + sg.b;
      } else {
        split = item / ds.gx;                            // split-major: down items of one act slice are consecutive
        rg_first = (item % ds.gx) * (kNW * ds.R);
        rg_limit = ds.N >> 4;
        if (sg.kind == K_GEMM_DOWN) wait_word += split * SY_LINE;   // the silu flag of slice `split`
      }
      switch (ds.R) {
        case 1: gemm_item<T, 1>(p, ds, split, rg_first, rg_limit, wait_word, smem, tid, tr); break;
        case 2: gemm_item<T, 2>(p, ds, split, rg_first, rg_limit, wait_word, smem, tid, tr); break;
        default: gemm_item<T, 4>(p, ds, split, rg_first, rg_limit, wait_word, smem, tid, tr); break;
      }
    }
    // end of item: the next ticket is requested first so that its round trip runs under the
    // drain of this item's write-through stores; then ONE lane signals for the workgroup
    uint32_t nt = 0;
    if (tid == 0)
      nt = __hip_atomic_fetch_add(p.sync + L_TICKET * SY_LINE, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (tid == 0) {
      *s_next = (int)nt;
      if (tr) tr[2] = __builtin_amdgcn_s_memrealtime();
    }
    __syncthreads();
    if (sg.count_word >= 0 && tid == kThreads - 64) signal(p.sync, sg.count_word, sg.flag_word, sg.target);
    t = __builtin_amdgcn_readfirstlane(*s_next);   // rewritten only at the end of the next item, barriers later
  }
}

int g_chain_trace = 0;   // debug: per-item timestamps behind the slabs in the workspace
int g_chain_r[4] = {1, 2, 2, 1};   // weight tiles per wave (1, 2 or 4): o, gate|up, down, qkv (tuning: HX_CHAIN_R / chain_r_*)

}  // namespace

namespace hx {

int chain_set_option(const char* name, int value) {
  if (!strcmp(name, "chain_trace")) { g_chain_trace = value ? 1 : 0; return HX_OK; }
  static const char* names[4] = {"chain_r_o", "chain_r_gu", "chain_r_down", "chain_r_qkv"};
  for (int i = 0; i < 4; ++i)
    if (!strcmp(name, names[i])) {
      if (value != 1 && value != 2 && value != 4) return HX_ERR_SHAPE;
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
  if (a->attn_out_stride % 8) return HX_ERR_STRIDE;
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
        if (r[i] == 1 || r[i] == 2 || r[i] == 4) g_chain_r[i] = r[i];
  }

  CParams p;
  memset(&p, 0, sizeof(p));
  p.M = a->M; p.Mpad = 32; p.hidden = a->hidden; p.inter = a->inter; p.eps = a->eps;
  p.sync = a->sync;
  float* ws = (float*)a->workspace;
  float* slabs_o = ws;
  float* slabs_gu = slabs_o + (int64_t)chain_splits(a->q_size) * p.Mpad * a->hidden;
  float* slabs_dn = slabs_gu + (int64_t)chain_splits(a->hidden) * p.Mpad * 2 * a->inter;
  auto gemm = [&](int di, int ri, const void* w, const void* x, int64_t ldx, float* partial, int N, int K, int tiled) {
    Desc& g = p.desc[di];
    g.w = (const u16*)w; g.x = (const u16*)x; g.partial = partial; g.ldw = K; g.ldx = ldx;
    g.N = N; g.K = K; g.n_splits = chain_splits(K); g.R = g_chain_r[ri];
    g.gx = ((N >> 4) + kNW * g.R - 1) / (kNW * g.R);
    g.tiled = tiled;
    return N ? g.gx * g.n_splits : 0;
  };
  const int items_o = gemm(D_O, 0, a->w_o, a->attn_out, a->attn_out_stride, slabs_o, a->hidden, a->q_size, 1);
  gemm(D_GU, 1, a->w_gate_up, a->x_post, a->hidden, slabs_gu, 2 * a->inter, a->hidden, 1);
  const int items_dn = gemm(D_DOWN, 2, a->w_down, a->act, a->inter, slabs_dn, a->hidden, a->inter, 1);
  const int items_qkv = gemm(D_QKV, 3, a->w_qkv_next, a->x_next, a->hidden, a->qkv_partial, a->qkv_n, a->hidden, 0);
  auto norm = [&](int di, float* slabs, int n_splits, const void* res_in, void* res_out, void* x_out, const void* weight,
                  int handoff) {
    Desc& n = p.desc[di];
    n.partial = slabs; n.x = (const u16*)res_in; n.out0 = (u16*)res_out; n.out1 = (u16*)x_out; n.w = (const u16*)weight;
    n.n_splits = n_splits; n.n_rg = a->hidden >> 4; n.handoff = handoff;
  };
  norm(D_NORM1, slabs_o, p.desc[D_O].n_splits, a->h_in, a->h_mid, a->x_post, a->norm_post_weight, 0);
  norm(D_NORM2, slabs_dn, p.desc[D_DOWN].n_splits, a->h_mid, a->h_out, a->x_next, a->norm_next_weight, 1);
  {
    Desc& sd = p.desc[D_SILU];
    sd.partial = slabs_gu; sd.n_splits = p.desc[D_GU].n_splits; sd.n_rg = (2 * a->inter) >> 4; sd.out0 = (u16*)a->act;
  }

  // segments in ticket (= dependency) order:
  //   o | norm1 | gu[0] gu[1] silu[0] gu[2] silu[1] ... gu[S-1] silu[S-2] silu[S-1] | down | norm2 | qkv
  // (a silu segment sits one gate|up slice behind the slice it reads, so its items rarely wait;
  // down items are split-major and each waits for the silu flag of its own slice)
  const int n_pairs = a->inter >> 4;                         // tile pairs (gate tile, up tile)
  const int n_slices = p.desc[D_DOWN].n_splits;              // K splits of the down projection
  if (n_slices > kMaxSlices || 5 + 2 * n_slices > kMaxSeg) return HX_ERR_SHAPE;
  enum { SLOT_O = 0, SLOT_NORM1 = 1, SLOT_DOWN = 2, SLOT_NORM2 = 3 };
  auto word = [](int line) { return line * SY_LINE; };   // counters: region A; flags: relative to a copy
  int total = 0, ns = 0;
  auto add_seg = [&](int kind, int desc, int idx, int n, int wait_word, int count_line, int flag_line, int sa, int sb) {
    if (n <= 0 || ns >= kMaxSeg) return;
    Seg& sg = p.seg[ns++];
    total += n;
    sg.end = total; sg.kind = kind; sg.desc = desc; sg.idx = idx; sg.wait_word = wait_word;
    sg.count_word = count_line >= 0 ? word(count_line) : -1;
    sg.flag_word = flag_line >= 0 ? word(flag_line) : -1;
    sg.target = n; sg.a = sa; sg.b = sb;
  };
  add_seg(K_GEMM_O, D_O, 0, items_o, -1, L_COUNT + SLOT_O, F_PHASE + SLOT_O, 0, 0);
  add_seg(K_NORM, D_NORM1, 0, a->M, word(F_PHASE + SLOT_O), L_COUNT + SLOT_NORM1, F_PHASE + SLOT_NORM1, 0, 0);
  auto slice_pairs = [&](int j) { const int lo = j * 64; return n_pairs - lo < 64 ? n_pairs - lo : 64; };
  auto add_gu = [&](int j) {
    const int per = kNW * p.desc[D_GU].R, pairs = slice_pairs(j);
    const int n = 2 * ((pairs + per - 1) / per) * p.desc[D_GU].n_splits;
    add_seg(K_GEMM_GU, D_GU, j, n, word(F_PHASE + SLOT_NORM1), L_GCOUNT + j, F_G + j, j * 64, pairs);
  };
  auto add_silu = [&](int j) {
    const int pairs = slice_pairs(j);
    const int rows_per_item = kThreads / (2 * pairs) > 1 ? kThreads / (2 * pairs) : 1;
    add_seg(K_SILU, D_SILU, j, (a->M + rows_per_item - 1) / rows_per_item, word(F_G + j), L_SCOUNT + j, F_S + j,
            j * 64, pairs);
  };
  add_gu(0);
  for (int j = 1; j < n_slices; ++j) { add_gu(j); add_silu(j - 1); }
  add_silu(n_slices - 1);
  add_seg(K_GEMM_DOWN, D_DOWN, 0, items_dn, word(F_S), L_COUNT + SLOT_DOWN, F_PHASE + SLOT_DOWN, 0, 0);
  add_seg(K_NORM, D_NORM2, 1, a->M, word(F_PHASE + SLOT_DOWN), L_COUNT + SLOT_NORM2, F_PHASE + SLOT_NORM2, 0, 0);
  add_seg(K_GEMM_QKV, D_QKV, 0, items_qkv, word(F_PHASE + SLOT_NORM2), -1, -1, 0, 0);
  p.n_seg = ns; p.total = total;

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
  // persistent workgroups: two per CU fit (LDS, registers); more workgroups than tickets are useless
  static int n_cu = 0;
  if (!n_cu) {
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return HX_ERR_HIP;
    n_cu = prop.multiProcessorCount;
  }
  const int grid = total < 2 * n_cu ? total : 2 * n_cu;
  if (a->dtype == HX_F16) hx::launcher(decode_chain_kernel<F16>, grid, kThreads, kLdsBytes, s)(p);
  else hx::launcher(decode_chain_kernel<BF16>, grid, kThreads, kLdsBytes, s)(p);
  int rc = check_launch();
  if (rc) return rc;
  return a->qkv_n ? p.desc[D_QKV].n_splits : 0;
}
