// debug_bench.hip — read-streaming microbenchmarks behind hx_debug_stream_read: which access shape
// reaches which fraction of the HBM peak on this GPU.  Not on the product path; used by
// tools/bench_stream.py to choose the load shapes of the weight-streaming and decode-attention
// kernels (MI355X_MICROARCH.md quotes 6.3-6.8 TB/s for read streams; the library GEMM of the
// lm_head reaches ~7 TB/s in the decode graph, profiles/r2_base_timeline.md).
#include "hx_common.h"

namespace {

using namespace hx;

// variant 0: contiguous — wave instruction = 64 lanes x 16 B = 1 KiB contiguous; U instructions in
//            flight per wave; consecutive instructions of a wave are 1 KiB apart.
// variant 1: 8 rows x 128 B per instruction, row pitch `pitch` bytes (the GEMM weight shape)
// variant 2: 4 rows x 256 B per instruction
// variant 3: 2 rows x 512 B per instruction
// variant 4: 1 row x 1024 B per instruction, rows pitch apart (the decode-attention K/V shape is
//            4 rows x 256 B = variant 2 with pitch 8 KiB)
template <int U, int POLICY>
__global__ __launch_bounds__(256) void stream_read_kernel(const char* __restrict__ base, int64_t bytes,
                                                          int variant, int64_t pitch, float* sink) {
  const int lane = threadIdx.x & 63;
  const int wave = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  const int n_waves = gridDim.x * (blockDim.x >> 6);
  const int chunk = U * 1024;                              // bytes per wave per step
  const int n_chunks = (int)(bytes / chunk);
  // everything that does not depend on the chunk is computed once, in 32 bits: the address arithmetic of
  // the strided variants must not cost more issue slots than that of the contiguous one (an earlier
  // version divided 64-bit integers per chunk and under-reported the strided shapes by ~8 %)
  const int S = variant == 0 ? 1024 : variant == 1 ? 128 : variant == 2 ? 256 : variant == 3 ? 512 : 1024;
  const int R = 1024 / S;
  const int lanes_per_row = S / 16;
  const int r = lane / lanes_per_row, piece = lane % lanes_per_row;
  const int per_rowblock = variant ? (int)(pitch / ((int64_t)S * U)) : 1;   // chunks per row block
  const int64_t lane_off = variant ? (int64_t)r * pitch + piece * 16 : lane * 16;
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  for (int c = wave; c < n_chunks; c += n_waves) {
    int64_t base_off;
    if (variant == 0) {
      base_off = (int64_t)c * chunk;
    } else {
      // the chunk is a [R rows][S*U bytes] block of a row-major matrix with row pitch `pitch`: one
      // instruction covers R rows x S bytes (R*S = 1024), U instructions walk along the rows
      const int rb = c / per_rowblock, cb = c - rb * per_rowblock;
      base_off = (int64_t)rb * R * pitch + (int64_t)cb * (S * U);
    }
    const char* p0 = base + base_off + lane_off;
    f32x4 v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const f32x4* p = reinterpret_cast<const f32x4*>(p0 + (variant ? u * S : u * 1024));
      if (POLICY == 1) v[u] = __builtin_nontemporal_load(p);
      else v[u] = *p;
    }
#pragma unroll
    for (int u = 0; u < U; ++u) acc += v[u];
  }
  if (acc[0] + acc[1] + acc[2] + acc[3] == 123.456f) sink[0] = acc[0];
}

// variant 5: the contiguous shape by LDS-DMA (global_load_lds_dwordx4: no VGPR write-back), U pieces of 1 KiB in
//            flight per wave in a ring of U LDS slots, counted vmcnt; nothing reads the LDS
template <int U, int POLICY>
__global__ __launch_bounds__(256) void stream_read_lds_kernel(const char* __restrict__ base, int64_t bytes) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int wave = blockIdx.x * 4 + w, n_waves = gridDim.x * 4;
  const int chunk = U * 1024;
  const int n_chunks = (int)(bytes / chunk);
  const uint32_t lds = (uint32_t)(size_t)(__attribute__((address_space(3))) char*)(smem) + (uint32_t)(w * U * 1024);
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Winline-asm"
  for (int c = wave; c < n_chunks; c += n_waves) {
    const char* p0 = base + (int64_t)c * chunk + lane * 16;
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const uint32_t d = __builtin_amdgcn_readfirstlane(lds + 1024u * u);
      if (POLICY == 1)
        asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off nt" :: "v"(p0 + 1024 * u), "s"(d) : "memory", "m0");
      else
        asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" :: "v"(p0 + 1024 * u), "s"(d) : "memory", "m0");
      // the slot written U pieces ago must have landed before it is overwritten: at most U - 1 older pieces pending
      asm volatile("s_waitcnt vmcnt(%0)" :: "n"(U - 1) : "memory");
    }
  }
#pragma clang diagnostic pop
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

template <int U>
int launch_u(const void* p, int64_t bytes, int variant, int64_t pitch, int policy, int wgs, float* sink, hipStream_t s) {
  if (variant == 5) {
    const size_t lds = 4 * U * 1024;
    if (policy) {
      hipError_t e = hipFuncSetAttribute((const void*)stream_read_lds_kernel<U, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      if (e != hipSuccess) return hip_rc(e);
      hx::launcher(stream_read_lds_kernel<U, 1>, wgs, 256, lds, s)((const char*)p, bytes);
    } else {
      hipError_t e = hipFuncSetAttribute((const void*)stream_read_lds_kernel<U, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      if (e != hipSuccess) return hip_rc(e);
      hx::launcher(stream_read_lds_kernel<U, 0>, wgs, 256, lds, s)((const char*)p, bytes);
    }
    return check_launch();
  }
  if (policy) hx::launcher(stream_read_kernel<U, 1>, wgs, 256, 0, s)((const char*)p, bytes, variant, pitch, sink);
  else hx::launcher(stream_read_kernel<U, 0>, wgs, 256, 0, s)((const char*)p, bytes, variant, pitch, sink);
  return check_launch();
}

// The decode-attention access pattern with the arithmetic removed: grid (head, sequence), 4 waves; wave w
// owns the 16-key tiles w, w + 4, ...; a tile is one page of the paged cache (block_size 16): its K and V
// rows for this head are 16 x 256 B at a pitch of row_bytes inside the page; pages come from a table
// (random order).  Two tiles in flight per wave (register double buffer) like attn_decode_kernel.
// What this reads per launch is exactly what the attention kernel reads: its rate is that kernel's ceiling.
template <int HPW, int NW, int DEPTH>   // heads per workgroup (contiguous 256*HPW bytes per key row), waves, tiles in flight per wave
__global__ __launch_bounds__(NW * 64) void paged_read_kernel(const char* __restrict__ kbase, const char* __restrict__ vbase,
                                                             const int32_t* __restrict__ table, int tiles, int n_splits,
                                                             int64_t page_bytes, int row_bytes, float* sink) {
  // one wave instruction = 1 KiB = (64 / (16 * HPW)) key rows x (256 * HPW) bytes; a 16-key tile of K is
  // 4 * HPW instructions, of V as many
  constexpr int LPR = 16 * HPW;            // lanes per key row
  constexpr int RPI = 64 / LPR;            // key rows per instruction
  constexpr int NI = 16 / RPI;             // instructions per tile and tensor
  const int hg = blockIdx.x, b = blockIdx.y, split = blockIdx.z;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int64_t lane_off = (int64_t)(lane / LPR) * row_bytes + (int64_t)hg * (256 * HPW) + (lane % LPR) * 16;
  const int32_t* tb = table + (int64_t)b * tiles;
  const int per = (tiles + n_splits - 1) / n_splits;
  const int t0 = split * per, t1 = min(tiles, t0 + per);
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  f32x4 buf[DEPTH][2 * NI];
  auto load = [&](f32x4 (&bf)[2 * NI], int t) {
    const int64_t off = (int64_t)tb[t] * page_bytes + lane_off;
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      bf[i] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(kbase + off + (int64_t)i * RPI * row_bytes));
      bf[NI + i] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(vbase + off + (int64_t)i * RPI * row_bytes));
    }
  };
#pragma unroll
  for (int d = 0; d < DEPTH - 1; ++d)
    if (t0 + w + d * NW < t1) load(buf[d], t0 + w + d * NW);
  for (int t = t0 + w; t < t1; t += NW * DEPTH) {
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) {
      const int tc = t + d * NW, tn = tc + (DEPTH - 1) * NW;
      if (tn < t1) load(buf[(d + DEPTH - 1) % DEPTH], tn);
      if (tc < t1) {
#pragma unroll
        for (int i = 0; i < 2 * NI; ++i) acc += buf[d][i];
      }
    }
  }
  if (acc[0] + acc[1] + acc[2] + acc[3] == 123.456f) sink[0] = acc[0];
}

// The same bytes with HG adjacent heads served by DIFFERENT waves of one workgroup (wave w: head hg*HG + w % HG, tile
// phase w / HG), every wave instruction still 4 key rows x 256 B like attn_decode_kernel: do waves of one CU that ask
// for neighbouring 256-byte pieces of the same key rows at about the same time stream faster than workgroups
// scattered over the chip?  row_bytes == 256 asks for head-major pages ([page][head][16 rows][256 B]: one head's piece
// of a page contiguous).  Measured (ctx 832, 436 MB): 1 head x 4 phases 68.5 us, 4 x 4 67.4, 2 x 8 67.4; head-major
// pages 66.5 / 66.1 / 65.9 — 1-3 %, not worth a cache-layout change.
template <int HG, int NP>
__global__ __launch_bounds__(HG * NP * 64) void paged_read_hg_kernel(const char* __restrict__ kbase, const char* __restrict__ vbase,
                                                                     const int32_t* __restrict__ table, int tiles,
                                                                     int64_t page_bytes, int row_bytes, float* sink) {
  const int hgrp = blockIdx.x, b = blockIdx.y;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int head = hgrp * HG + (w % HG), ph = w / HG;
  const int64_t head_bytes = row_bytes == 256 ? 16 * 256 : 256;
  const int64_t lane_off = (int64_t)(lane >> 4) * row_bytes + (int64_t)head * head_bytes + (lane & 15) * 16;
  const int32_t* tb = table + (int64_t)b * tiles;
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  f32x4 buf[2][8];
  auto load = [&](f32x4 (&bf)[8], int t) {
    const int64_t off = (int64_t)tb[t] * page_bytes + lane_off;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      bf[i] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(kbase + off + (int64_t)i * 4 * row_bytes));
      bf[4 + i] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(vbase + off + (int64_t)i * 4 * row_bytes));
    }
  };
  if (ph < tiles) load(buf[0], ph);
  for (int t = ph; t < tiles; t += 2 * NP) {
    if (t + NP < tiles) load(buf[1], t + NP);
#pragma unroll
    for (int i = 0; i < 8; ++i) acc += buf[0][i];
    if (t + NP < tiles) {
      if (t + 2 * NP < tiles) load(buf[0], t + 2 * NP);
#pragma unroll
      for (int i = 0; i < 8; ++i) acc += buf[1][i];
    }
  }
  if (acc[0] + acc[1] + acc[2] + acc[3] == 123.456f) sink[0] = acc[0];
}

}  // namespace

extern "C" int hx_debug_paged_read(const void* kbase, const void* vbase, const int32_t* table, int n_seq, int n_heads,
                                   int tiles, int64_t page_bytes, int row_bytes, int heads_per_wg, int waves,
                                   int depth, int n_splits, float* sink, hx_stream stream) {
  if (!kbase || !vbase || !table || !sink || n_seq <= 0 || n_heads <= 0 || tiles <= 0 || n_splits <= 0) return HX_ERR_NULL;
  hipStream_t s = (hipStream_t)stream;
  if (heads_per_wg < 0) {       // -HG: HG adjacent heads by different waves of one workgroup, `waves` / HG tile phases each
    const int HG = -heads_per_wg;
    if (n_heads % HG || n_splits != 1) return HX_ERR_SHAPE;
    const dim3 g2(n_heads / HG, n_seq);
#define HX_PH(HGv, NPv)                                                                                              \
    if (HG == HGv && waves == HGv * NPv) {                                                                           \
      hx::launcher(paged_read_hg_kernel<HGv, NPv>, g2, HGv * NPv * 64, 0, s)((const char*)kbase, (const char*)vbase, table, tiles, \
                                                                           page_bytes, row_bytes, sink);             \
      return check_launch();                                                                                         \
    }
    HX_PH(4, 4) HX_PH(2, 4) HX_PH(4, 2) HX_PH(8, 2) HX_PH(2, 8) HX_PH(1, 4) HX_PH(2, 2) HX_PH(4, 1)
#undef HX_PH
    return HX_ERR_SHAPE;
  }
  if (n_heads % heads_per_wg) return HX_ERR_SHAPE;
  const dim3 grid(n_heads / heads_per_wg, n_seq, n_splits);
#define HX_P(HPW, NW, DP)                                                                                          \
  if (heads_per_wg == HPW && waves == NW && depth == DP) {                                                         \
    hx::launcher(paged_read_kernel<HPW, NW, DP>, grid, NW * 64, 0, s)((const char*)kbase, (const char*)vbase, table, tiles, n_splits, \
                                                            page_bytes, row_bytes, sink);                          \
    return check_launch();                                                                                         \
  }
  HX_P(1, 4, 2) HX_P(1, 4, 3) HX_P(1, 4, 4) HX_P(1, 8, 2)
  HX_P(2, 4, 2) HX_P(2, 4, 3) HX_P(4, 4, 2) HX_P(4, 4, 3) HX_P(4, 8, 2) HX_P(2, 8, 2)
#undef HX_P
  return HX_ERR_SHAPE;
}

extern "C" int hx_debug_stream_read(const void* p, int64_t bytes, int variant, int64_t pitch, int unroll,
                                    int policy, int wgs, float* sink, hx_stream stream) {
  if (!p || !sink || bytes <= 0 || wgs <= 0) return HX_ERR_NULL;
  if (variant < 0 || variant > 5 || (variant && variant < 5 && (pitch <= 0 || pitch % 1024))) return HX_ERR_SHAPE;
  hipStream_t s = (hipStream_t)stream;
  switch (unroll) {
    case 4: return launch_u<4>(p, bytes, variant, pitch, policy, wgs, sink, s);
    case 8: return launch_u<8>(p, bytes, variant, pitch, policy, wgs, sink, s);
    case 16: return launch_u<16>(p, bytes, variant, pitch, policy, wgs, sink, s);
    case 32: return launch_u<32>(p, bytes, variant, pitch, policy, wgs, sink, s);
    default: return HX_ERR_SHAPE;
  }
}
