// gemm_skinny.hip — weight-streaming GEMM for decode batches (M <= 64 rows):
//     partial[s][m][n] = sum_{k in split s} x[m][k] * W[n][k]         (fp32 slabs)
// i.e. y = x @ W^T for the nn.Linear weights of the decoder layers
// (hydrainfer/model/llama.py:24-27,48-50) at decode batch sizes, where the library GEMM
// reaches only 3.5-4.7 TB/s of weight streaming on MI355X (profiles/r1_bench7b_*).
//
// HBM-bound design (every weight byte is read exactly once, by one wave):
//   * grid = (N tiles, K splits); 8 waves per workgroup.  The workgroup's slice of x
//     ([M_pad][<=1024 k], <= 66 KB) is staged ONCE into LDS (row stride KR*2+32 B: the
//     ds_read_b128 of the B fragments is bank-conflict free), then never touched again.
//   * each wave streams whole 16-row groups of W HBM -> VGPR with FULL-LINE loads (one
//     instruction = 8 rows x 128 B; a fragment-shaped 16 rows x 64 B load issues twice the line
//     requests and measured ~20 % slower), 16 k-steps (16 KiB) per register buffer, two buffers
//     in flight, non-temporal.  Each 2-k-step column block is transposed into the MFMA
//     A-operand layout through a 2 KiB wave-private, XOR-swizzled LDS image.
//   * MFMA 16x16x32: A = W fragment, B = x^T fragment from LDS (one per 16 batch rows), so a
//     weight fragment is used for M/16 MFMAs; the accumulator holds out^T[n][m].
//   * K is split across workgroups (<= 1024 k each) to have >= 2 workgroups per CU in flight;
//     partial sums go to fp32 slabs that the consumer kernels (finalize / fused epilogues)
//     add in a fixed order — deterministic, no atomics.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include "attn_common.h"

namespace {

using namespace hx;

struct GemmParams {
  const void* x;
  const void* w;
  float* partial;   // [S][M][N]
  int64_t ldx, ldw; // row strides in elements
  int32_t M, N, K;
  int32_t ks_per_split;  // k-steps (of 32) per split, multiple of 8
  int32_t n_splits;
  int32_t packed;        // w is in fragment order (hx_pack_decode_weight)
};

constexpr int kChunk = 16;       // k-steps per register buffer (16 KiB of W per wave)
constexpr int kMaxKs = 32;       // k-steps per split (KR <= 1024) = 2 chunks
constexpr int kRS = kMaxKs * 64 + 32;   // LDS row stride in bytes

// R = 16-row groups per wave (each 2 chunks); NW = waves per workgroup.  The per-wave work is
// a compile-time constant so the load/compute schedule below is straight-line code and the
// compiler's vmcnt counts are exact (a loop makes it fall back to vmcnt(0) and serialises the
// two buffers).
template <typename T, int MB, int R, int NW>
__global__ __launch_bounds__(NW * 64) void gemm_skinny_kernel(const GemmParams p, const int g_nt_store) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int kThreads = NW * 64;
  const int lane = threadIdx.x & 63;
  const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int r = lane & 15, g = lane >> 4, c = lane & 15;

  const int split = blockIdx.y;
  const int total_ks = p.K >> 5;
  const int ks0 = split * p.ks_per_split;
  const int nks = min(p.ks_per_split, total_ks - ks0);   // multiple of 8, <= 32
  const int KR = nks << 5;
  const int n_rg_all = p.N >> 4;
  // wave's row groups: rg0 + j*NW, j < R (groups past the end are clamped for loads and
  // skipped at the store)
  const int rg0 = blockIdx.x * (NW * R) + w;

  // ---- 1. x slice -> registers.  The LDS image always spans kMaxKs k-steps; k beyond this
  // split's range and rows beyond M are zero, so a short last split needs no predication.
  constexpr int kCpr = kMaxKs * 4;                         // 16-byte chunks per LDS row
  constexpr int XPT = MB * 16 * kCpr / kThreads;           // chunks per thread
  u16x8 xr[XPT];
  {
    const u16* xb = reinterpret_cast<const u16*>(p.x) + (int64_t)ks0 * 32;
#pragma unroll
    for (int j = 0; j < XPT; ++j) {
      const int i = threadIdx.x + j * kThreads;
      const int row = i / kCpr, ch = i % kCpr;
      const bool ok = row < p.M && ch * 8 < KR;
      // always a valid address; the zero-fill select happens at the LDS write so that no
      // loaded value is consumed before the weight prefetch has been issued
      xr[j] = *reinterpret_cast<const u16x8*>(xb + (int64_t)(ok ? row : 0) * p.ldx + (ok ? ch * 8 : 0));
    }
  }

  // ---- 2. W prefetch.  Load layout: instruction j of a chunk covers rows 8*(j&1) + (lane>>3)
  // and the 128-byte column block j>>1 (2 k-steps); lane&7 selects the 16-byte piece.
  const int lrow = lane >> 3, lpiece = lane & 7;
  const u16* wb = reinterpret_cast<const u16*>(p.w) + (int64_t)ks0 * 32 + 8 * lpiece;
  auto load = [&](u16x8 (&buf)[kChunk], int it) {
    const int rgi = it >> 1, ch = it & 1;
    const int n0 = min(rg0 + rgi * NW, n_rg_all - 1) << 4;
    const int last_cb = ((nks - ch * kChunk) >> 1) - 1;   // column blocks past the range re-read a valid one
    const u16* wp = wb + (int64_t)(n0 + lrow) * p.ldw + ch * (kChunk * 32);
#pragma unroll
    for (int j = 0; j < kChunk; ++j) {
      const int cb = max(min(j >> 1, last_cb), -ch * (kChunk / 2));
      buf[j] = __builtin_nontemporal_load(
          reinterpret_cast<const u16x8*>(wp + (int64_t)(8 * (j & 1)) * p.ldw + 64 * cb));
    }
  };
  u16x8 buf[2][kChunk];
  load(buf[0], 0);
  load(buf[1], 1);
  // keep the 32 KiB weight prefetch in flight UNDER the x staging: without this fence hipcc
  // sinks the weight loads below the LDS writes (one exposed L2 round trip per workgroup)
  __builtin_amdgcn_sched_barrier(0);

  // ---- 3. x slice -> LDS
#pragma unroll
  for (int j = 0; j < XPT; ++j) {
    const int i = threadIdx.x + j * kThreads;
    const int row = i / kCpr, ch = i % kCpr;
    const bool ok = row < p.M && ch * 8 < KR;
    *reinterpret_cast<u16x8*>(smem + row * kRS + ch * 16) = ok ? xr[j] : u16x8{0, 0, 0, 0, 0, 0, 0, 0};
  }
  __syncthreads();

  // wave-private transpose image: [16 rows][128 B], 16-byte slot s of row r stored at slot
  // s ^ ((r >> 1) & 7): row writes (8 lanes x 16 B per row, two rows per 16-lane group) and
  // A-fragment reads (lane (r,g) <- row r, slot 4*st+g) are both bank-conflict free.
  char* tl = smem + MB * 16 * kRS + w * 2048;
  const int wr_off0 = lrow * 128 + 16 * (lpiece ^ ((lrow >> 1) & 7));
  const int wr_off1 = (lrow + 8) * 128 + 16 * (lpiece ^ (((lrow + 8) >> 1) & 7));
  const char* xl = smem + c * kRS + g * 16;
  f32x4 acc[MB];
#pragma unroll
  for (int mb = 0; mb < MB; ++mb) acc[mb] = f32x4{0.f, 0.f, 0.f, 0.f};

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
      const int rg = rg0 + rgi * NW;
#pragma unroll
      for (int mb = 0; mb < MB; ++mb) {
        const int m = mb * 16 + c;
        if (m < p.M && rg < n_rg_all) {
          f32x4* dst = reinterpret_cast<f32x4*>(p.partial + ((int64_t)split * p.M + m) * p.N + (rg << 4) + 4 * g);
          if (g_nt_store) __builtin_nontemporal_store(acc[mb], dst);
          else *dst = acc[mb];
        }
        acc[mb] = f32x4{0.f, 0.f, 0.f, 0.f};
      }
    }
  }
}

// The same product with the weights PACKED in MFMA-fragment order (hx_pack_decode_weight): the
// 1 KiB block (row group rg, k-step s) holds, at lane (r = l & 15, g = l >> 4), the 16 bytes
// W[16 rg + r][32 s + 8 g .. + 8] — exactly one A operand.  A wave reads its (row group, split) as
// ONE contiguous 32 KiB run, 1 KiB per instruction, straight into the MFMA operand registers: no
// transpose through LDS (tools/bench_stream.py, corrected: every load shape streams at 6.5-7.0 TB/s when
// neighbouring waves cover neighbouring bytes; what the packed form saves is the per-wave LDS transpose).  Same k order, same
// accumulation chains: bit-identical to gemm_skinny_kernel.
template <typename T, int MB, int R, int NW, int DBG = 0>
__global__ __launch_bounds__(NW * 64) void gemm_packed_kernel(const void* __restrict__ h_x, const void* __restrict__ h_w,
                                                              float* __restrict__ h_partial, const int64_t h_ldx,
                                                              const int32_t h_M, const int32_t h_N, const int32_t h_K,
                                                              const int32_t h_ks_per_split, const int g_nt_store,
                                                              const GemmParams p_in) {
  // leading scalars = what the first loads need, preloaded into SGPRs with the wave (gemm_xreg.hip, KERNARG PRELOADING)
  GemmParams p = p_in;
  p.x = h_x; p.w = h_w; p.partial = h_partial; p.ldx = h_ldx; p.M = h_M; p.N = h_N; p.K = h_K; p.ks_per_split = h_ks_per_split;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int kThreads = NW * 64;
  const int lane = threadIdx.x & 63;
  const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int g = lane >> 4, c = lane & 15;
  const int split = blockIdx.y;
  const int total_ks = p.K >> 5;
  const int ks0 = split * p.ks_per_split;
  const int nks = min(p.ks_per_split, total_ks - ks0);
  const int KR = nks << 5;
  const int n_rg_all = p.N >> 4;
  const int rg0 = blockIdx.x * (NW * R) + w;

  constexpr int kCpr = kMaxKs * 4;
  constexpr int XPT = MB * 16 * kCpr / kThreads;
  u16x8 xr[XPT];
  {
    const u16* xb = reinterpret_cast<const u16*>(p.x) + (int64_t)ks0 * 32;
#pragma unroll
    for (int j = 0; j < XPT; ++j) {
      const int i = threadIdx.x + j * kThreads;
      const int row = i / kCpr, ch = i % kCpr;
      const bool ok = row < p.M && ch * 8 < KR;
      if (DBG & 4) xr[j] = u16x8{1, 2, 3, 4, 5, 6, 7, 8};   // ablation: no x loads
      else xr[j] = *reinterpret_cast<const u16x8*>(xb + (int64_t)(ok ? row : 0) * p.ldx + (ok ? ch * 8 : 0));
    }
  }
  // the fragments of (split, rg) are one run of nks KiB at KiB offset ks0 * n_rg + rg * nks: the
  // row groups of a split are adjacent, so the waves of a launch sweep consecutive memory (a
  // layout with each row group's whole K contiguous puts concurrent waves 128 KiB apart — all on
  // the same channels: measured 2x slower).  k-steps past the split's range re-read its last one
  // (x is zero there)
  const u16* wb = reinterpret_cast<const u16*>(p.w) + 8 * lane;
  auto load = [&](u16x8 (&buf)[kChunk], int it) {
    const int rgi = it >> 1, ch = it & 1;
    const int rg = min(rg0 + rgi * NW, n_rg_all - 1);
    const u16* wp = wb + ((int64_t)ks0 * n_rg_all + (int64_t)rg * nks) * 512;
#pragma unroll
    for (int j = 0; j < kChunk; ++j) {
      const int s = min(ch * kChunk + j, nks - 1);
      buf[j] = __builtin_nontemporal_load(reinterpret_cast<const u16x8*>(wp + (int64_t)s * 512));
    }
  };
  u16x8 buf[2][kChunk];
  load(buf[0], 0);
  load(buf[1], 1);
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int j = 0; j < XPT; ++j) {
    const int i = threadIdx.x + j * kThreads;
    const int row = i / kCpr, ch = i % kCpr;
    const bool ok = row < p.M && ch * 8 < KR;
    *reinterpret_cast<u16x8*>(smem + row * kRS + ch * 16) = ok ? xr[j] : u16x8{0, 0, 0, 0, 0, 0, 0, 0};
  }
  __syncthreads();

  const char* xl = smem + c * kRS + g * 16;
  f32x4 acc[MB];
#pragma unroll
  for (int mb = 0; mb < MB; ++mb) acc[mb] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int it = 0; it < 2 * R; ++it) {
    const int rgi = it >> 1, ch = it & 1;
    const char* xp = xl + ch * (kChunk * 64);
#pragma unroll
    for (int j = 0; j < kChunk; ++j) {
#pragma unroll
      for (int mb = 0; mb < MB; ++mb) {
        if (DBG & 8) {   // ablation: no LDS reads, no MFMA
          acc[mb][0] += __builtin_bit_cast(float, (uint32_t)buf[it & 1][j][0] << 16);
          continue;
        }
        const u16x8 xf = *reinterpret_cast<const u16x8*>(xp + mb * 16 * kRS + j * 64);
        acc[mb] = Mfma<T>::mma(buf[it & 1][j], xf, acc[mb]);
      }
    }
    // pin the refill behind this chunk's MFMAs: left to itself the scheduler hoists the next
    // loads above them (new registers for every in-flight fragment) and spills
    __builtin_amdgcn_sched_barrier(0);
    if (it + 2 < 2 * R) load(buf[it & 1], it + 2);
    __builtin_amdgcn_sched_barrier(0);
    if (ch == 1) {
      const int rg = rg0 + rgi * NW;
#pragma unroll
      for (int mb = 0; mb < MB; ++mb) {
        const int m = mb * 16 + c;
        if (m < p.M && rg < n_rg_all && !((DBG & 2) && acc[mb][0] != 123.25f)) {   // bit 1: ablation, no stores
          f32x4* dst = reinterpret_cast<f32x4*>(p.partial + ((int64_t)split * p.M + m) * p.N + (rg << 4) + 4 * g);
          if (g_nt_store & 1) __builtin_nontemporal_store(acc[mb], dst);
          else *dst = acc[mb];
        }
        acc[mb] = f32x4{0.f, 0.f, 0.f, 0.f};
      }
    }
  }
}

// fragment (rg, s) of split = s / kMaxKs lives at KiB index ks0 * n_rg + rg * nks + (s - ks0)
// (ks0 = split * kMaxKs, nks = k-steps of that split); inside it lane l holds
// W[16 rg + (l & 15)][32 s + 8 (l >> 4) .. + 8]
__global__ __launch_bounds__(256) void pack_weight_kernel(u16* __restrict__ packed, const u16* __restrict__ w,
                                                          int64_t n_pieces, int total_ks, int n_rg, int64_t ldw) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;   // 16-byte piece index of the OUTPUT
  if (i >= n_pieces) return;
  const int lane = (int)(i & 63);
  const int64_t blk = i >> 6;                                   // KiB index
  const int64_t per_split = (int64_t)kMaxKs * n_rg;
  const int split = (int)(blk / per_split);
  const int ks0 = split * kMaxKs;
  const int nks = min(kMaxKs, total_ks - ks0);
  const int64_t rem = blk - (int64_t)ks0 * n_rg;
  const int64_t rg = rem / nks;
  const int s = ks0 + (int)(rem % nks);
  const u16* src = w + (16 * rg + (lane & 15)) * ldw + 32 * s + 8 * (lane >> 4);
  *reinterpret_cast<u16x8*>(packed + i * 8) = *reinterpret_cast<const u16x8*>(src);
}

// out[m][n] = (T) sum_s partial[s][m][n]   (fixed summation order)
template <typename T>
__global__ __launch_bounds__(256) void slab_reduce_kernel(const float* __restrict__ partial,
                                                          u16* __restrict__ out, int64_t mn,
                                                          int n_splits, int64_t N, int64_t ldo) {
  const int64_t i4 = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4;
  if (i4 >= mn) return;
  f32x4 a = *reinterpret_cast<const f32x4*>(partial + i4);
  for (int s = 1; s < n_splits; ++s) a += *reinterpret_cast<const f32x4*>(partial + s * mn + i4);
  const int64_t m = i4 / N, n = i4 - m * N;
  u16x4 r;
#pragma unroll
  for (int e = 0; e < 4; ++e) r[e] = T::from_float(a[e]);
  *reinterpret_cast<u16x4*>(out + m * ldo + n) = r;
}

int g_slab_nt = 0;       // tuning: non-temporal slab stores
int g_force_r = 0;       // tuning: 0 = automatic, else row groups per wave
int g_force_nw = 0;      // tuning: 0 = automatic, else waves per workgroup (4 or 8)

}  // namespace

namespace hx {

int gemm_set_option(const char* name, int value) {
  if (!strcmp(name, "gemm_rows_per_wave")) { g_force_r = value; return HX_OK; }
  if (!strcmp(name, "gemm_waves")) { g_force_nw = value; return HX_OK; }
  if (!strcmp(name, "gemm_slab_nt")) { g_slab_nt = value; return HX_OK; }
  return HX_ERR_UNSUPPORTED;
}

bool gemm_skinny_supported(int64_t M, int64_t N, int64_t K, int64_t ldx, int64_t ldw) {
  return M >= 1 && M <= 64 && N % 16 == 0 && K % 256 == 0 && ldx % 8 == 0 && ldw % 8 == 0;
}

int gemm_skinny_splits(int64_t K) {
  const int total_ks = (int)(K >> 5);
  return (total_ks + kMaxKs - 1) / kMaxKs;
}

template <typename T, int MB, int R, int NW>
int launch_gemm_cfg(const GemmParams& p, hipStream_t stream) {
  const int n_rg = p.N >> 4;
  dim3 grid((unsigned)((n_rg + NW * R - 1) / (NW * R)), (unsigned)p.n_splits);
  if (p.packed) {
    const size_t plds = (size_t)MB * 16 * kRS;   // x slice only
    if (plds > 48 * 1024) {
      hipError_t e = hipFuncSetAttribute((const void*)gemm_packed_kernel<T, MB, R, NW>,
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)plds);
      if (e != hipSuccess) return hip_rc(e);
    }
    if (MB == 2 && (g_slab_nt >> 1)) {   // ablation variants (tools/gemm_ablate.py), batch 17..32 only
      if ((g_slab_nt >> 1) == 1) hx::launcher(gemm_packed_kernel<T, MB, R, NW, 2>, grid, NW * 64, plds, stream)(p.x, p.w, p.partial, p.ldx, p.M, p.N, p.K, p.ks_per_split, g_slab_nt & 1, p);
      else hx::launcher(gemm_packed_kernel<T, MB, R, NW, 14>, grid, NW * 64, plds, stream)(p.x, p.w, p.partial, p.ldx, p.M, p.N, p.K, p.ks_per_split, g_slab_nt & 1, p);
      return check_launch();
    }
    hx::launcher(gemm_packed_kernel<T, MB, R, NW>, grid, NW * 64, plds, stream)(p.x, p.w, p.partial, p.ldx, p.M, p.N, p.K, p.ks_per_split, g_slab_nt & 1, p);
    return check_launch();
  }
  const size_t lds = (size_t)MB * 16 * kRS + (size_t)NW * 2048;   // x slice + per-wave transpose images
  if (lds > 48 * 1024) {
    hipError_t e = hipFuncSetAttribute((const void*)gemm_skinny_kernel<T, MB, R, NW>,
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return hip_rc(e);
  }
  hx::launcher(gemm_skinny_kernel<T, MB, R, NW>, grid, NW * 64, lds, stream)(p, g_slab_nt);
  return check_launch();
}

template <typename T, int MB>
int launch_gemm_mb(const GemmParams& p, hipStream_t stream) {
  // pick (waves, row groups per wave) so that the launch has ~256..640 workgroups: all
  // resident at 2 per CU, every CU busy
  const int64_t units = (int64_t)(p.N >> 4) * p.n_splits;   // (row group, split) pairs
  // measured cold-weight on MI355X (profiles/r1_gemm_variants.txt): 4-wave workgroups with one
  // row group per wave win up to ~1000 workgroups; beyond that two row groups per wave
  int nw = 4;
  int rpw = 1;
  while (rpw < 4 && units / ((int64_t)nw * rpw) > 1024) ++rpw;
  // in-pipeline sweep on the 7B shapes (HX_GEMM_CFG, 96 decode steps each): the widest
  // projection (gate|up, 5504 units) and the many-split one (down, 11 K splits) prefer 8-wave
  // workgroups with 3 resp. 2 row groups per wave; qkv / o keep 4 waves x 1
  if (units > 4096) { nw = 8; rpw = 3; }
  else if (p.n_splits >= 8) { nw = 8; rpw = 2; }
  // 33..64 batch rows: the x slice alone is 133 KB of LDS, so one workgroup per CU whatever its
  // size — four waves cannot keep the CU streaming.  8 waves everywhere; cold-weight timing at
  // M = 48 / 64 (tools/bench_gemm_m.py): qkv 51 -> 31 us with two row groups per wave, o 21.5 -> 18.5
  // (round 5: up to 1024 units — the o projection — 8 waves x 1 row group are 128 workgroups, half the CUs idle; 4-wave
  //  workgroups fill the chip: 12.1 -> 10.9 us inside the 64-row step)
  if (MB == 4) {
    nw = units <= 1024 ? 4 : 8;
    if (units <= 4096 && p.n_splits < 8) rpw = units >= 2048 ? 2 : 1;
  }
  // per-shape override for tuning runs: HX_GEMM_CFG="N:K:R:NW;N:K:R:NW;..."
  static const char* nt_env = getenv("HX_GEMM_SLAB_NT");
  if (nt_env) g_slab_nt = atoi(nt_env);
  static const char* cfg = getenv("HX_GEMM_CFG");
  if (cfg) {
    const char* q = cfg;
    while (*q) {
      long n = 0, k = 0, r = 0, w = 0;
      if (sscanf(q, "%ld:%ld:%ld:%ld", &n, &k, &r, &w) == 4 && n == p.N && k == p.K) {
        if (r >= 1 && r <= 4) rpw = (int)r;
        if (w == 4 || w == 8) nw = (int)w;
      }
      const char* semi = strchr(q, ';');
      if (!semi) break;
      q = semi + 1;
    }
  }
  if (g_force_nw == 4 || g_force_nw == 8) nw = g_force_nw;
  if (g_force_r >= 1 && g_force_r <= 4) rpw = g_force_r;
  if (nw == 4) {
    switch (rpw) {
      case 1: return launch_gemm_cfg<T, MB, 1, 4>(p, stream);
      case 2: return launch_gemm_cfg<T, MB, 2, 4>(p, stream);
      case 3: return launch_gemm_cfg<T, MB, 3, 4>(p, stream);
      default: return launch_gemm_cfg<T, MB, 4, 4>(p, stream);
    }
  }
  switch (rpw) {
    case 1: return launch_gemm_cfg<T, MB, 1, 8>(p, stream);
    case 2: return launch_gemm_cfg<T, MB, 2, 8>(p, stream);
    case 3: return launch_gemm_cfg<T, MB, 3, 8>(p, stream);
    default: return launch_gemm_cfg<T, MB, 4, 8>(p, stream);
  }
}

// partial must hold splits*M*N floats
int launch_gemm_skinny(const void* x, const void* w, float* partial, int64_t M, int64_t N,
                       int64_t K, int64_t ldx, int64_t ldw, int dtype, hipStream_t stream, int packed = 0) {
  if (!gemm_skinny_supported(M, N, K, ldx, ldw)) return HX_ERR_SHAPE;
  if (!aligned16(x) || !aligned16(w) || !aligned16(partial)) return HX_ERR_STRIDE;
  GemmParams p;
  p.x = x; p.w = w; p.partial = partial; p.ldx = ldx; p.ldw = ldw;
  p.M = (int)M; p.N = (int)N; p.K = (int)K;
  p.ks_per_split = kMaxKs;
  p.n_splits = gemm_skinny_splits(K);
  p.packed = packed;
  const int MB = (int)((M + 15) / 16);
  if (dtype == HX_F16) {
    if (MB == 1) return launch_gemm_mb<F16, 1>(p, stream);
    if (MB == 2) return launch_gemm_mb<F16, 2>(p, stream);
    return launch_gemm_mb<F16, 4>(p, stream);
  }
  if (dtype == HX_BF16) {
    if (MB == 1) return launch_gemm_mb<BF16, 1>(p, stream);
    if (MB == 2) return launch_gemm_mb<BF16, 2>(p, stream);
    return launch_gemm_mb<BF16, 4>(p, stream);
  }
  return HX_ERR_DTYPE;
}

int launch_slab_reduce(const float* partial, void* out, int64_t M, int64_t N, int64_t ldo,
                       int n_splits, int dtype, hipStream_t stream) {
  const int64_t mn = M * N;
  const unsigned blocks = (unsigned)((mn / 4 + 255) / 256);
  if (dtype == HX_F16)
    hx::launcher(slab_reduce_kernel<F16>, blocks, 256, 0, stream)(partial, (u16*)out, mn, n_splits, N, ldo);
  else if (dtype == HX_BF16)
    hx::launcher(slab_reduce_kernel<BF16>, blocks, 256, 0, stream)(partial, (u16*)out, mn, n_splits, N, ldo);
  else
    return HX_ERR_DTYPE;
  return check_launch();
}

}  // namespace hx

using namespace hx;

extern "C" int64_t hx_linear_decode_workspace_bytes(int64_t M, int64_t N, int64_t K) {
  if (M <= 0 || N <= 0 || K <= 0) return 0;
  return (int64_t)gemm_skinny_splits(K) * M * N * (int64_t)sizeof(float);
}

extern "C" int hx_linear_decode(void* out, const void* x, const void* weight, int64_t M,
                                int64_t N, int64_t K, int64_t ldx, int64_t ldw, int64_t ldo,
                                void* workspace, int64_t workspace_bytes, int dtype,
                                hx_stream stream) {
  if (M <= 0 || N <= 0 || K <= 0) return HX_ERR_SHAPE;
  if (!out || !x || !weight || !workspace) return HX_ERR_NULL;
  if (!gemm_skinny_supported(M, N, K, ldx, ldw) || ldo % 4 != 0) return HX_ERR_SHAPE;
  if (workspace_bytes < hx_linear_decode_workspace_bytes(M, N, K)) return HX_ERR_WORKSPACE;
  hipStream_t s = (hipStream_t)stream;
  int rc = launch_gemm_skinny(x, weight, (float*)workspace, M, N, K, ldx, ldw, dtype, s);
  if (rc) return rc;
  return launch_slab_reduce((const float*)workspace, out, M, N, ldo, gemm_skinny_splits(K), dtype, s);
}

// GEMM only: leaves the split-K slabs for a fused consumer (hx_add_rms_norm_slabs,
// hx_silu_and_mul_slabs, hx_decode_attention_fused).  Returns the number of slabs (>= 1) or a
// negative status.
extern "C" int hx_linear_decode_partial(float* partial, const void* x, const void* weight,
                                        int64_t M, int64_t N, int64_t K, int64_t ldx, int64_t ldw,
                                        int64_t partial_bytes, int dtype, hx_stream stream) {
  if (M <= 0 || N <= 0 || K <= 0) return HX_ERR_SHAPE;
  if (!partial || !x || !weight) return HX_ERR_NULL;
  if (!gemm_skinny_supported(M, N, K, ldx, ldw)) return HX_ERR_SHAPE;
  if (partial_bytes < hx_linear_decode_workspace_bytes(M, N, K)) return HX_ERR_WORKSPACE;
  int rc = launch_gemm_skinny(x, weight, partial, M, N, K, ldx, ldw, dtype, (hipStream_t)stream);
  if (rc) return rc;
  return gemm_skinny_splits(K);
}

// ---- packed weights ----------------------------------------------------------------------------
extern "C" int hx_pack_decode_weight(void* packed, const void* weight, int64_t N, int64_t K, int64_t ldw,
                                     int dtype, hx_stream stream) {
  if (!packed || !weight) return HX_ERR_NULL;
  if (N <= 0 || K <= 0 || N % 16 || K % 32 || ldw % 8) return HX_ERR_SHAPE;
  if (dtype != HX_F16 && dtype != HX_BF16) return HX_ERR_DTYPE;
  if (!aligned16(packed) || !aligned16(weight)) return HX_ERR_STRIDE;
  const int64_t n_pieces = N * K / 8;
  hx::launcher(pack_weight_kernel, (unsigned)((n_pieces + 255) / 256), 256, 0, (hipStream_t)stream)(
      (u16*)packed, (const u16*)weight, n_pieces, (int)(K >> 5), (int)(N >> 4), ldw);
  return check_launch();
}

extern "C" int hx_linear_decode_partial_packed(float* partial, const void* x, const void* packed_weight,
                                               int64_t M, int64_t N, int64_t K, int64_t ldx,
                                               int64_t partial_bytes, int dtype, hx_stream stream) {
  if (M <= 0 || N <= 0 || K <= 0) return HX_ERR_SHAPE;
  if (!partial || !x || !packed_weight) return HX_ERR_NULL;
  if (!gemm_skinny_supported(M, N, K, ldx, K)) return HX_ERR_SHAPE;
  if (partial_bytes < hx_linear_decode_workspace_bytes(M, N, K)) return HX_ERR_WORKSPACE;
  int rc = launch_gemm_skinny(x, packed_weight, partial, M, N, K, ldx, K, dtype, (hipStream_t)stream, 1);
  if (rc) return rc;
  return gemm_skinny_splits(K);
}
