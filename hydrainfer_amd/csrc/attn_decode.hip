// attn_decode.hip — paged decode attention (q_len == 1 per sequence), the HBM-roofline
// kernel of the path (replaces the reference's 64x128-tile split-KV flash kernel run
// with one valid query row: csrc/kernel/flash_attn/src/flash_fwd_kernel.h:435-1019).
//
// Design (gfx950):
//   * grid = (q_head, sequence, kv_split); 4 waves per workgroup, wave w owns KV tiles
//     t = t0+w, t0+w+4, ... of 16 keys each.
//   * Block table: each wave reads the page ids of ALL its tiles with one coalesced
//     vector load (lane j <- page of its j-th tile) and broadcasts them with v_readlane;
//     no dependent table lookup sits in front of a K/V load.
//   * K and V stream HBM -> VGPR with 16-byte loads, double-buffered in registers
//     (tile t+4 is in flight while tile t is computed).  No LDS round trip: every byte
//     is used exactly once by exactly one wave.
//   * Q.K^T on MFMA 16x16x32: A = K tile (16 keys x 32 dims per step, loaded directly in
//     the A-operand lane layout), B = q broadcast into all 16 columns.  The accumulator
//     gives lane (g=l>>4, c=l&15) the 4 scores of keys 4g..4g+3, replicated over c —
//     no cross-lane reduction.
//   * Each 16-lane group g keeps its own online-softmax state (m, l) and an fp32 partial
//     output for its keys; P.V is fp32 FMA on V rows loaded so that lane (g,c) holds
//     dims [c*D/16, (c+1)*D/16) of keys 4g+i — p stays in fp32 (the fp32 CPU oracle's
//     numerics; the reference CUDA kernel rounds P to T first).
//   * The 16 partial states of a workgroup (4 waves x 4 groups) are merged through LDS
//     once; kv_splits > 1 write (m, l, o) partials that attn_decode_combine merges.
//   * RANKED form (round 6; big batches, template flag RANKED).  Measured (tools/probes/probe_cu_balance.hip,
//     tools/ragged_timeline.py): the dispatcher deals the workgroups of a launch round-robin — workgroups w, w + 256,
//     w + 512, w + 768 share a CU — and ONE CU streams at most ~30-32 GB/s, 1.3x its fair share of the HBM rate.  The
//     static (head, sequence) grid — the reference's own partition, flash_fwd_launch_template.h:77 — therefore makes a
//     RAGGED batch (hydrainfer/engine/scheduler.py:99-194 builds one every step) as slow as its heaviest CU: the four
//     sequences that happen to be 256 workgroup numbers apart.  Here the step's sequences are ranked by length ONCE (the
//     RANK DESCRIPTOR: hx_decode_step_head / hx_decode_advance_ranked / hx_decode_rank on the device, or the host that
//     builds the step) and a workgroup takes its (sequence, head) in SNAKE order over the CUs — round 0 the longest to CUs
//     0..255, round 1 the next ones to CUs 255..0, ... — so that every CU gets the same bytes within a few per cent.  No
//     counter, no workspace, no second launch, the same results bit for bit; a batch whose lengths are all within 12.5 %
//     of the mean keeps the static numbering and pays nothing (the descriptor's flag arrives with the sequence's own
//     metadata, one batch of scalar loads).
//     (Tried and dropped in the same round: key ranges dealt from a counter — one word serves ~88 draws / us, and every
//     item's start-up is two or three dependent loads of 8-17 us each under load (a CU keeps ~256 KiB of tile requests
//     queued); cutting outlier sequences with a last-arriver merge — no gain over ranking whole sequences on any batch
//     measured; ranking inside every workgroup from the lengths — four serial rounds of scalar loads, 2 us per launch;
//     page ids through the scalar cache, one per tile two tiles ahead — the s_waitcnt lgkmcnt(0) each use needs also drains
//     the tile's LDS traffic: 72 -> 92 us at 32 x 832 keys.)
#include <cstring>
#include "attn_common.h"

namespace {

using namespace hx;

template <int D> struct VRow;  // per-lane slice of one V row: D/16 elements
template <> struct VRow<64> {
  typedef u16x4 type;
  static constexpr int NV = 1, E = 4;
};
template <> struct VRow<128> {
  typedef u16x8 type;
  static constexpr int NV = 1, E = 8;
};
template <> struct VRow<256> {
  typedef u16x8 type;
  static constexpr int NV = 2, E = 8;  // dims 8c..8c+7 and 128+8c..128+8c+7
};

// Keys per tile.  D <= 128: 16 (lane group g holds keys 4g+i, i = 0..3).  D = 256 (round 5): 8 — lane group g holds keys
// 2g+i, i = 0..1 — so that two register tile buffers are 64 registers instead of 128: the 16-key form of D = 256 needed
// 28..310 spilled registers under its 256-register budget (two workgroups per CU), the 8-key form none.  The MFMA still
// sees 16 A rows: key 2g+i sits in row 4g+i, rows 4g+2, 4g+3 are never written and their scores never read.
template <int D> struct TileGeom {
  static constexpr int KPL = D == 256 ? 2 : 4;      // keys per 16-lane group
  static constexpr int TK = 4 * KPL;                // keys per tile
  static constexpr int SH = D == 256 ? 3 : 4;       // log2(TK)
};

template <int D>
struct KVTile {
  typename VRow<D>::type k[TileGeom<D>::KPL][VRow<D>::NV];  // keys KPL*g+i: dims [c*D/16, (c+1)*D/16)
  typename VRow<D>::type v[TileGeom<D>::KPL][VRow<D>::NV];  // same layout for values
};

// LDS image of one K tile: [16 keys][2*D bytes + 32] — the +32 makes both the row-wise
// ds_write (16 lanes x 16 B contiguous) and the A-operand ds_read_b128 (lane (r,g) <- key r,
// bytes 64s+16g) bank-conflict free.
template <int D> struct KLds { static constexpr int RS = 2 * D + 32, BYTES = 16 * RS; };

template <bool NT, typename V>
__device__ __forceinline__ V ld(const V* p) {
  if (NT) return __builtin_nontemporal_load(p);
  return *p;
}

template <typename T, int D, bool NT>
__device__ __forceinline__ void load_tile(KVTile<D>& buf, const AttnParams& p, const u16* kbase,
                                          const u16* vbase, int page, int row0, int valid,
                                          int lane) {
  // valid = number of in-range keys in this tile (>= 1; may exceed 16)
  const int g = lane >> 4, c = lane & 15;
  constexpr int E = VRow<D>::E, KPL = TileGeom<D>::KPL;
  // every load instruction covers 4 whole key (value) rows of 2*D contiguous bytes: full
  // cache lines, half the line requests of a fragment-shaped (16 rows x 64 B) load
#pragma unroll
  for (int i = 0; i < KPL; ++i) {
    const int tok = min(KPL * g + i, valid - 1);
    const u16* kp = kbase + (int64_t)page * p.k_block_stride + (int64_t)(row0 + tok) * p.k_row_stride +
                    E * c;
#pragma unroll
    for (int n = 0; n < VRow<D>::NV; ++n)
      buf.k[i][n] = ld<NT>(reinterpret_cast<const typename VRow<D>::type*>(kp + 128 * n));
  }
#pragma unroll
  for (int i = 0; i < KPL; ++i) {
    const int tok = min(KPL * g + i, valid - 1);
    const u16* vp = vbase + (int64_t)page * p.v_block_stride + (int64_t)(row0 + tok) * p.v_row_stride +
                    E * c;
#pragma unroll
    for (int n = 0; n < VRow<D>::NV; ++n)
      buf.v[i][n] = ld<NT>(reinterpret_cast<const typename VRow<D>::type*>(vp + 128 * n));
  }
}

// The same tile through BUFFER loads whose resource is the tile's page (round 5): `ok = false` gives the resource zero
// records, every lane is out of range and the instruction touches no memory — a tile load that is always ISSUED, so the
// main loop has no branch around its loads and the compiler's vmcnt counts stay exact (with a branch it waits for the
// tile it has just requested before computing the previous one: tools/decode_timeline.py).
typedef __amdgpu_buffer_rsrc_t drsrc_t;
typedef unsigned int du32x4 __attribute__((__vector_size__(16)));
typedef unsigned int du32x2 __attribute__((__vector_size__(8)));
template <bool NT>
__device__ __forceinline__ u16x8 bld(drsrc_t r, uint32_t off, u16x8) {
  return __builtin_bit_cast(u16x8, __builtin_amdgcn_raw_buffer_load_b128(r, off, 0, NT ? 2 : 0));
}
template <bool NT>
__device__ __forceinline__ u16x4 bld(drsrc_t r, uint32_t off, u16x4) {
  return __builtin_bit_cast(u16x4, __builtin_amdgcn_raw_buffer_load_b64(r, off, 0, NT ? 2 : 0));
}
template <typename T, int D, bool NT>
__device__ __forceinline__ void load_tile_b(KVTile<D>& buf, const AttnParams& p, const u16* kbase, const u16* vbase,
                                            int page, int row0, int valid, int lane, bool ok) {
  const int g = lane >> 4, c = lane & 15;
  constexpr int E = VRow<D>::E, KPL = TileGeom<D>::KPL;
  const uint32_t nrec = ok ? 0x7fffffffu : 0u;
  const drsrc_t kr = __builtin_amdgcn_make_buffer_rsrc(const_cast<u16*>(kbase + (int64_t)page * p.k_block_stride), 0, nrec, 0x00020000);
  const drsrc_t vr = __builtin_amdgcn_make_buffer_rsrc(const_cast<u16*>(vbase + (int64_t)page * p.v_block_stride), 0, nrec, 0x00020000);
  const int vmax = max(valid, 1) - 1;
#pragma unroll
  for (int i = 0; i < KPL; ++i) {
    const uint32_t off = (uint32_t)((row0 + min(KPL * g + i, vmax)) * (int)p.k_row_stride + E * c) * 2u;
#pragma unroll
    for (int n = 0; n < VRow<D>::NV; ++n) buf.k[i][n] = bld<NT>(kr, off + 256u * n, typename VRow<D>::type{});
  }
#pragma unroll
  for (int i = 0; i < KPL; ++i) {
    const uint32_t off = (uint32_t)((row0 + min(KPL * g + i, vmax)) * (int)p.v_row_stride + E * c) * 2u;
#pragma unroll
    for (int n = 0; n < VRow<D>::NV; ++n) buf.v[i][n] = bld<NT>(vr, off + 256u * n, typename VRow<D>::type{});
  }
}

// T-arithmetic rotation of one (x, y) pair, identical to norm_rope_act.hip::rotate_pair
template <typename T>
__device__ __forceinline__ void rope_pair(float x, float y, float c, float s, float& xo, float& yo) {
#pragma clang fp contract(off)
  const float xc = round_to<T>(x * c), ys = round_to<T>(y * s);
  const float xs = round_to<T>(x * s), yc = round_to<T>(y * c);
  xo = round_to<T>(xc - ys);
  yo = round_to<T>(xs + yc);
}

// NeoX rotation of the fragment set f[s] (lane holds dims 32s+8g+j): the partner of dim d is
// d +- D/2, i.e. fragment s +- NS/2 of the SAME lane.  cs points at cos_sin[pos] ([2][D/2]).
template <typename T, int D>
__device__ __forceinline__ void rope_frags(u16x8 (&f)[D / 32], const u16x8 (&rc)[D / 64], const u16x8 (&rs)[D / 64]) {
  constexpr int NS = D / 32, HS = NS / 2;
#pragma unroll
  for (int s = 0; s < HS; ++s) {
    const u16x8 c = rc[s];
    const u16x8 sn = rs[s];
    u16x8 x = f[s], y = f[s + HS];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      float xo, yo;
      rope_pair<T>(T::to_float(x[j]), T::to_float(y[j]), T::to_float(c[j]), T::to_float(sn[j]), xo, yo);
      x[j] = T::from_float(xo);
      y[j] = T::from_float(yo);
    }
    f[s] = x;
    f[s + HS] = y;
  }
}

template <typename T, int D>
__device__ __forceinline__ void compute_tile(const KVTile<D>& buf, const u16x8 (&qf)[D / 32],
                                             int valid, float scale_log2, int lane, float& m,
                                             float& l, float (&o)[D / 16], char* klds) {
  const int g = lane >> 4, c = lane & 15;
  constexpr int RS = KLds<D>::RS, KPL = TileGeom<D>::KPL;
  // K rows -> wave-private LDS -> MFMA A fragments (key r = lane&15, dims 32s+8g..+8).
  // DS operations of one wave execute in order, so no barrier is needed.
#pragma unroll
  for (int i = 0; i < KPL; ++i)
#pragma unroll
    for (int n = 0; n < VRow<D>::NV; ++n)
      *reinterpret_cast<typename VRow<D>::type*>(klds + (4 * g + i) * RS + 256 * n + 2 * VRow<D>::E * c) =
          buf.k[i][n];
  __builtin_amdgcn_wave_barrier();
  f32x4 s = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int st = 0; st < D / 32; ++st) {
    const u16x8 kf = *reinterpret_cast<const u16x8*>(klds + c * RS + 64 * st + 16 * g);
    s = Mfma<T>::mma(kf, qf[st], s);
  }
  __builtin_amdgcn_wave_barrier();
  float x[KPL];
  float mx = m;
#pragma unroll
  for (int i = 0; i < KPL; ++i) {
    x[i] = (KPL * g + i < valid) ? s[i] * scale_log2 : -INFINITY;
    mx = fmaxf(mx, x[i]);
  }
  const float alpha = fast_exp2(m - mx);
  float pr[KPL];
  float ps = 0.f;
#pragma unroll
  for (int i = 0; i < KPL; ++i) {
    pr[i] = fast_exp2(x[i] - mx);
    ps += pr[i];
  }
  m = mx;
  l = l * alpha + ps;
  constexpr int E = VRow<D>::E, NV = VRow<D>::NV;
#pragma unroll
  for (int n = 0; n < NV; ++n)
#pragma unroll
    for (int e = 0; e < E; ++e) {
      float acc = o[n * E + e] * alpha;
#pragma unroll
      for (int i = 0; i < KPL; ++i) acc = fmaf(pr[i], T::to_float(buf.v[i][n][e]), acc);
      o[n * E + e] = acc;
    }
}

// launch bound: 4 workgroups of 4 waves (or 2 of 8) per CU => <= 128 VGPRs for D <= 128; the
// D = 256 instantiation needs more registers and runs at half that occupancy.
template <typename T, int D, int NW, bool NT, bool FUSE, bool RANKED = false>
__global__ __launch_bounds__(NW * 64, (D <= 128 ? 4 : 2)) void attn_decode_kernel(
    const void* __restrict__ h_k, const void* __restrict__ h_v, const int32_t* __restrict__ h_cu_k,
    const int32_t* __restrict__ h_cu_q, const int32_t* __restrict__ h_block_table,
    const int32_t* __restrict__ h_cu_block_lens, const int32_t h_meta, const int32_t h_block_size,
    const int32_t* __restrict__ h_rank_desc, const AttnParams p_in) {
  // Leading scalars (16 dwords) = what the head of the dependent chain  kernarg -> cu_* -> page ids -> first K / V tile
  // needs; they arrive in SGPRs WITH the wave (gemm_xreg.hip, KERNARG PRELOADING).  h_meta = group | n_splits << 8 |
  // (block_shift & 0xff) << 16.  Round 5: the prologue used to be three scalar round trips in a row (the struct, then
  // cu_k, then cu_block_lens + the rest of the struct) with two runtime integer divisions between them; now ONE batch of
  // scalar loads through the preloaded pointers, shifts instead of divisions where the page size is a power of two.
  AttnParams p = p_in;
  p.k = h_k; p.v = h_v; p.cu_k = h_cu_k; p.cu_q = h_cu_q; p.block_table = h_block_table;
  p.cu_block_lens = h_cu_block_lens; p.block_size = h_block_size;
  p.group = h_meta & 0xff; p.n_splits = (h_meta >> 8) & 0xff; p.block_shift = (int)(int8_t)((h_meta >> 16) & 0xff);
  constexpr int OE = D / 16;  // fp32 partial-output elements per lane
  constexpr int NP = NW * 4;  // partial softmax states per workgroup
  __shared__ float s_m[NP], s_l[NP];
  __shared__ float s_o[NP][D];
  __shared__ __attribute__((aligned(16))) char s_k[NW][KLds<D>::BYTES];
  __shared__ __attribute__((aligned(16))) u16 s_qkv[FUSE ? 3 : 1][FUSE ? D : 8];

  int h = blockIdx.x, b = blockIdx.y;      // (RANKED: a flat grid — both are worked out below)
  const int split = blockIdx.z;
#if HX_EXPERIMENTS
  // in-kernel time stamps (100 MHz), 16 per workgroup, wave 0 only: tools/decode_timeline.py
  auto STAMP = [&](int k) {
    if (p_in.stamps && threadIdx.x == 0)
      p_in.stamps[((size_t)(blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x) * 16 + k] = __builtin_amdgcn_s_memrealtime();
  };
#else
  auto STAMP = [&](int) {};
#endif
  STAMP(0);
  const int lane = threadIdx.x & 63;
  const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int g = lane >> 4, c = lane & 15;
  constexpr int KPL = TileGeom<D>::KPL, TK = TileGeom<D>::TK, SH = TileGeom<D>::SH;
  int32_t kv_len, cbl, q_row;
  if (RANKED) {
    // Grid (heads, sequences) like the static form; workgroup number q = blockIdx.y * n_heads + blockIdx.x runs on CU
    // q mod n_cus in the dispatcher's round R = q / n_cus.  The descriptor's flag travels with the metadata of the
    // sequence the STATIC numbering would give this workgroup: an even batch (flag 0) has lost nothing; a ragged one
    // takes item k of the length-ranked list in snake order and fetches that sequence's metadata in a second round.
    int32_t ck0 = h_cu_k[b], ck1 = h_cu_k[b + 1];
    cbl = h_cu_block_lens[b];
    q_row = h_cu_q[b];
    if (h_rank_desc[0]) {
      const uint32_t n_heads = gridDim.x, q = blockIdx.y * n_heads + blockIdx.x, n_items = n_heads * gridDim.y;
      const uint32_t nc = (uint32_t)p_in.n_cus, R = q / nc, cq = q - R * nc, in_round = min(nc, n_items - R * nc);
      const uint32_t k = R * nc + ((R & 1u) ? in_round - 1u - cq : cq);
      const uint32_t rho = k / n_heads;
      h = (int)(k - rho * n_heads);
      b = h_rank_desc[1 + rho];
      ck0 = h_cu_k[b];
      ck1 = h_cu_k[b + 1];
      cbl = h_cu_block_lens[b];
      q_row = h_cu_q[b];
    }
    kv_len = ck1 - ck0;
#if HX_EXPERIMENTS
    if (p_in.stamps && threadIdx.x == 0)      // who this workgroup is: tools/ragged_timeline.py
      p_in.stamps[((size_t)blockIdx.y * gridDim.x + blockIdx.x) * 16 + 8] = 1ull << 40 | (unsigned long long)b << 16 | (unsigned long long)h;
#endif
  } else {
    // the sequence's metadata: one batch of scalar loads
    const int32_t ck0 = h_cu_k[b], ck1 = h_cu_k[b + 1];
    cbl = h_cu_block_lens[b];
    q_row = h_cu_q[b];
    kv_len = ck1 - ck0;
  }
  const int hk = p.group == 1 ? h : h / p.group;
  const int n_tiles = (kv_len + TK - 1) >> SH;
  const int per_split = p.n_splits == 1 ? n_tiles : (n_tiles + p.n_splits - 1) / p.n_splits;
  const int t_begin = split * per_split;
  const int t_end = min(n_tiles, t_begin + per_split);
  const int tpp = p.block_size >> SH;  // tiles per page
  const int tsh = p.block_shift - SH;  // >= 0: a page is 2^tsh tiles (no runtime division on the way to the first load)
  auto page_of = [&](int t) { return tsh >= 0 ? t >> tsh : t / tpp; };
  auto row0_of = [&](int t) { return (tsh >= 0 ? t & ((1 << tsh) - 1) : t % tpp) << SH; };
  const int32_t* bt = h_block_table + cbl;

  const u16* kbase = reinterpret_cast<const u16*>(p.k) + (int64_t)hk * p.k_head_stride;
  const u16* vbase = reinterpret_cast<const u16*>(p.v) + (int64_t)hk * p.v_head_stride;

  // ORDER OF THE FIRST REQUESTS (round 5; in-kernel stamps, tools/decode_timeline.py).  A wave's loads return in the
  // order they were issued.  Rounds 1-4 requested the first K / V tile BEFORE the fused prologue's own small loads (qkv
  // slab, cos / sin): 4096 waves x 8 KiB = 33 MB of HBM traffic stood in front of a few L2 hits in every wave, the
  // prologue ended 11 us (median; up to 44 us) after the kernel's start, and until then every wave had ONE tile in
  // flight — HBM idled.  Now: page ids first, then the prologue's loads, then the first tile; the prologue's arithmetic
  // runs under the tile's latency.
  KVTile<D> bufA, bufB;
  int my_page = 0, n_my = 0;
  int chunk0 = t_begin + w;
  auto request_pages = [&]() {
    const int tj = chunk0 + NW * lane;
    my_page = (tj < t_end) ? bt[page_of(tj)] : 0;
    n_my = max(0, min(64, (t_end - chunk0 + NW - 1) / NW));  // wave-uniform
  };
  // (UNCONDITIONAL in the prologue — under a branch the compiler's wait-count model merges the two paths and makes the
  // prologue wait for the tile it has just requested — but a wave without a tile (a sequence shorter than 16 w keys, a
  // late split) touches no memory: its buffer resource has zero records, as in the main loop.  Round-5 ADVICE: it used to
  // read rows of physical block 0, i.e. relied on block 0 of whatever view the caller passed being mapped.)
  auto request_first_tile = [&]() {
    load_tile_b<T, D, NT>(bufA, p, kbase, vbase, __builtin_amdgcn_readlane(my_page, 0), row0_of(chunk0),
                          kv_len - (chunk0 << SH), lane, chunk0 < t_end);
  };
  auto begin_chunk = [&]() { request_pages(); request_first_tile(); };
  // (D = 256 with the fused prologue: the prologue's q / new-key fragments and the first tile together do not fit the
  // 256-register budget — the tile is requested behind the prologue instead of spilling)
  constexpr bool EARLY = !(FUSE && D == 256);
  STAMP(1);      // scalar metadata in
  if (EARLY) request_pages();

  // q as the MFMA B operand, identical in all 16 columns
  u16x8 qf[D / 32];
  if (!(FUSE && p.qkv_partial)) {
    const u16* qp = reinterpret_cast<const u16*>(p.q) + (int64_t)q_row * p.q_row_stride +
                    (int64_t)h * D + 8 * g;
#pragma unroll
    for (int s = 0; s < D / 32; ++s) qf[s] = *reinterpret_cast<const u16x8*>(qp + 32 * s);
  }
  if (!FUSE) {
    if (EARLY) request_first_tile();
    STAMP(2);
  }

  // FUSE: q and the new token's k arrive un-rotated and the cache does not hold the new token
  // yet.  Rotate both in registers (T arithmetic, same rounding as apply_rotary_pos_emb), use
  // k/v of the new token from registers in its tile, and append them to the cache once.
  u16x8 kn[D / 32];                              // rotated new key, fragment layout (RoPE is lane-local there)
  typename VRow<D>::type knr[VRow<D>::NV];       // the same row in the row layout of KVTile
  typename VRow<D>::type vn[VRow<D>::NV];
  const int t_new = (kv_len - 1) >> SH, r_new = (kv_len - 1) & (TK - 1);
  if (FUSE) {
    // cos / sin of the new token's position: requested FIRST (round 5), so that the position's scalar load and the two
    // vector loads behind it run under the slab reduction and its barrier instead of after them
    const u16* cs = reinterpret_cast<const u16*>(p.cos_sin) + (int64_t)p.positions[b] * D;
    u16x8 rc[D / 64], rs[D / 64];
#pragma unroll
    for (int s = 0; s < D / 64; ++s) {
      rc[s] = *reinterpret_cast<const u16x8*>(cs + 32 * s + 8 * g);
      rs[s] = *reinterpret_cast<const u16x8*>(cs + D / 2 + 32 * s + 8 * g);
    }
    if (p.qkv_partial) {
      // q, k, v of this token/head straight from the qkv GEMM's split-K slabs: thread d < D adds
      // the splits of column d in order and rounds once to T (the projection's output
      // rounding); the three rows are shared through LDS so every slab element is read once
      // per workgroup.  Every thread issues the loads (threads >= D the addresses of thread d - D: no divergent
      // branch around them, so the first tile's request below is not fenced off by a full wait).
      const float* row = p.qkv_partial + (int64_t)b * p.qkv_row;
      const int64_t col0[3] = {(int64_t)h * D, (int64_t)p.n_heads * D + (int64_t)hk * D,
                               (int64_t)p.n_heads * D + (int64_t)(p.n_heads / p.group) * D + (int64_t)hk * D};
      const int dcol = threadIdx.x & (D - 1);
      float sacc[3];
#pragma unroll
      for (int which = 0; which < 3; ++which) sacc[which] = row[col0[which] + dcol];
      if (p.qkv_splits > 1) {
#pragma unroll
        for (int which = 0; which < 3; ++which) {
          const float* src = row + col0[which] + dcol;
          for (int s = 1; s < p.qkv_splits; ++s) sacc[which] += src[s * p.qkv_slab_stride];
        }
      }
      if (EARLY) request_first_tile();      // behind the prologue's loads, in front of their use
      STAMP(2);      // page ids in, first tile requested
      if (threadIdx.x < D) {
#pragma unroll
        for (int which = 0; which < 3; ++which) s_qkv[which][threadIdx.x] = T::from_float(sacc[which]);
      }
      __syncthreads();
#pragma unroll
      for (int s = 0; s < D / 32; ++s) {
        qf[s] = *reinterpret_cast<const u16x8*>(&s_qkv[0][32 * s + 8 * g]);
        kn[s] = *reinterpret_cast<const u16x8*>(&s_qkv[1][32 * s + 8 * g]);
      }
#pragma unroll
      for (int n = 0; n < VRow<D>::NV; ++n)
        vn[n] = *reinterpret_cast<const typename VRow<D>::type*>(&s_qkv[2][128 * n + VRow<D>::E * c]);
    } else {
    const u16* kp = reinterpret_cast<const u16*>(p.k_new) + (int64_t)b * p.kn_row_stride +
                    (int64_t)hk * D + 8 * g;
#pragma unroll
    for (int s = 0; s < D / 32; ++s) kn[s] = *reinterpret_cast<const u16x8*>(kp + 32 * s);
    const u16* vp = reinterpret_cast<const u16*>(p.v_new) + (int64_t)b * p.vn_row_stride +
                    (int64_t)hk * D + VRow<D>::E * c;
#pragma unroll
    for (int n = 0; n < VRow<D>::NV; ++n)
      vn[n] = *reinterpret_cast<const typename VRow<D>::type*>(vp + 128 * n);
    if (EARLY) request_first_tile();
    STAMP(2);
    }
    rope_frags<T, D>(qf, rc, rs);
    rope_frags<T, D>(kn, rc, rs);
    // fragment layout -> row layout through one row of this wave's LDS region
    if (c == 0) {
#pragma unroll
      for (int s = 0; s < D / 32; ++s) *reinterpret_cast<u16x8*>(s_k[w] + 64 * s + 16 * g) = kn[s];
    }
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int n = 0; n < VRow<D>::NV; ++n)
      knr[n] = *reinterpret_cast<const typename VRow<D>::type*>(s_k[w] + 256 * n + 2 * VRow<D>::E * c);
    __builtin_amdgcn_wave_barrier();
    // one writer per kv head and per split set: the first q head of the group, split 0, wave 0
    if (h == hk * p.group && split == 0 && w == 0) {
      const int slot = p.new_slots[b];
      const int64_t blk = slot / p.block_size, off = slot % p.block_size;
      if (c == 0) {   // lanes (r=0, g): 4 x 16 B per step cover the 2*D-byte key row
        u16* kd = const_cast<u16*>(kbase) + blk * p.k_block_stride + off * p.k_row_stride + 8 * g;
#pragma unroll
        for (int s = 0; s < D / 32; ++s) *reinterpret_cast<u16x8*>(kd + 32 * s) = kn[s];
      }
      if (g == 0) {   // lanes (g=0, c): D/16 elements each cover the value row
        u16* vd = const_cast<u16*>(vbase) + blk * p.v_block_stride + off * p.v_row_stride + VRow<D>::E * c;
#pragma unroll
        for (int n = 0; n < VRow<D>::NV; ++n)
          *reinterpret_cast<typename VRow<D>::type*>(vd + 128 * n) = vn[n];
      }
    }
  }
  auto patch = [&](KVTile<D>& buf, int t) {
    if (FUSE && t == t_new) {     // wave-uniform
#pragma unroll
      for (int i = 0; i < KPL; ++i)
        if (KPL * g + i == r_new) {
#pragma unroll
          for (int n = 0; n < VRow<D>::NV; ++n) {
            buf.k[i][n] = knr[n];
            buf.v[i][n] = vn[n];
          }
        }
    }
  };

  STAMP(3);      // prologue done (q / new key rotated, appended)
  float m = HX_NEG_BIG, l = 0.f;
  float o[OE];
#pragma unroll
  for (int e = 0; e < OE; ++e) o[e] = 0.f;

  // my tiles: t_begin + w + NW*j.  Chunks of 64 tiles per wave share one page-id vector.  The loop body is branch-free
  // around its loads: tile j + 1 is ALWAYS requested before tile j is computed (out of range: a buffer load with zero
  // records), tile j + 2 before tile j + 1 is computed — two tiles per wave in flight, exact wait counts.
  auto issue = [&](KVTile<D>& buf, int jj) {
    const int t = chunk0 + NW * jj;
    load_tile_b<T, D, NT>(buf, p, kbase, vbase, __builtin_amdgcn_readlane(my_page, min(jj, 63)), row0_of(t),
                          kv_len - (t << SH), lane, jj < n_my);
  };
  auto run_chunk = [&](bool stamp) {
    for (int j = 0; j < n_my; j += 2) {
      issue(bufB, j + 1);
      patch(bufA, chunk0 + NW * j);
      compute_tile<T, D>(bufA, qf, kv_len - ((chunk0 + NW * j) << SH), p.scale_log2, lane, m, l, o, s_k[w]);
      if (stamp && j == 0) STAMP(4);      // first tile computed
      issue(bufA, j + 2);
      if (j + 1 < n_my) {
        patch(bufB, chunk0 + NW * (j + 1));
        compute_tile<T, D>(bufB, qf, kv_len - ((chunk0 + NW * (j + 1)) << SH), p.scale_log2, lane, m, l, o, s_k[w]);
      }
    }
  };
  if (!EARLY) begin_chunk();
  run_chunk(true);
  for (chunk0 += NW * 64; chunk0 < t_end; chunk0 += NW * 64) {      // contexts past 64 tiles per wave (4096 keys at NW = 4)
    begin_chunk();
    run_chunk(false);
  }
  STAMP(5);      // last tile computed
  // ---- merge the 16 partial states of this workgroup --------------------------------
  constexpr int E = VRow<D>::E, NV = VRow<D>::NV;
  const int slot = w * 4 + g;
  if (c == 0) {
    s_m[slot] = m;
    s_l[slot] = l;
  }
#pragma unroll
  for (int n = 0; n < NV; ++n)
#pragma unroll
    for (int e = 0; e < E; ++e) s_o[slot][128 * n + E * c + e] = o[n * E + e];
  __syncthreads();

  STAMP(6);      // merge barrier passed
  const int d = threadIdx.x;
  if (d < D) {
    float M = HX_NEG_BIG;
#pragma unroll
    for (int k = 0; k < NP; ++k) M = fmaxf(M, s_m[k]);
    float L = 0.f, O = 0.f;
#pragma unroll
    for (int k = 0; k < NP; ++k) {
      const float wgt = fast_exp2(s_m[k] - M);
      L = fmaf(s_l[k], wgt, L);
      O = fmaf(s_o[k][d], wgt, O);
    }
    if (p.n_splits == 1) {
      const float r = (L > 0.f) ? O / L : 0.f;
      reinterpret_cast<u16*>(p.out)[(int64_t)q_row * p.o_row_stride + (int64_t)h * D + d] =
          T::from_float(r);
    } else {
      const int64_t idx = ((int64_t)b * p.n_heads + h) * p.n_splits + split;
      p.ws_o[idx * D + d] = O;
      if (d == 0) {
        p.ws_ml[idx * 2 + 0] = M;
        p.ws_ml[idx * 2 + 1] = L;
      }
    }
  }
  STAMP(7);
}

// one workgroup of D threads per (head, sequence)
template <typename T, int D>
__global__ __launch_bounds__(D) void attn_decode_combine_kernel(const AttnParams p) {
  const int h = blockIdx.x, b = blockIdx.y, d = threadIdx.x;
  const int q_row = p.cu_q[b];
  const int64_t base = ((int64_t)b * p.n_heads + h) * p.n_splits;
  float M = HX_NEG_BIG;
  for (int s = 0; s < p.n_splits; ++s) M = fmaxf(M, p.ws_ml[(base + s) * 2]);
  float L = 0.f, O = 0.f;
  for (int s = 0; s < p.n_splits; ++s) {
    const float wgt = fast_exp2(p.ws_ml[(base + s) * 2] - M);
    L = fmaf(p.ws_ml[(base + s) * 2 + 1], wgt, L);
    O = fmaf(p.ws_o[(base + s) * D + d], wgt, O);
  }
  const float r = (L > 0.f) ? O / L : 0.f;
  reinterpret_cast<u16*>(p.out)[(int64_t)q_row * p.o_row_stride + (int64_t)h * D + d] =
      T::from_float(r);
}

// (sequence, head) pair counts for which one 8-wave workgroup per pair replaces key splits + combine
int g_decode_small_lo = 160, g_decode_small_hi = 576;
int g_decode_waves = 4;  // tuning knobs (hx_debug_set_option)
int g_decode_nt = 1;   // K/V are read once: non-temporal loads measured +4 % (profiles/r1_attn_decode_variants.txt)

inline int32_t decode_meta(const AttnParams& p) {      // attn_decode_kernel's h_meta
  return (p.group & 0xff) | ((p.n_splits & 0xff) << 8) | ((p.block_shift & 0xff) << 16);
}

int g_ranked = 1;             // the RANKED form for big batches (hx_debug_set_option("decode_ranked", 0) = the static grid)

int ranked_n_cus() {
  static int n = [] {
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return 256;
    return prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
  }();
  return n;
}

template <typename T, int D>
int launch_decode(const AttnParams& p, int batch, hipStream_t stream) {
  if (p.group > 255 || p.n_splits > 255) return HX_ERR_SHAPE;
  // big batches, no key split, the step's rank descriptor at hand: (sequence, head) pairs in length-ranked snake order
  if (g_ranked && p.rank_desc && p.n_splits == 1 && (int64_t)batch * p.n_heads >= 768 && D <= 128) {
    AttnParams pp = p;
    pp.n_cus = ranked_n_cus();
    const dim3 grid2(p.n_heads, batch);
    if (p.k_new || p.qkv_partial) hx::launcher(attn_decode_kernel<T, D, 4, true, true, true>, grid2, 256, 0, stream)(pp.k, pp.v, pp.cu_k, pp.cu_q, pp.block_table, pp.cu_block_lens, decode_meta(pp), pp.block_size, pp.rank_desc, pp);
    else hx::launcher(attn_decode_kernel<T, D, 4, true, false, true>, grid2, 256, 0, stream)(pp.k, pp.v, pp.cu_k, pp.cu_q, pp.block_table, pp.cu_block_lens, decode_meta(pp), pp.block_size, pp.rank_desc, pp);
    return check_launch();
  }
  dim3 grid(p.n_heads, batch, p.n_splits);
  // 160..576 (sequence, head) pairs and no key split: 8 waves per workgroup share the keys
  // (see decode_pick_splits)
  const int64_t pairs = (int64_t)batch * p.n_heads;
  const bool wide = pairs >= g_decode_small_lo && pairs <= g_decode_small_hi && p.n_splits == 1;
  if (p.k_new || p.qkv_partial) {
    if (wide) hx::launcher(attn_decode_kernel<T, D, 8, true, true>, grid, 512, 0, stream)(p.k, p.v, p.cu_k, p.cu_q, p.block_table, p.cu_block_lens, decode_meta(p), p.block_size, (const int32_t*)nullptr, p);
    else hx::launcher(attn_decode_kernel<T, D, 4, true, true>, grid, 256, 0, stream)(p.k, p.v, p.cu_k, p.cu_q, p.block_table, p.cu_block_lens, decode_meta(p), p.block_size, (const int32_t*)nullptr, p);
  } else if (g_decode_waves == 8 || wide) {
    if (g_decode_nt) hx::launcher(attn_decode_kernel<T, D, 8, true, false>, grid, 512, 0, stream)(p.k, p.v, p.cu_k, p.cu_q, p.block_table, p.cu_block_lens, decode_meta(p), p.block_size, (const int32_t*)nullptr, p);
    else hx::launcher(attn_decode_kernel<T, D, 8, false, false>, grid, 512, 0, stream)(p.k, p.v, p.cu_k, p.cu_q, p.block_table, p.cu_block_lens, decode_meta(p), p.block_size, (const int32_t*)nullptr, p);
  } else {
    if (g_decode_nt) hx::launcher(attn_decode_kernel<T, D, 4, true, false>, grid, 256, 0, stream)(p.k, p.v, p.cu_k, p.cu_q, p.block_table, p.cu_block_lens, decode_meta(p), p.block_size, (const int32_t*)nullptr, p);
    else hx::launcher(attn_decode_kernel<T, D, 4, false, false>, grid, 256, 0, stream)(p.k, p.v, p.cu_k, p.cu_q, p.block_table, p.cu_block_lens, decode_meta(p), p.block_size, (const int32_t*)nullptr, p);
  }
  int rc = check_launch();
  if (rc) return rc;
  if (p.n_splits > 1) {
    hx::launcher(attn_decode_combine_kernel<T, D>, dim3(p.n_heads, batch), D, 0, stream)(p);
    rc = check_launch();
  }
  return rc;
}

}  // namespace

namespace hx {

int decode_set_option(const char* name, int value) {
  if (!strcmp(name, "decode_waves")) { g_decode_waves = (value == 8) ? 8 : 4; return HX_OK; }
  if (!strcmp(name, "decode_nt")) { g_decode_nt = value ? 1 : 0; return HX_OK; }
  if (!strcmp(name, "decode_small_lo")) { g_decode_small_lo = value; return HX_OK; }
  if (!strcmp(name, "decode_small_hi")) { g_decode_small_hi = value; return HX_OK; }
  if (!strcmp(name, "decode_ranked")) { g_ranked = value ? 1 : 0; return HX_OK; }
  return HX_ERR_UNSUPPORTED;
}

bool decode_supported(int head_dim) { return head_dim == 64 || head_dim == 128 || head_dim == 256; }

// final pass over split-KV partials in AttnParams::ws_o / ws_ml (shared with attn_decode_gqa.hip)
int launch_decode_combine(const AttnParams& p, int batch, int head_dim, int dtype, hipStream_t stream) {
  const dim3 grid(p.n_heads, batch);
#define HX_COMBINE_CASE(TT, DD) \
  case DD: hx::launcher(attn_decode_combine_kernel<TT, DD>, grid, DD, 0, stream)(p); break;
  if (dtype == HX_F16) {
    switch (head_dim) { HX_COMBINE_CASE(F16, 64) HX_COMBINE_CASE(F16, 128) HX_COMBINE_CASE(F16, 256) default: return HX_ERR_SHAPE; }
  } else if (dtype == HX_BF16) {
    switch (head_dim) { HX_COMBINE_CASE(BF16, 64) HX_COMBINE_CASE(BF16, 128) HX_COMBINE_CASE(BF16, 256) default: return HX_ERR_SHAPE; }
  } else {
    return HX_ERR_DTYPE;
  }
#undef HX_COMBINE_CASE
  return check_launch();
}

int decode_pick_splits(int batch, int n_heads, int max_seqlen_k, int requested) {
  if (requested >= 1) return requested > 128 ? 128 : requested;
  const int64_t base = (int64_t)batch * n_heads;
  if (base >= 768) return 1;
  const int n_tiles = (max_seqlen_k + 15) / 16;
  // mid-size batches, context <= 2048 keys: one 8-wave workgroup per (sequence, head) and no
  // combine launch.  Measured in the 7B decode graph (bench.py --batch N, whole step): batch 6 / 8 /
  // 12 / 16: -2.5 / -2.0 / -3.4 / -1.9 %; batch <= 4 or >= 20: the split (resp. 4-wave) form wins
  if (base >= g_decode_small_lo && base <= g_decode_small_hi && n_tiles <= 128) return 1;
  int64_t want = (1024 + base - 1) / base;       // ~4 workgroups per CU
  int64_t cap = n_tiles / 16;                    // >= 16 tiles (4 per wave) per split
  if (cap < 1) cap = 1;
  int64_t s = want < cap ? want : cap;
  if (s > 64) s = 64;
  return (int)s;
}

int launch_attn_decode(const AttnParams& p, int batch, int head_dim, int dtype,
                       hipStream_t stream) {
  if (dtype == HX_F16) {
    switch (head_dim) {
      case 64: return launch_decode<F16, 64>(p, batch, stream);
      case 128: return launch_decode<F16, 128>(p, batch, stream);
      case 256: return launch_decode<F16, 256>(p, batch, stream);
    }
  } else if (dtype == HX_BF16) {
    switch (head_dim) {
      case 64: return launch_decode<BF16, 64>(p, batch, stream);
      case 128: return launch_decode<BF16, 128>(p, batch, stream);
      case 256: return launch_decode<BF16, 256>(p, batch, stream);
    }
  } else {
    return HX_ERR_DTYPE;
  }
  return HX_ERR_SHAPE;
}

}  // namespace hx
