// attn_fwd.hip — general variable-length attention forward (prefill, chunked prefill,
// dense vision-tower MHA, and any decode shape the specialised kernel does not take).
// Replaces csrc/kernel/flash_attn/src/flash_fwd_kernel.h:26-431 (dense) and :435-1019
// (paged) of the reference; online-softmax recipe as softmax.h:95-145, mask as
// mask.h:130-211 (bottom-right aligned causal).
//
// Design (gfx950, MFMA 16x16x32, "swapped" products so no P transpose is needed):
//   * workgroup = 4 waves x QR x 16 query rows (QR = 2, an option: every K / V fragment read from
//     LDS then feeds two MFMAs); grid = (ceil(max_q/(64 QR)), head, sequence), renumbered so that
//     the query tiles of one head run on one XCD (shared L2), longest causal tiles first.  The 32-key K and V tiles are staged ONCE per workgroup (coalesced
//     16-byte loads of whole rows, register prefetch of tile t+1 under tile t's MFMAs,
//     double-buffered LDS images, one barrier per tile) and shared by the four waves.
//   * the running output is rescaled only when some row's maximum moved (wave-uniform test):
//     after the first tiles it rarely does, and multiplying by exactly 1 is the identity.
//   * S^T[key][query] = K . Q^T : A = K fragments read from the LDS image (row stride 2D+32 B:
//     conflict free), B = Q^T fragments kept in registers.
//     The accumulator leaves lane (g=l>>4, c=l&15) with query c, keys 4g..4g+3 of each
//     16-key sub-tile: row statistics need only two cross-group shuffles (xor 16, 32).
//   * O^T[dim][query] += V^T . P^T : P^T is used as the B operand straight from the
//     softmax registers (k-slot j of lane group g <-> key 16*(j>>2)+4g+(j&3)); V^T is the
//     A operand, read with ds_read_b64_tr_b16 from the shared LDS image of the V tile
//     [32 keys][D] (row stride 2D+32 bytes => conflict-free transposed reads).
//   * P is rounded to T before P.V exactly like the reference kernel
//     (flash_fwd_kernel.h:878); accumulation is fp32.
#include <algorithm>
#include <cstring>
#include <type_traits>
#include "attn_common.h"

namespace {

using namespace hx;

template <typename T, int D, bool PAGED, int QR, int KU>
__global__ __launch_bounds__(256) void attn_fwd_kernel(const AttnParams p) {
  constexpr int NS = D / 32;       // QK k-steps
  constexpr int NDB = D / 16;      // 16-dim output blocks
  constexpr int RS = 2 * D + 32;   // LDS row stride in bytes (K and V images)
  constexpr int LPR = D / 8;       // 16-byte chunks per key row
  constexpr int KT = 32 * KU;      // keys per tile (KU = 2: half the barriers and softmax passes, and
                                   // four independent score sub-tiles per wave to hide MFMA / exp latency)
  constexpr int TILE_CHUNKS = KT * LPR;
  constexpr int NL = (TILE_CHUNKS + 255) / 256;   // chunks per thread per tile
  constexpr int TILE_BYTES = KT * RS;
  extern __shared__ __attribute__((aligned(16))) char smem[];   // K[2][KT][RS] | V[2][KT][RS]

  // Workgroup -> (sequence, query tile, head).  The grid is COMPACT: x runs over tile slots, at most
  // total_q / tile_rows + batch of them, and each workgroup finds the sequence that owns its slot
  // by walking the cumulative lengths (a launch sized (max_q tiles) x batch would, for one 2017-token
  // chunk next to 31 decode rows, dispatch 30 752 workgroups that exit at once — measured 847 us for
  // work that takes 190 us).  Hardware deals consecutive workgroups round-robin to the 8 XCDs, each
  // with its own L2; the query tiles of one (sequence, head) stream the same K / V, so ids are
  // renumbered to put them on ONE XCD (ids congruent mod 8 form a contiguous range of slots).
  // Within a sequence the causal tiles get longer with the row index: the long ones go first.
  constexpr int WROWS = 16 * QR;             // query rows per wave
  constexpr int TQ = 4 * WROWS;              // query rows per workgroup
  int mblk, h, b;
  {
    const int gx = gridDim.x, gy = gridDim.y;
    const int total = gx * gy;
    int wg = blockIdx.x + gx * blockIdx.y;
    if (p.xcd_remap && total % 8 == 0) wg = (wg % 8) * (total / 8) + wg / 8;
    int slot = wg % gx;
    h = wg / gx;
    b = 0;
    int tiles = 0;
    for (; b < p.batch; ++b) {
      tiles = (p.cu_q[b + 1] - p.cu_q[b] + TQ - 1) / TQ;
      if (slot < tiles) break;
      slot -= tiles;
    }
    if (b == p.batch) return;                // spare slot (the grid is an upper bound)
    mblk = tiles - 1 - slot;
  }
  const int lane = threadIdx.x & 63;
  const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int g = lane >> 4, c = lane & 15;
  const int hk = h / p.group;

  const int q_start = p.cu_q[b];
  const int q_len = p.cu_q[b + 1] - q_start;
  const int k_start = p.cu_k[b];
  const int kv_len = p.cu_k[b + 1] - k_start;
  const int q_row0_wg = mblk * TQ;
  if (q_row0_wg >= q_len) return;            // workgroup-uniform
  const int q_row0 = q_row0_wg + w * WROWS;  // may exceed q_len for the last workgroup's waves:
                                             // those waves still load / synchronise, never store

  char* kbuf = smem;
  char* vbuf = smem + 2 * TILE_BYTES;
  const u16* kbase = reinterpret_cast<const u16*>(p.k) + (int64_t)hk * p.k_head_stride;
  const u16* vbase = reinterpret_cast<const u16*>(p.v) + (int64_t)hk * p.v_head_stride;
  const int32_t* bt = PAGED ? p.block_table + p.cu_block_lens[b] : nullptr;

  // Q^T fragments (B operand): lane (c,g) holds Q[row 16 rb + c][32s + 8g + j]
  u16x8 qf[QR][NS];
#pragma unroll
  for (int rb = 0; rb < QR; ++rb) {
    const int qr = min(q_row0 + 16 * rb + c, q_len - 1);
    const u16* qp = reinterpret_cast<const u16*>(p.q) + (int64_t)(q_start + qr) * p.q_row_stride +
                    (int64_t)h * D + 8 * g;
#pragma unroll
    for (int s = 0; s < NS; ++s) qf[rb][s] = *reinterpret_cast<const u16x8*>(qp + 32 * s);
  }

  const int shift = kv_len - q_len;
  // visible keys of query row i: [i + shift - window_left, i + shift + window_right] (mask.h:173-193);
  // causal = (unbounded, 0); no mask = both unbounded
  const bool local = p.window_left >= 0;             // the dispatcher sets both sides for a local call
  const int wr = p.causal ? 0 : local ? p.window_right : 0x3fffffff;
  const int wl = local ? p.window_left : 0x3fffffff;
  int limit_c[QR], first_c[QR];
#pragma unroll
  for (int rb = 0; rb < QR; ++rb) {
    const int row = q_row0 + 16 * rb + c + shift;
    limit_c[rb] = (int)min((int64_t)kv_len - 1, (int64_t)row + wr);
    first_c[rb] = (int)max((int64_t)0, (int64_t)row - wl);
  }
  // a wave whose rows all lie past the sequence (q_len = 1: three of the four) only helps staging
  const int last_key_wave = q_row0 >= q_len ? -1 : (int)min((int64_t)kv_len - 1, (int64_t)q_row0 + WROWS - 1 + shift + wr);
  const int last_key_wg =
      (int)min((int64_t)kv_len - 1, (int64_t)min(q_row0_wg + 4 * WROWS - 1, q_len - 1) + shift + wr);
  const int first_key_wg = (int)max((int64_t)0, (int64_t)q_row0_wg + shift - wl);
  const int first_key_wave = (int)max((int64_t)0, (int64_t)q_row0 + shift - wl);
  const int t_first = local ? first_key_wg / KT : 0;                   // tiles left of the window are skipped
  const int n_tiles = (last_key_wg >= 0) ? last_key_wg / KT + 1 : 0;   // workgroup-uniform

  // ---- cooperative tile staging: thread owns chunks idx = tid + 256*j of the [32][D] tile
  int my_row[NL], my_chunk[NL];
#pragma unroll
  for (int j = 0; j < NL; ++j) {
    const int idx = threadIdx.x + 256 * j;
    my_row[j] = idx / LPR;
    my_chunk[j] = idx % LPR;
  }
  // element offset of key row `key` given its page (paged) — the page id is looked up ONE TILE
  // AHEAD of the loads that need it, so a tile's loads are a single round trip, not two
  auto page_of = [&](int key) -> int {
    return PAGED ? bt[page_slot(min(key, kv_len - 1), p.block_size, p.block_shift)] : 0;
  };
  auto key_offset = [&](int key, int page, bool is_v) -> int64_t {
    key = min(key, kv_len - 1);
    if (PAGED) {
      const int row = page_row(key, p.block_size, p.block_shift);
      return is_v ? (int64_t)page * p.v_block_stride + (int64_t)row * p.v_row_stride
                  : (int64_t)page * p.k_block_stride + (int64_t)row * p.k_row_stride;
    }
    return (int64_t)(k_start + key) * (is_v ? p.v_row_stride : p.k_row_stride);
  };
  u16x8 kreg[NL], vreg[NL];
  int page_next[NL];      // pages of the tile that will be loaded next
  auto lookup_pages = [&](int t) {
#pragma unroll
    for (int j = 0; j < NL; ++j) page_next[j] = page_of(t * KT + my_row[j]);
  };
  auto load_tile = [&](int t) {
#pragma unroll
    for (int j = 0; j < NL; ++j) {
      if (TILE_CHUNKS % 256 == 0 || threadIdx.x + 256 * j < TILE_CHUNKS) {
        const int key = t * KT + my_row[j];
        kreg[j] = *reinterpret_cast<const u16x8*>(kbase + key_offset(key, page_next[j], false) + 8 * my_chunk[j]);
        vreg[j] = *reinterpret_cast<const u16x8*>(vbase + key_offset(key, page_next[j], true) + 8 * my_chunk[j]);
      }
    }
  };
  auto store_tile = [&](int buf) {
#pragma unroll
    for (int j = 0; j < NL; ++j) {
      if (TILE_CHUNKS % 256 == 0 || threadIdx.x + 256 * j < TILE_CHUNKS) {
        const int off = buf * TILE_BYTES + my_row[j] * RS + my_chunk[j] * 16;
        *reinterpret_cast<u16x8*>(kbuf + off) = kreg[j];
        *reinterpret_cast<u16x8*>(vbuf + off) = vreg[j];
      }
    }
  };

  f32x4 acc[QR][NDB];
  float m[QR], l[QR];
#pragma unroll
  for (int rb = 0; rb < QR; ++rb) {
#pragma unroll
    for (int i = 0; i < NDB; ++i) acc[rb][i] = f32x4{0.f, 0.f, 0.f, 0.f};
    m[rb] = HX_NEG_BIG;
    l[rb] = 0.f;
  }

  if (n_tiles > t_first) {
    lookup_pages(t_first);
    load_tile(t_first);
    lookup_pages(t_first + 1);
    store_tile(t_first & 1);
  }
  __syncthreads();

  const int q4 = c >> 2, p4 = c & 3;
  for (int t = t_first; t < n_tiles; ++t) {
    const int cur = t & 1;
    if (t + 1 < n_tiles) {
      load_tile(t + 1);                          // in flight under this tile's MFMAs
      lookup_pages(t + 2);                       // (clamped to the last key) for the next iteration
    }

    if (t * KT <= last_key_wave && t * KT + KT - 1 >= first_key_wave) {   // wave-uniform: tiles outside this wave's band
      const char* kt = kbuf + cur * TILE_BYTES;
      const char* vt = vbuf + cur * TILE_BYTES;
      // ---- S^T = K . Q^T for the 2 KU 16-key sub-tiles (A fragments from the shared K image,
      //      each used for the QR row blocks of this wave)
      f32x4 s[QR][2 * KU];
#pragma unroll
      for (int u = 0; u < 2 * KU; ++u) {
#pragma unroll
        for (int rb = 0; rb < QR; ++rb) s[rb][u] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int st = 0; st < NS; ++st) {
          const u16x8 kf = *reinterpret_cast<const u16x8*>(kt + (16 * u + c) * RS + 64 * st + 16 * g);
#pragma unroll
          for (int rb = 0; rb < QR; ++rb) s[rb][u] = Mfma<T>::mma(kf, qf[rb][st], s[rb][u]);
        }
      }
      if (p.softcap_scale > 0.f) {                 // scores = softcap * tanh(q.k * scale / softcap), utils.h:383-388
#pragma unroll
        for (int rb = 0; rb < QR; ++rb)
#pragma unroll
          for (int u = 0; u < 2 * KU; ++u)
#pragma unroll
            for (int i = 0; i < 4; ++i) s[rb][u][i] = tanhf(s[rb][u][i] * p.softcap_scale);
      }
      // ---- mask + online softmax (per query column c; state replicated over g).  The scale is
      //      folded into the exponent's fma; tiles that no row of this wave masks skip the compares
      //      (wave-uniform) — the inner loop is VALU-bound (PMC: ~15 vector instructions per MFMA).
      const bool interior = !local && t * KT + KT - 1 <= min(kv_len - 1, p.causal ? q_row0 + shift : kv_len - 1);
      u16x8 pf[QR][KU];
#pragma unroll
      for (int rb = 0; rb < QR; ++rb) {
        float mx = HX_NEG_BIG;
        if (interior) {
#pragma unroll
          for (int u = 0; u < 2 * KU; ++u)
#pragma unroll
            for (int i = 0; i < 4; ++i) mx = fmaxf(mx, s[rb][u][i]);
        } else {
#pragma unroll
          for (int u = 0; u < 2 * KU; ++u)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
              const int key = t * KT + u * 16 + 4 * g + i;
              if (key > limit_c[rb] || key < first_c[rb]) s[rb][u][i] = -INFINITY;
              mx = fmaxf(mx, s[rb][u][i]);
            }
        }
        mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        const float m_new = fmaxf(m[rb], mx * p.scale_log2);
        const float alpha = fast_exp2(m[rb] - m_new);
        m[rb] = m_new;
        float ps = 0.f;
#pragma unroll
        for (int hf = 0; hf < KU; ++hf)
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            const float e = fast_exp2(fmaf(s[rb][2 * hf + (j >> 2)][j & 3], p.scale_log2, -m_new));
            ps += e;
            pf[rb][hf][j] = T::from_float(e);
          }
        l[rb] = l[rb] * alpha + ps;
        if (__builtin_amdgcn_ballot_w64(alpha != 1.0f)) {   // wave-uniform: some row's maximum moved
#pragma unroll
          for (int i = 0; i < NDB; ++i) acc[rb][i] *= alpha;
        }
      }
      // ---- O^T += V^T . P^T (V^T fragments by transposed LDS reads of the shared V image)
#pragma unroll
      for (int hf = 0; hf < KU; ++hf) {
        const char* vrd = vt + (32 * hf + 4 * g + q4) * RS + p4 * 8;
#pragma unroll
        for (int db = 0; db < NDB; ++db) {
          const u16x4 lo = lds_tr_read(vrd + db * 32);
          const u16x4 hi = lds_tr_read(vrd + 16 * RS + db * 32);
          u16x8 vf;
          vf[0] = lo[0]; vf[1] = lo[1]; vf[2] = lo[2]; vf[3] = lo[3];
          vf[4] = hi[0]; vf[5] = hi[1]; vf[6] = hi[2]; vf[7] = hi[3];
#pragma unroll
          for (int rb = 0; rb < QR; ++rb) acc[rb][db] = Mfma<T>::mma(vf, pf[rb][hf], acc[rb][db]);
        }
      }
    }
    if (t + 1 < n_tiles) store_tile(cur ^ 1);
    __syncthreads();    // tile t+1 visible; everyone is done with tile t's image
  }

  // ---- epilogue: O[q 16 rb + c][dim 16db + 4g + i] = acc[rb][db][i] / L
#pragma unroll
  for (int rb = 0; rb < QR; ++rb) {
    float lr = l[rb];
    lr += __shfl_xor(lr, 16, 64);
    lr += __shfl_xor(lr, 32, 64);
    const float inv = (lr > 0.f) ? 1.0f / lr : 0.f;
    const int row = q_row0 + 16 * rb + c;
    if (row < q_len) {
      u16* op = reinterpret_cast<u16*>(p.out) + (int64_t)(q_start + row) * p.o_row_stride +
                (int64_t)h * D + 4 * g;
#pragma unroll
      for (int db = 0; db < NDB; ++db) {
        u16x4 r;
#pragma unroll
        for (int i = 0; i < 4; ++i) r[i] = T::from_float(acc[rb][db][i] * inv);
        *reinterpret_cast<u16x4*>(op + 16 * db) = r;
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------
// 32x32x16 form for D = 64 / 128 and more than 64 query rows (prefill, the CLIP tower): a wave owns
// 32 query rows, a workgroup 128; 64-key tiles.  Per FLOP it spends half the softmax VALU work of
// the 16x16x32 form above (each exp / max / convert now feeds a 32-wide MFMA column instead of a
// 16-wide one) — round 1's profile had that form at 9 % MFMA busy with ~15 vector instructions
// per MFMA (profiles/r1_attn_prefill_pmc.json).
//   * S^T[key][query] = K . Q^T with v_mfma_f32_32x32x16: A = K rows from the LDS image (lane
//     (c = l & 31, h = l >> 5): key c, dims 16 ks + 8 h .. + 8), B = Q^T in registers.  The
//     accumulator leaves lane (c, h) with query c and the 16 keys (r & 3) + 8 (r >> 2) + 4 h: a
//     row statistic is a local reduction plus ONE exchange with lane l ^ 32.
//   * O^T[dim][query] += V^T . P^T: the accumulator registers 8 ks2 .. 8 ks2 + 7, rounded to T, ARE
//     the B operand of k-step ks2 (k-slot 8 h + j <-> key 16 ks2 + 8 (j >> 2) + 4 h + (j & 3)); V^T
//     is the A operand, two ds_read_b64_tr_b16 per MFMA (4-row blocks at keys .. + 4 h and
//     .. + 8 + 4 h, 16 dims per 16-lane group).
//   * LDS images: K rows 2 D + 16 bytes apart (the 32-key ds_read_b128 is bank-conflict free),
//     V rows 2 D + 64 bytes apart (the transposed reads of a 32-lane half are conflict free).
// Same tile pipeline (register prefetch of tile t + 1, double-buffered images, one barrier per
// tile), same masking, same rounding points (P to T before P.V) as the 16x16x32 kernel.
// ---------------------------------------------------------------------------------------------
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <typename T> struct Mfma32;
template <> struct Mfma32<F16> {
  static __device__ __forceinline__ f32x16 mma(u16x8 a, u16x8 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
  }
};
template <> struct Mfma32<BF16> {
  static __device__ __forceinline__ f32x16 mma(u16x8 a, u16x8 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
  }
};

// two workgroups per CU (LDS: 2 x 74 KiB): registers + accumulator registers must stay within 256,
// or one wave per SIMD runs with nothing to hide its latencies behind (measured: 0.8 waves per SIMD
// on average and 80 us for 4 x 704 tokens with the default bound)
// ABL (EXPERIMENTS builds, fwd_ablate option; timing only, wrong results): 1 no softmax arithmetic, 2 no P V product,
// 4 no Q K product, 8 no tile staging and no barrier after the first tile, 16 no barrier, 32 no tile requests
template <typename T, int D, bool PAGED, int ABL = 0>
__global__ __launch_bounds__(256, 2) void attn_fwd32_kernel(const AttnParams p) {
  constexpr int KS = D / 16;         // QK k-steps
  constexpr int NDB = D / 32;        // 32-dim output blocks
  constexpr int KT = 64;             // keys per tile (two 32-key sub-tiles)
  // LDS images: UNPADDED rows of 2 D bytes with the 16-byte chunks XOR-swizzled per row — the tiles arrive by LDS-DMA
  // (global_load_lds_dwordx4: 64 lanes x 16 bytes land contiguously, so a row cannot be padded; which chunk of its row
  // a lane fetches is free).  Chunk c of row r sits at position c ^ kswz(r) in the K and Q images, c ^ vswz(r) in V:
  //   K / Q, ds_read_b128 of chunk 2 ks + h of rows c = 0 .. 31 (16-lane groups {0-3,12-15,20-27}, ... on 64 banks,
  //     MI355X_MICROARCH.md LDS): the 16 rows of a group need 16 different positions (D = 128: row & 15) resp. 8
  //     different ones per row parity (D = 64, two rows per bank row: (row >> 1) & 7);
  //   V, ds_read_b64_tr_b16 of rows q = 0 .. 3 x 64 bytes (32-lane groups): the four rows go to four different
  //     64-byte quarters of the bank row (D = 128: (row & 3) << 2) resp. two different ones per row parity (D = 64).
  constexpr int RSK = 2 * D, RSV = 2 * D;
  constexpr int LPR = D / 8;
  constexpr int NL = KT * LPR / 256;
  constexpr int KTILE = KT * RSK, VTILE = KT * RSV;
  constexpr int IMG = KTILE + VTILE;       // one tile: K image, V image; two of them
  constexpr int TQ = 128;
  extern __shared__ __attribute__((aligned(16))) char smem[];   // [2][K[KT][RSK] | V[KT][RSV]] | priority flag
  auto kswz = [](int row) { return LPR == 16 ? (row & 15) : ((row >> 1) & 7); };
  auto vswz = [](int row) { return LPR == 16 ? ((row & 3) << 2) : (((row >> 1) & 1) << 2); };
  if (ABL == 256) return;                                        // the launch alone

  // Workgroup -> (sequence, query tile, head).  Compact grid (x = tile slots, y = heads) as in the
  // kernel above, with two differences that matter for the causal tail: slots are ordered by tile
  // LENGTH across all sequences (the r-th longest tile of every sequence before any (r+1)-th), and
  // inside an XCD's contiguous id range the head is the fastest index — so an XCD starts with the
  // longest tiles of ALL its heads and sequences, not with every tile of its first head
  // (4 x 704 tokens: 49 -> see tools/bench_attn_prefill32.py).
  // the per-sequence index arrays through the scalar cache (uniform indices; as vector loads each was a ~1 us round
  // trip in a chain of five before the first K / V request could be formed)
  typedef __attribute__((address_space(4))) const int32_t c_i32;
  c_i32* cu_q_s = (c_i32*)p.cu_q;
  c_i32* cu_k_s = (c_i32*)p.cu_k;
  int mblk = -1, h, b = 0;
  int q_start = 0, q_len = 0, k_start = 0, kv_len = 0, bt_off = 0;      // of sequence b: read with its group's offsets
  {
    const int gx = gridDim.x, gy = gridDim.y;
    const int total = gx * gy;
    int wg = blockIdx.x + gx * blockIdx.y;
    int slot;
    if (p.xcd_remap && total % 8 == 0 && gy % 8 == 0) {
      const int hp = gy / 8;                               // XCD x runs ids x, x + 8, ...: hp heads each
      const int x = wg % 8, i = wg / 8;
      h = x * hp + i % hp;
      slot = i / hp;
    } else {
      slot = wg % gx;
      h = wg / gx;
    }
    // the slot-th (tile rank, sequence) pair: sequences in groups of 4; inside a group rank 0 (each
    // sequence's longest tile) of its sequences, then rank 1, ...  (Rank-major over ALL sequences
    // spreads the tiles of one (sequence, head) so far apart in time that they stop sharing K / V
    // in L2: 32 x 704 tokens ran 1.5x slower that way.)
    // (the group's five offsets are requested together and kept: with a load inside the rank loop the decode was a
    // chain of up to 24 scalar round trips, ~2 us in front of every workgroup's first request for K / V)
    for (int g0 = 0; g0 < p.batch && mblk < 0; g0 += 4) {
      // (cu_k and cu_block_lens of the group's sequences ride along: fetched after the sequence was known they were one
      // more dependent round trip in front of the page lookup and the first K / V request)
      int cq[5], ck[5], cb[4];
#pragma unroll
      for (int i = 0; i < 5; ++i) cq[i] = cu_q_s[min(g0 + i, p.batch)];
#pragma unroll
      for (int i = 0; i < 5; ++i) ck[i] = cu_k_s[min(g0 + i, p.batch)];
#pragma unroll
      for (int i = 0; i < 4; ++i) cb[i] = PAGED ? ((c_i32*)p.cu_block_lens)[min(g0 + i, p.batch - 1)] : 0;
      int tl[4], max_tiles = 0, group_tiles = 0;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        tl[i] = (cq[i + 1] - cq[i] + TQ - 1) / TQ;          // 0 past the last sequence (its offsets repeat)
        max_tiles = max(max_tiles, tl[i]);
        group_tiles += tl[i];
      }
      if (slot >= group_tiles) { slot -= group_tiles; continue; }
      for (int rank = 0; rank < max_tiles && mblk < 0; ++rank) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          if (mblk < 0 && tl[i] > rank) {
            if (slot == 0) {
              b = g0 + i; mblk = tl[i] - 1 - rank;
              q_start = cq[i]; q_len = cq[i + 1] - cq[i]; k_start = ck[i]; kv_len = ck[i + 1] - ck[i]; bt_off = cb[i];
            }
            --slot;
          }
        }
      }
    }
    if (mblk < 0) return;                // spare slot (the grid is an upper bound)
  }
  const int lane = threadIdx.x & 63;
  const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int c = lane & 31, hi = lane >> 5;
  const int hk = h / p.group;
  b = __builtin_amdgcn_readfirstlane(b);
// (priority of this workgroup's waves: see below)
  // Query tiles are aligned to the END of the sequence: the partial tile (q_len % 128 rows) is the FIRST one — under the
  // causal mask the tile with the fewest keys — and every other tile is full.  Aligned to the start, the partial tile was
  // the one with the most keys: 704 rows = 5 x 128 + 64 cost 2 + 4 + 6 + 8 + 10 + 11 = 41 tile steps, the last eleven of
  // them with half of the waves idle; this way 1 + 3 + 5 + 7 + 9 + 11 = 36.  Rows below 0 do not exist: their loads are
  // clamped, their results are not stored.
  const int q_row0_wg = q_len - ((q_len + TQ - 1) / TQ - mblk) * TQ;
  if (q_len <= 0) return;
  const int q_row0 = q_row0_wg + w * 32;

  const u16* kbase = reinterpret_cast<const u16*>(p.k) + (int64_t)hk * p.k_head_stride;
  const u16* vbase = reinterpret_cast<const u16*>(p.v) + (int64_t)hk * p.v_head_stride;
  const int32_t* bt = PAGED ? p.block_table + bt_off : nullptr;

  // Staging map of a wave (Q here, K / V tiles below, O at the end): instruction j takes rows RPI j .. RPI j + RPI - 1
  // of the wave's block, D / 8 lanes per row — the whole row contiguous.
  constexpr int RPI = 64 / LPR;         // rows per instruction
  const int st_r4 = lane / LPR;         // row RPI j + st_r4
  const int st_ch = lane % LPR;         // 16-byte chunk of the row
  const int shift = kv_len - q_len;
  const int limit_c = p.causal ? min(kv_len - 1, q_row0 + c + shift) : kv_len - 1;
  const int last_key_wave = q_row0 + 31 < 0 ? -1 : p.causal ? min(kv_len - 1, q_row0 + 31 + shift) : kv_len - 1;
  const int last_key_wg = p.causal ? min(kv_len - 1, min(q_row0_wg + TQ - 1, q_len - 1) + shift) : kv_len - 1;
  const int n_tiles = (last_key_wg >= 0) ? last_key_wg / KT + 1 : 0;

  // ---- tile staging by LDS-DMA.  Wave w stages the tile's keys 16 w .. 16 w + 15 (a 16-key group never straddles a
  // page: block_size % 16 == 0), instruction j its rows RPI j .. RPI j + RPI - 1 — D / 8 lanes per row, 1 KiB of the
  // image per instruction and cache, every cache line touched by one instruction.  Through registers (global load,
  // ds_write at the end of the tile) the same requests cost 8 of the 37 us of the 4 x 704 launch
  // (tools/ablate_attn_prefill32.py: "no tile loads"), this way 3.  The loads are inline assembly: told about an LDS-DMA,
  // hipcc waits for it (vmcnt(0)) in front of every LDS read that follows — the reads of the OTHER image.
  static_assert(NL * RPI == 16 && RPI * RSK == 1024, "a wave stages one 16-key group, 1 KiB per instruction");
  // Page of my keys in the tile that will be requested next.  A wave stages ONE 16-key group, hence one page: the
  // table entry is read through the SCALAR cache, in front of the barrier (a scalar load shares its counter with the
  // LDS reads and returns out of order — while one is outstanding every LDS read is waited for singly).
  c_i32* bt_s = (c_i32*)bt;
  int page_next = 0;
  auto lookup_page = [&](int t) {
    if (PAGED) page_next = bt_s[__builtin_amdgcn_readfirstlane(page_slot(min(t * KT + 16 * w, kv_len - 1), p.block_size, p.block_shift))];
  };
  // Addresses: a wave-uniform part per tile (page and first row of the wave's 16-key group: scalar arithmetic) plus a
  // per-thread part that changes only where the group runs past the last key (rows are clamped to it).
  const int last_key = kv_len - 1;
  const uint32_t lds0 = (uint32_t)(size_t)(__attribute__((address_space(3))) char*)(smem);
  auto request_tile = [&](int t, int img) {
    const int g0 = min(t * KT + 16 * w, last_key);      // first key of the wave's group, clamped: uniform
    const int r_max = last_key - g0;                     // rows of the group that exist (>= 0)
    int64_t kb, vb;
    if (PAGED) {
      const int rowg = page_row(g0, p.block_size, p.block_shift);
      kb = (int64_t)page_next * p.k_block_stride + (int64_t)rowg * p.k_row_stride;
      vb = (int64_t)page_next * p.v_block_stride + (int64_t)rowg * p.v_row_stride;
    } else {
      kb = (int64_t)(k_start + g0) * p.k_row_stride;
      vb = (int64_t)(k_start + g0) * p.v_row_stride;
    }
    const uint32_t kd = __builtin_amdgcn_readfirstlane(lds0 + (uint32_t)(img * IMG + 16 * w * RSK));
    const uint32_t vd = kd + KTILE;
#pragma unroll
    for (int j = 0; j < NL; ++j) {
      const int R = RPI * j + st_r4;                               // row of the group = row of the tile mod 16
      const uint32_t r = (uint32_t)min(R, r_max);                  // rows past the last key repeat it
      const u16* ka = kbase + kb + r * (uint32_t)p.k_row_stride + 8 * (st_ch ^ kswz(R));
      const u16* va = vbase + vb + r * (uint32_t)p.v_row_stride + 8 * (st_ch ^ vswz(R));
      // (m0 — the LDS address of the DMA — is a reserved register: naming it as clobbered is all that can be done, and
      // nothing else in this kernel uses it)
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Winline-asm"
      asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" :: "v"(ka), "s"(kd + 1024u * j) : "memory", "m0");
      asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" :: "v"(va), "s"(vd + 1024u * j) : "memory", "m0");
#pragma clang diagnostic pop
    }
  };
  auto tiles_landed = [&]() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); };
  using Set0 = std::integral_constant<int, 0>;
  using Set1 = std::integral_constant<int, 1>;

  f32x16 acc[NDB];
#pragma unroll
  for (int i = 0; i < NDB; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  float m = HX_NEG_BIG, l = 0.f;

  const int t_last = max(n_tiles - 1, 0);
  if (n_tiles > 0) {
    lookup_page(0);
    request_tile(0, 0);
    lookup_page(min(1, t_last));
  }
  // (Q behind the first K / V request: the two round trips overlap instead of following each other)
  // Q: whole rows into a wave-private LDS block (the second tile image is idle until tile 1 is requested, after the
  // barrier below; 4 waves x 32 rows fill it exactly), the B-operand fragments out of it, same swizzle as K.  Loaded
  // straight into the fragment layout every instruction took 32 bytes of each of 32 rows.
  u16x8 qf[KS];
  {
    constexpr int RSO = RSK;
    char* qb = smem + IMG + w * 32 * RSO;
    const u16* qbase = reinterpret_cast<const u16*>(p.q) + (int64_t)h * D + 8 * st_ch;
    u16x8 qrow[32 / RPI];
#pragma unroll
    for (int j = 0; j < 32 / RPI; ++j) {
      const int qr = max(q_row0 + RPI * j + st_r4, 0);
      if (ABL == 512) qrow[j] = u16x8{(u16)(0x3c00 + lane), 0x3800, 0x3400, 0x3000, 0x2c00, 0x2800, (u16)(0x2400 + j), 0x2000};
      else qrow[j] = *reinterpret_cast<const u16x8*>(qbase + (int64_t)(q_start + qr) * p.q_row_stride);
    }
#pragma unroll
    for (int j = 0; j < 32 / RPI; ++j)
      *reinterpret_cast<u16x8*>(qb + (RPI * j + st_r4) * RSO + 16 * (st_ch ^ kswz(RPI * j + st_r4))) = qrow[j];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) qf[ks] = *reinterpret_cast<const u16x8*>(qb + c * RSO + 16 * ((2 * ks + hi) ^ kswz(c)));
  }
  if (n_tiles > 0) tiles_landed();
  // The two workgroups of a CU at DIFFERENT priorities.  At equal priority two waves of a SIMD that happen to be in
  // the same phase slow each other equally and stay in phase — MFMA burst against MFMA burst, softmax against softmax
  // — and a tile costs its MFMA plus its VALU time (4 x 704: 1.8 us per tile and pair of workgroups with the staging
  // removed = 2 x (1024 + ~1100) cycles; PMC: a VALU instruction co-executes in 28 % of the MFMA cycles).  With one
  // workgroup always issuing first, the other fills what it leaves: its MFMAs under the first one's softmax and the
  // other way round.  Which of the two: the parity of the hardware wave slot of wave 0 (two resident workgroups whose
  // first waves share a SIMD sit in different slots there; otherwise both may get the same priority — no loss).
  // Only when workgroups queue for the CUs (more than two per CU in the launch): of two workgroups that start together
  // and have nobody waiting for their slot, the favoured one finishes early and the other runs its last tiles alone
  // (one round of 512 equal workgroups, 2048 new tokens of 4096: 128 us without, 135 us with priorities).
  uint32_t* prio_flag = reinterpret_cast<uint32_t*>(smem + 2 * IMG);
  const bool queued = p.wg_priority != 0;
  if (queued && threadIdx.x == 0) *prio_flag = __builtin_amdgcn_s_getreg(4 | (0 << 6) | (3 << 11)) & 1;     // HW_ID.wave_id
  __syncthreads();
  if (queued && *prio_flag) __builtin_amdgcn_s_setprio(3);

  // transposed-read lane address inside a 4-row x 32-dim block: lane 4q + pp of each 16-lane group
  // supplies row q, dims 16 half + 4 pp .. + 3 (half = which 16 of the 32 dims this group takes)
  const int tr_q = (lane & 15) >> 2, tr_pp = lane & 3, tr_half = (lane >> 4) & 1;
  const int tr_off = (4 * hi + tr_q) * RSV + (16 * tr_half + 4 * tr_pp) * 2;      // + 64 * (db ^ tr_x): the swizzled quarter
  const int tr_x = vswz(tr_q) >> 2;

  auto tile_step = [&](int t, auto par_tag) {
    constexpr int PAR = decltype(par_tag)::value;       // t & 1: this tile's LDS image
    const int cur = (ABL & 8) ? 0 : PAR;
    if (!(ABL & (8 | 32)) && t + 1 < n_tiles) request_tile(t + 1, 1 - PAR);   // lands under this tile's arithmetic
    if (t * KT <= last_key_wave) {
      const char* kt = smem + cur * IMG;
      const char* vt = kt + KTILE;
      f32x16 s[2];
#pragma unroll
      for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int r = 0; r < 16; ++r) s[u][r] = 0.f;
      {
        // all K fragments of sub-tile 0 are requested before its first MFMA, those of sub-tile 1
        // under sub-tile 0's MFMAs: no MFMA waits for a read issued just before it
        const char* krd = kt + c * RSK;
        const int kz = kswz(c);
        u16x8 kfa[KS], kfb[KS];
        if (ABL & 4) {
#pragma unroll
          for (int r = 0; r < 16; ++r) { s[0][r] = 0.01f * (r + lane); s[1][r] = 0.02f * (r + t); }
        } else {
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) kfa[ks] = *reinterpret_cast<const u16x8*>(krd + 16 * ((2 * ks + hi) ^ kz));
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
          kfb[ks] = *reinterpret_cast<const u16x8*>(krd + 32 * RSK + 16 * ((2 * ks + hi) ^ kz));
          s[0] = Mfma32<T>::mma(kfa[ks], qf[ks], s[0]);
        }
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) s[1] = Mfma32<T>::mma(kfb[ks], qf[ks], s[1]);
        // keep that order: left alone the scheduler issues read, wait, MFMA, read, wait, MFMA (fewest registers)
        __builtin_amdgcn_sched_group_barrier(0x100, KS, 0);
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        }
        __builtin_amdgcn_sched_group_barrier(0x008, KS, 0);
        }
      }
      const bool interior = t * KT + KT - 1 <= min(kv_len - 1, p.causal ? q_row0 + shift : kv_len - 1);
      float mx = HX_NEG_BIG;
      u16x8 pf[2][2];
      if (ABL & 1) {
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
          for (int r = 0; r < 16; ++r) pf[u][r >> 3][r & 7] = T::from_float(s[u][r]);
        l += 1.f;
      } else {
      if (interior) {
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
          for (int r = 0; r < 16; ++r) mx = fmaxf(mx, s[u][r]);
      } else {
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int key = t * KT + 32 * u + (r & 3) + 8 * (r >> 2) + 4 * hi;
            if (key > limit_c) s[u][r] = -INFINITY;
            mx = fmaxf(mx, s[u][r]);
          }
      }
      mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
      // Lazy running maximum: a row keeps its reference m until a score exceeds it by more than 2^8 (the exponentials
      // then stay below 256: exact in fp32, eight mantissa bits as ever in T) — with the exact maximum as reference some
      // row of the 32 moved in nearly every tile and all 64 accumulator registers were rescaled every time (PMC: 33
      // v_pk_mul per wave and tile).  O = acc / l is unchanged in exact arithmetic: both carry the same factor.
      const float m_cand = fmaxf(m, mx * p.scale_log2);
      const bool grow = m_cand > m + 8.0f;
      float m_new = m;
      if (__builtin_amdgcn_ballot_w64(grow)) {
        m_new = grow ? m_cand : m;
        const float alpha = fast_exp2(m - m_new);      // 1 for the rows that keep their reference
        l *= alpha;
#pragma unroll
        for (int i = 0; i < NDB; ++i)
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[i][r] *= alpha;
        m = m_new;
      }
      // (single-instruction fma / add: a packed f32 instruction beside MFMAs costs more than the two it replaces —
      // MI355X_MICROARCH.md, per-instruction constants; the file is compiled without the SLP vectorizer for the same reason)
      float pa = 0.f, pb = 0.f;
#pragma unroll
      for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int r = 0; r < 16; r += 2) {
          const float e0 = fast_exp2(fmaf(s[u][r], p.scale_log2, -m_new));
          const float e1 = fast_exp2(fmaf(s[u][r + 1], p.scale_log2, -m_new));
          pa += e0;
          pb += e1;
          pf[u][r >> 3][r & 7] = T::from_float(e0);
          pf[u][r >> 3][(r & 7) + 1] = T::from_float(e1);
        }
      l += pa + pb;
      }
      if (ABL & 2) {
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
          for (int k2 = 0; k2 < 2; ++k2) acc[0][u * 2 + k2] += __builtin_bit_cast(float, (uint32_t)pf[u][k2][0] << 16);
      } else
#pragma unroll
      for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int k2 = 0; k2 < 2; ++k2) {
          const char* vrd = vt + (32 * u + 16 * k2) * RSV + tr_off;
#pragma unroll
          for (int db = 0; db < NDB; ++db) {
            const u16x4 lo = lds_tr_read(vrd + 64 * (db ^ tr_x));
            const u16x4 hh = lds_tr_read(vrd + 8 * RSV + 64 * (db ^ tr_x));
            u16x8 vf;
            vf[0] = lo[0]; vf[1] = lo[1]; vf[2] = lo[2]; vf[3] = lo[3];
            vf[4] = hh[0]; vf[5] = hh[1]; vf[6] = hh[2]; vf[7] = hh[3];
            acc[db] = Mfma32<T>::mma(vf, pf[u][k2], acc[db]);
          }
        }
    }
    if (!(ABL & 8)) {
      if (!(ABL & 32)) lookup_page(min(t + 2, t_last));       // the table entry for the next request, in front of the barrier's wait
      tiles_landed();                                        // this wave's part of tile t+1 is in LDS
      if (!(ABL & 16)) __syncthreads();    // tile t+1 visible; everyone is done with tile t's image
    }
  };
  for (int t = 0; t < n_tiles; t += 2) {
    tile_step(t, Set0{});
    if (t + 1 < n_tiles) tile_step(t + 1, Set1{});
  }

  if (ABL == 79 && acc[0][0] != 123.f) return;   // (1039 = 1024 + 15: empty loop, epilogue without its LDS reads)                   // everything but the epilogue
  // epilogue: O[query c][dim 32 db + 8 (r >> 2) + 4 hi + (r & 3)] = acc[db][r] / L
  float lr = l + __shfl_xor(l, 32, 64);
  const float inv = (lr > 0.f) ? 1.0f / lr : 0.f;
  // Through LDS (the K / V images are free: every wave has passed the last tile's barrier), so that a store
  // instruction writes whole rows: straight from the accumulator layout each instruction put 16 bytes into each of 32
  // rows — sixteen such instructions per wave, ~10 us of the 4 x 704 launch by themselves
  // (tools/ablate_attn_prefill32.py, "empty loop" 16.3 us against 6.1 without the stores).
  if (q_row0 + 31 >= 0) {
    // Unpadded rows with an XOR swizzle of the 8-byte slots (MI355X_MICROARCH.md, LDS): a ds_write_b64 is served in four
    // groups of 16 contiguous lanes on 32 banks — 16 rows at the same column need 16 different slot positions mod 16
    // (slot ^ row does it; a padded stride of 4 banks met pairwise, PMC 6 % of the LDS cycles) — and the ds_read_b128
    // of whole rows in its four non-contiguous groups is conflict-free exactly when the rows are 256 bytes apart.
    constexpr int RSO = 2 * D;
    char* ob = smem + w * 32 * RSO;      // this wave's 32 rows
#pragma unroll
    for (int db = 0; db < NDB; ++db)
#pragma unroll
      for (int rq = 0; rq < 4; ++rq) {
        u16x4 o;
#pragma unroll
        for (int i = 0; i < 4; ++i) o[i] = T::from_float(acc[db][4 * rq + i] * inv);
        *reinterpret_cast<u16x4*>(ob + c * RSO + (((8 * db + 2 * rq + hi) ^ (c & (LPR - 1))) << 3)) = o;
      }
    u16* obase = reinterpret_cast<u16*>(p.out) + (int64_t)h * D + 8 * st_ch;
#pragma unroll
    for (int j = 0; j < 32 / RPI; ++j) {
      const int rl = RPI * j + st_r4;
      // slots 2 ch, 2 ch + 1 of row rl sit in chunk ch ^ (rl' >> 1), swapped when rl' is odd (rl' = rl mod D / 8)
      u16x8 v = *reinterpret_cast<const u16x8*>(ob + rl * RSO + 16 * (st_ch ^ ((rl & (LPR - 1)) >> 1)));
      if (rl & 1) v = u16x8{v[4], v[5], v[6], v[7], v[0], v[1], v[2], v[3]};
      if (q_row0 + rl >= 0) *reinterpret_cast<u16x8*>(obase + (int64_t)(q_start + q_row0 + rl) * p.o_row_stride) = v;
    }
  }
}


// ---------------------------------------------------------------------------------------------
// The same arithmetic in PERSISTENT workgroups: two per CU, each walking a list of (sequence, query tile, head) items.
// Why: a 128-row query tile of a 704-token prompt has 1 .. 11 key tiles (6 on average), and with one workgroup per item
// the time around the tile loop — index decode, Q rows and the first K / V tile from HBM with nothing to compute
// meanwhile, the O rows through LDS and out — was ~9 us per workgroup against ~11 us in its loop (32 x 704 tokens:
// 266 us where 82 tile steps per workgroup slot at the 1.87 us of the long launches are 153).  Here the next item's Q
// rows arrive by LDS-DMA in the image that the last tile of the current item leaves free, its first K / V tile is in
// flight while the current item's O rows go out, and the item after that is decoded meanwhile.
// Items are dealt out statically, in snake order over the length-sorted item list (round r: item r G + i to workgroup
// i, the next round backwards), inside the XCD whose L2 holds that head's K / V.
// ---------------------------------------------------------------------------------------------
// A uniform value the compiler may not reason about: what is derived from it (strides times constants, the reciprocal
// of a divisor) is computed where it is used instead of once at kernel entry and then kept, or spilled, across the
// persistent loop.
template <typename V> __device__ __forceinline__ V sfresh(V x) { asm volatile("" : "+s"(x)); return x; }

// STAMPS (EXPERIMENTS builds, hx_debug_fwd_stamps): wave 0 of every workgroup appends (100 MHz time << 8 | event) words
// to its 512-word list in p.stamps (tools/fwd_timeline.py)
template <typename T, int D, bool PAGED, bool STAMPS = false>
__global__ __launch_bounds__(256, 2) void attn_fwd32p_kernel(const AttnParams p) {
  constexpr int KS = D / 16;         // QK k-steps
  constexpr int NDB = D / 32;        // 32-dim output blocks
  constexpr int KT = 64;             // keys per tile (two 32-key sub-tiles)
  constexpr int RSK = 2 * D, RSV = 2 * D;
  constexpr int LPR = D / 8;
  constexpr int NL = KT * LPR / 256;
  constexpr int KTILE = KT * RSK, VTILE = KT * RSV;
  constexpr int IMG = KTILE + VTILE;
  constexpr int TQ = 128;
  constexpr int RPI = 64 / LPR;         // rows per staging instruction
  constexpr int NQI = 32 / RPI;         // staging instructions per 32-row block of Q / O
  static_assert(NL * RPI == 16 && RPI * RSK == 1024, "a wave stages one 16-key group, 1 KiB per instruction");
  extern __shared__ __attribute__((aligned(16))) char smem[];   // [2][K[KT][RSK] | V[KT][RSV]] | priority flag
  auto kswz = [](int row) { return LPR == 16 ? (row & 15) : ((row >> 1) & 7); };
  auto vswz = [](int row) { return LPR == 16 ? ((row & 3) << 2) : (((row >> 1) & 1) << 2); };
  typedef __attribute__((address_space(4))) const int32_t c_i32;

  const int lane = threadIdx.x & 63;
  const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int c = lane & 31, hi = lane >> 5;
  const int st_r4 = lane / LPR;         // staging map: row RPI j + st_r4 of the wave's block, 16-byte chunk st_ch
  const int st_ch = lane % LPR;
  const uint32_t lds0 = (uint32_t)(size_t)(__attribute__((address_space(3))) char*)(smem);
  // The arguments as the code BETWEEN items reads them: from the kernel-argument segment, through a pointer the compiler
  // cannot see through (sfresh).  Read as `p.field`, everything the item decode, the Q request and the epilogue derive
  // from the arguments is loop-invariant, is hoisted in front of the persistent loop and then lives in — or is spilled
  // from — the scalar registers the tile loop needs (346 v_readlane / v_writelane in the first build).
  typedef __attribute__((address_space(4))) const AttnParams c_params;
  auto kargs = [&]() __attribute__((always_inline)) { return sfresh((c_params*)__builtin_amdgcn_kernarg_segment_ptr()); };

  int n_stamps = 0;
  auto stamp = [&](int id) __attribute__((always_inline)) {
    if constexpr (STAMPS) {
      if (w == 0 && n_stamps < 512) {
        const uint64_t v = (__builtin_amdgcn_s_memrealtime() << 8) | (uint64_t)id;
        if (lane == 0) p.stamps[(size_t)blockIdx.x * 512 + n_stamps] = v;
        ++n_stamps;
      }
    }
  };
  stamp(1);
  auto stamp_cycles = [&](int id) __attribute__((always_inline)) {      // the shader clock's counter beside the 100 MHz one
    if constexpr (STAMPS) {
      if (w == 0 && n_stamps < 512) {
        const uint64_t v = (__builtin_readcyclecounter() << 8) | (uint64_t)id;
        if (lane == 0) p.stamps[(size_t)blockIdx.x * 512 + n_stamps] = v;
        ++n_stamps;
      }
    }
  };
  stamp_cycles(30);
  if constexpr (STAMPS) {      // where the workgroup runs: HW_ID (CU_ID [11:8], SH_ID [12], SE_ID [15:13]) and XCC_ID
    if (w == 0 && n_stamps < 512) {
      const uint64_t v = ((uint64_t)__builtin_amdgcn_s_getreg(20 | (0 << 6) | (31 << 11)) << 40) |
                         ((uint64_t)__builtin_amdgcn_s_getreg(4 | (0 << 6) | (31 << 11)) << 8) | 32u;
      if (lane == 0) p.stamps[(size_t)blockIdx.x * 512 + n_stamps] = v;
      ++n_stamps;
    }
  }

  // ---- the items of this workgroup: a table in LDS, built once.  Thread r decodes the item of round r — the r-th entry
  // of the snake walk over the item list —, all rounds at the same time.  (Decoded one at a time by every wave, through
  // the scalar cache, an item cost 3 + batch / 4 dependent round trips: 4 - 5 us in front of the first request at
  // 4 sequences, 7.6 us at 32, and 2 - 3 us at every seam — in-kernel stamps, tools/fwd_timeline.py.  And everything
  // uniform that the tile loop does not need must stay out of the scalar registers: see kargs above.)
  // record: n_tiles (0 = no item in this round, -1 = an item without keys: its rows are zero), h, q_row0_wg, q_start,
  //         q_len, k_start, kv_len, bt_off, first page of the 16-key group of wave 0 .. 3 in the item's first tile
  struct Work { int h, q_row0_wg, q_start, q_len, k_start, kv_len, bt_off, n_tiles; };
  constexpr int REC = 12;
  // Sequences are dealt in groups of `gs`: inside a group rank 0 (each sequence's longest tile) of its sequences, then
  // rank 1, ...  gs = 4 for up to 4 sequences (longest tiles first over the whole launch), gs = 1 beyond: the tiles of one
  // (sequence, head) then sit next to each other in the item order, run in the same round on the same XCD, and their K / V
  // is fetched once — with groups of 4, 32 x 704 tokens read 1014 MB from beyond L2 for 554 MB of unique Q / K / V
  // (FETCH_SIZE, tools/pmc_prefill_traffic.sh: an L2 hit rate of 27 % on 1.4 GB requested, 5.3 TB/s).
  const int gs = p.seq_group;
  const int n_groups = (p.batch + gs - 1) / gs;
  int* tb_g = reinterpret_cast<int*>(smem + 2 * IMG + 16);             // slots before group g, g = 0 .. n_groups
  int* items = tb_g + ((n_groups + 1 + 3) & ~3);                        // [n_rounds][REC], 16-byte aligned
  // Unit mode (many sequences: p.unit_mode): what is dealt is a UNIT — the k-th longest and the k-th shortest query tile
  // of one (sequence, head), run back to back by one workgroup (two records per round).  Under the causal mask every
  // unit of a sequence costs about the same (704 tokens: 11 + 1, 9 + 3, 7 + 5 tile steps), so the workgroups that share a
  // sequence's K / V start their units together in every round, not only in the first, and fetch its tiles once.
  const bool units = p.unit_mode != 0;
  const int n_rounds = (units ? 2 : 1) * ((p.n_tile_slots * p.n_heads + (int)gridDim.x - 1) / (int)gridDim.x);
  {
    // the per-sequence arrays first, into the (still unused) tile images: one round trip to global memory instead of one
    // per step of the decode (group counts -> the group's offsets -> the sequence's offsets -> its pages: 3.3 us in front
    // of the first request at 4 x 704 tokens)
    int* sq = reinterpret_cast<int*>(smem);          // cu_q[0 .. batch]
    int* sk = sq + p.batch + 1;                      // cu_k[0 .. batch]
    int* sb = sk + p.batch + 1;                      // cu_block_lens[0 .. batch - 1]
    for (int i = threadIdx.x; i <= p.batch; i += 256) {
      sq[i] = p.cu_q[i];
      sk[i] = p.cu_k[i];
      if (PAGED && i < p.batch) sb[i] = p.cu_block_lens[i];
    }
    __syncthreads();
    for (int g = threadIdx.x; g < n_groups; g += 256) {
      int n = 0;
#pragma unroll
      for (int i = 0; i < 4; ++i)
        if (i < gs) n += (sq[min(gs * g + i + 1, p.batch)] - sq[min(gs * g + i, p.batch)] + TQ - 1) / TQ;
      tb_g[g] = units ? (n + 1) >> 1 : n;            // (unit mode: gs = 1, a sequence of n tiles has (n + 1) / 2 units)
    }
    __syncthreads();
    if (w == 0) {
      int carry = 0;
      for (int g0 = 0; g0 < n_groups; g0 += 64) {
        const int g = g0 + lane;
        const int mine = g < n_groups ? tb_g[g] : 0;
        int incl = mine;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
          const int o = __shfl_up(incl, d, 64);
          if (lane >= d) incl += o;
        }
        if (g < n_groups) tb_g[g] = carry + incl - mine;
        carry += __shfl(incl, 63, 64);
      }
      if (lane == 0) tb_g[n_groups] = carry;
    }
    __syncthreads();
    const int n_slots = p.n_tile_slots, gy = p.n_heads;
    const int total = n_slots * gy;
    const int G = gridDim.x;
    const bool remap = p.xcd_remap && total % 8 == 0 && gy % 8 == 0 && G % 8 == 0;
    // (threads n_rounds .. 2 n_rounds - 1 decode the items of the workgroup this one shares its CU with — only their
    // lengths are kept, for the priority below)
    const int partner = (int)blockIdx.x < p.n_cus ? (int)blockIdx.x + p.n_cus : (int)blockIdx.x - p.n_cus;
    int* p_tiles = sb + p.batch;                     // [n_rounds]: tile steps of the partner's items
    for (int idx = threadIdx.x; idx < 2 * n_rounds; idx += 256) {
      const bool mine = units || idx < n_rounds;
      if (units && idx >= n_rounds) { p_tiles[idx - n_rounds] = 0; continue; }
      const int r = units ? idx >> 1 : mine ? idx : idx - n_rounds;          // round (unit mode: records 2 r, 2 r + 1)
      const int which = units ? idx & 1 : 0;
      const int wg = mine ? (int)blockIdx.x : partner;
      int* rec = mine ? items + idx * REC : p_tiles + r;
      rec[0] = 0;
      if (wg >= G) continue;
      int slot, h;
      // Which of the round's items, longest first: the dispatcher gives every CU one workgroup before it gives any CU a
      // second one (in-kernel HW_ID stamps, tools/fwd_timeline.py: workgroups wg and wg + n_cus share a CU, always), so
      // the second workgroup of a CU takes the items from the short end — CU c gets the c-th longest and the c-th
      // shortest of a round.  (In id order the two longest tiles of 2048 new tokens of 4096 met on one CU: 112 tile steps
      // against 84 on the last CU, where every CU now has 98.)
      if (remap) {
        const int Gx = G / 8, Tx = total / 8, hp = gy / 8;      // XCD x runs workgroups x, x + 8, ...: hp heads each
        const int x = wg % 8, j = wg / 8, H = p.n_cus / 8;
        const int pj = units || !p.cu_pairing || j < H ? j : Gx - 1 - (j - H);
        const int i = (r & 1) && !units ? (r + 1) * Gx - 1 - pj : r * Gx + pj;
        if (i >= Tx) continue;
        h = x * hp + i % hp;
        slot = i / hp;
      } else {
        const int pw = units || !p.cu_pairing || wg < p.n_cus ? wg : G - 1 - (wg - p.n_cus);
        const int i = (r & 1) && !units ? (r + 1) * G - 1 - pw : r * G + pw;
        if (i >= total) continue;
        slot = i % n_slots;
        h = i / n_slots;
      }
      // the slot-th (tile rank, sequence) pair: sequences in groups of 4; inside a group rank 0 (each sequence's
      // longest tile) of its sequences, then rank 1, ...  (Rank-major over ALL sequences spreads the tiles of one
      // (sequence, head) so far apart in time that they stop sharing K / V in L2.)
      if (slot >= tb_g[n_groups]) continue;                    // spare slot (the slot count is an upper bound)
      int lo = 0, hi_g = n_groups;                             // last group whose first slot is <= slot
      while (hi_g - lo > 1) {
        const int mid = (lo + hi_g) >> 1;
        if (tb_g[mid] <= slot) lo = mid; else hi_g = mid;
      }
      slot -= tb_g[lo];
      int tl[4], max_tiles = 0;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        tl[i] = i < gs ? (sq[min(gs * lo + i + 1, p.batch)] - sq[min(gs * lo + i, p.batch)] + TQ - 1) / TQ : 0;
        max_tiles = max(max_tiles, tl[i]);
      }
      int mblk = -1, b = 0;
      if (units) {
        // unit `slot` of sequence lo: its (slot)-th longest tile first, then its (slot)-th shortest (if another one)
        b = lo;
        const bool short_one = (which != 0) != (p.unit_mode == 2);      // unit_mode 2: the short tile first
        const int k = short_one ? slot : tl[0] - 1 - slot;
        mblk = short_one && 2 * slot == tl[0] - 1 ? -1 : k;
      } else {
        for (int rank = 0; rank < max_tiles && mblk < 0; ++rank) {
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            if (mblk < 0 && tl[i] > rank) {
              if (slot == 0) { b = gs * lo + i; mblk = tl[i] - 1 - rank; }
              --slot;
            }
          }
        }
      }
      if (mblk < 0) continue;
      const int q_start = sq[b], q_len = sq[b + 1] - q_start;
      const int k_start = sk[b], kv_len = sk[b + 1] - k_start;
      if (q_len <= 0) continue;
      // query tiles aligned to the END of the sequence (see attn_fwd32_kernel)
      const int q_row0_wg = q_len - ((q_len + TQ - 1) / TQ - mblk) * TQ;
      const int last_key_wg = p.causal ? min(kv_len - 1, min(q_row0_wg + TQ - 1, q_len - 1) + kv_len - q_len) : kv_len - 1;
      if (!mine) { rec[0] = max(last_key_wg / KT + 1, 0); continue; }
      const int bt_off = PAGED ? sb[b] : 0;
      rec[1] = h; rec[2] = q_row0_wg; rec[3] = q_start; rec[4] = q_len; rec[5] = k_start; rec[6] = kv_len; rec[7] = bt_off;
#pragma unroll
      for (int i = 0; i < 4; ++i)
        rec[8 + i] = PAGED && last_key_wg >= 0 ? p.block_table[bt_off + page_slot(min(16 * i, kv_len - 1), p.block_size, p.block_shift)] : 0;
      rec[0] = last_key_wg < 0 ? -1 : last_key_wg / KT + 1;
    }
    __syncthreads();
  }
  // Which of the CU's two workgroups has more to do: tile steps of all its items, a third more per step for the younger
  // workgroup (the older one wins the arbitration for the vector issue slots: 1.5 against 2.0 us per step), two steps'
  // worth per item for its seam.  The one that would finish later gets the priority (s_setprio), for the whole launch.
  int work_mine = 0, work_partner = 0;
  {
    const bool young = (int)blockIdx.x >= p.n_cus;
    int* p_tiles = reinterpret_cast<int*>(smem) + 3 * p.batch + 2;
    for (int r = 0; r < n_rounds && !units; ++r) {
      const int a = max(__builtin_amdgcn_readfirstlane(items[r * REC]), 0), b = __builtin_amdgcn_readfirstlane(p_tiles[r]);
      work_mine += a ? a * (young ? 4 : 3) + 6 : 0;
      work_partner += b ? b * (young ? 3 : 4) + 6 : 0;
    }
    __syncthreads();                   // (p_tiles lives in the tile images: read before the first requests)
  }
  stamp(4);
  // the first record at or after round r that is an item; those without keys are finished on the way
  int ri = 0;
  auto next_item = [&](int r) __attribute__((always_inline)) -> int {
    for (; r < n_rounds; ++r) {
      const int n = __builtin_amdgcn_readfirstlane(items[r * REC]);
      if (n > 0) break;
      if (n < 0) {
        c_params* P = kargs();
        const int h = items[r * REC + 1], q_row0_wg = items[r * REC + 2], q_start = items[r * REC + 3];
        for (int i = threadIdx.x; i < TQ * LPR; i += 256) {
          const int row = q_row0_wg + i / LPR;
          if (row >= 0)
            *reinterpret_cast<u16x8*>(reinterpret_cast<u16*>(P->out) + (int64_t)(q_start + row) * P->o_row_stride +
                                      (int64_t)h * D + 8 * (i % LPR)) = u16x8{0, 0, 0, 0, 0, 0, 0, 0};
        }
      }
    }
    return r;
  };
  auto read_item = [&](int r, Work& wk) __attribute__((always_inline)) {
    typedef int i32x4 __attribute__((ext_vector_type(4)));
    const i32x4 a = *reinterpret_cast<const i32x4*>(items + r * REC), b = *reinterpret_cast<const i32x4*>(items + r * REC + 4);
    wk.n_tiles = __builtin_amdgcn_readfirstlane(a[0]); wk.h = __builtin_amdgcn_readfirstlane(a[1]);
    wk.q_row0_wg = __builtin_amdgcn_readfirstlane(a[2]); wk.q_start = __builtin_amdgcn_readfirstlane(a[3]);
    wk.q_len = __builtin_amdgcn_readfirstlane(b[0]); wk.k_start = __builtin_amdgcn_readfirstlane(b[1]);
    wk.kv_len = __builtin_amdgcn_readfirstlane(b[2]); wk.bt_off = __builtin_amdgcn_readfirstlane(b[3]);
  };

  // ---- per-item state: the uniform part lives in `cur`, the per-lane part here
  Work cur;
  int q_row0 = 0, limit_c = 0, last_key_wave = 0, last_key = 0, shift = 0, t_last = 0;
  const u16 *kbase = nullptr, *vbase = nullptr;
  c_i32* bt_s = nullptr;
  auto setup = [&](const Work& wk) __attribute__((always_inline)) {
    c_params* P = kargs();
    q_row0 = wk.q_row0_wg + w * 32;
    shift = wk.kv_len - wk.q_len;
    limit_c = P->causal ? min(wk.kv_len - 1, q_row0 + c + shift) : wk.kv_len - 1;
    last_key_wave = q_row0 + 31 < 0 ? -1 : P->causal ? min(wk.kv_len - 1, q_row0 + 31 + shift) : wk.kv_len - 1;
    last_key = wk.kv_len - 1;
    t_last = wk.n_tiles - 1;
    const int hk = wk.h / sfresh(P->group);
    kbase = reinterpret_cast<const u16*>(P->k) + (int64_t)hk * P->k_head_stride;
    vbase = reinterpret_cast<const u16*>(P->v) + (int64_t)hk * P->v_head_stride;
    bt_s = PAGED ? (c_i32*)(P->block_table + wk.bt_off) : nullptr;
  };

  // ---- K / V tile staging by LDS-DMA (see attn_fwd32_kernel).  The address of a request is a wave-uniform base — page,
  // first row of the wave's 16-key group, the instruction's rows: scalar arithmetic — plus a per-lane offset that never
  // changes (row inside the instruction, swizzled chunk): the vector ALU is not involved, except in a sequence's last
  // group, whose rows past the last key are clamped to it.
  int page_next = 0;
  auto lookup_page = [&](int t) __attribute__((always_inline)) {
    if (PAGED) page_next = bt_s[__builtin_amdgcn_readfirstlane(page_slot(min(t * KT + 16 * w, last_key), p.block_size, p.block_shift))];
  };
  uint32_t koff[NL], voff;
#pragma unroll
  for (int j = 0; j < NL; ++j)
    koff[j] = (uint32_t)st_r4 * (uint32_t)p.k_row_stride * 2u + 16u * (uint32_t)(st_ch ^ kswz(RPI * j + st_r4));
  voff = (uint32_t)st_r4 * (uint32_t)p.v_row_stride * 2u + 16u * (uint32_t)(st_ch ^ vswz(st_r4));   // vswz(RPI j + r) = vswz(r)
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Winline-asm"
  auto dma = [&](const void* base, uint32_t off, uint32_t lds) __attribute__((always_inline)) {
    // (m0 — the LDS address of the DMA — is a reserved register: naming it as clobbered is all that can be done, and
    // nothing else in this kernel uses it)
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" :: "v"(off), "s"(base), "s"(lds) : "memory", "m0");
  };
  auto dma_flat = [&](const void* addr, uint32_t lds) __attribute__((always_inline)) {
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" :: "v"(addr), "s"(lds) : "memory", "m0");
  };
#pragma clang diagnostic pop
  auto request_tile = [&](int t, int img) __attribute__((always_inline)) {
    const int g0 = min(t * KT + 16 * w, last_key);      // first key of the wave's group, clamped: uniform
    const int r_max = last_key - g0;                     // rows of the group that exist (>= 0)
    int64_t kb, vb;
    const int64_t krs = sfresh(p.k_row_stride), vrs = sfresh(p.v_row_stride);
    if (PAGED) {
      const int rowg = page_row(g0, p.block_size, p.block_shift);
      kb = (int64_t)page_next * p.k_block_stride + (int64_t)rowg * krs;
      vb = (int64_t)page_next * p.v_block_stride + (int64_t)rowg * vrs;
    } else {
      kb = (int64_t)(cur.k_start + g0) * krs;
      vb = (int64_t)(cur.k_start + g0) * vrs;
    }
    const uint32_t kd = __builtin_amdgcn_readfirstlane(lds0 + (uint32_t)(img * IMG + 16 * w * RSK));
    const uint32_t vd = kd + KTILE;
    if (r_max >= 15) {
      const u16* kp = kbase + kb;
      const u16* vp = vbase + vb;
#pragma unroll
      for (int j = 0; j < NL; ++j) {
        dma(kp, koff[j], kd + 1024u * j);
        dma(vp, voff, vd + 1024u * j);
        kp = sfresh(kp + RPI * krs);         // (step by step: no table of j x stride products)
        vp = sfresh(vp + RPI * vrs);
      }
    } else {
#pragma unroll
      for (int j = 0; j < NL; ++j) {
        const int R = RPI * j + st_r4;
        const uint32_t r = (uint32_t)min(R, r_max);                  // rows past the last key repeat it
        dma_flat(kbase + kb + r * (uint32_t)p.k_row_stride + 8 * (st_ch ^ kswz(R)), kd + 1024u * j);
        dma_flat(vbase + vb + r * (uint32_t)p.v_row_stride + 8 * (st_ch ^ vswz(R)), vd + 1024u * j);
      }
    }
  };
  // the 32 Q rows of this wave for item wk -> rows 32 w .. of image img (K's swizzle), whole rows per instruction
  auto request_q = [&](const Work& wk, int img) __attribute__((always_inline)) {
    c_params* P = kargs();
    const u16* qbase = reinterpret_cast<const u16*>(P->q) + (int64_t)wk.h * D;
    const uint32_t qd = __builtin_amdgcn_readfirstlane(lds0 + (uint32_t)(img * IMG + w * 32 * RSK));
#pragma unroll
    for (int j = 0; j < NQI; ++j) {
      const int R = RPI * j + st_r4;
      const int qr = min(max(wk.q_row0_wg + w * 32 + R, 0), wk.q_len - 1);
      dma_flat(qbase + (int64_t)(wk.q_start + qr) * P->q_row_stride + 8 * (st_ch ^ kswz(R)), qd + 1024u * j);
    }
  };
  auto tiles_landed = [&]() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); };
  using Set0 = std::integral_constant<int, 0>;
  using Set1 = std::integral_constant<int, 1>;

  u16x8 qf[KS];
  auto read_q = [&](int img) __attribute__((always_inline)) {
    const char* qb = smem + img * IMG + w * 32 * RSK;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) qf[ks] = *reinterpret_cast<const u16x8*>(qb + c * RSK + 16 * ((2 * ks + hi) ^ kswz(c)));
  };

  f32x16 acc[NDB];
  float m, l;
  auto reset_acc = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < NDB; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    m = HX_NEG_BIG;
    l = 0.f;
  };

  // ---- first item
  ri = next_item(0);
  if (ri >= n_rounds) return;
  read_item(ri, cur);
  stamp(5);
  setup(cur);
  request_q(cur, 1);
  page_next = __builtin_amdgcn_readfirstlane(items[ri * REC + 8 + w]);
  request_tile(0, 0);
  stamp(6);
  int rn = next_item(ri + 1);                 // the round of the item after this one (n_rounds: none)
  lookup_page(min(1, t_last));
  reset_acc();
  stamp(2);
  tiles_landed();
  stamp(3);
  read_q(1);
  // wg_priority: 0 = equal priorities, 1 = the workgroup of the CU with more left to do (above) runs at s_setprio 3
  const bool use_prio = p.wg_priority != 0;
  const bool favoured = use_prio && (work_mine > work_partner || (work_mine == work_partner && (int)blockIdx.x >= p.n_cus));
  __syncthreads();
  if (favoured) __builtin_amdgcn_s_setprio(3);

  // transposed-read lane address inside a 4-row x 32-dim block (see attn_fwd32_kernel)
  const int tr_q = (lane & 15) >> 2, tr_pp = lane & 3, tr_half = (lane >> 4) & 1;
  const int tr_off = (4 * hi + tr_q) * RSV + (16 * tr_half + 4 * tr_pp) * 2;
  const int tr_x = vswz(tr_q) >> 2;

  auto half_max = [](float x) __attribute__((always_inline)) {      // max with lane ^ 32, without the LDS crossbar
    const uint32_t u = __builtin_bit_cast(uint32_t, x);
    auto r = __builtin_amdgcn_permlane32_swap(u, u, false, false);
    return fmaxf(__builtin_bit_cast(float, (uint32_t)r[0]), __builtin_bit_cast(float, (uint32_t)r[1]));
  };
  auto half_sum = [](float x) __attribute__((always_inline)) {
    const uint32_t u = __builtin_bit_cast(uint32_t, x);
    auto r = __builtin_amdgcn_permlane32_swap(u, u, false, false);
    return __builtin_bit_cast(float, (uint32_t)r[0]) + __builtin_bit_cast(float, (uint32_t)r[1]);
  };

  int t = 0;
  // one 64-key tile of the current item, in image PAR; returns true when the workgroup has no more work
  auto tile_step = [&](auto par_tag) __attribute__((always_inline)) -> bool {
    constexpr int PAR = decltype(par_tag)::value;
    const bool last = t == t_last;
    stamp(10);
    if (!last) {
      request_tile(t + 1, 1 - PAR);                     // lands under this tile's arithmetic
    } else if (rn < n_rounds) {
      Work nx;
      read_item(rn, nx);
      request_q(nx, 1 - PAR);                           // the free image takes the next item's Q rows
    }
    if (t * KT <= last_key_wave) {
      const char* kt = smem + PAR * IMG;
      const char* vt = kt + KTILE;
      f32x16 s[2];
#pragma unroll
      for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int r = 0; r < 16; ++r) s[u][r] = 0.f;
      {
        const char* krd = kt + c * RSK;
        const int kz = kswz(c);
        u16x8 kfa[KS], kfb[KS];
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) kfa[ks] = *reinterpret_cast<const u16x8*>(krd + 16 * ((2 * ks + hi) ^ kz));
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
          kfb[ks] = *reinterpret_cast<const u16x8*>(krd + 32 * RSK + 16 * ((2 * ks + hi) ^ kz));
          s[0] = Mfma32<T>::mma(kfa[ks], qf[ks], s[0]);
        }
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) s[1] = Mfma32<T>::mma(kfb[ks], qf[ks], s[1]);
        __builtin_amdgcn_sched_group_barrier(0x100, KS, 0);
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        }
        __builtin_amdgcn_sched_group_barrier(0x008, KS, 0);
      }
      const bool interior = t * KT + KT - 1 <= min(last_key, p.causal ? q_row0 + shift : last_key);
      float mx = HX_NEG_BIG;
      u16x8 pf[2][2];
      if (interior) {
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
          for (int r = 0; r < 16; ++r) mx = fmaxf(mx, s[u][r]);
      } else {
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int key = t * KT + 32 * u + (r & 3) + 8 * (r >> 2) + 4 * hi;
            if (key > limit_c) s[u][r] = -INFINITY;
            mx = fmaxf(mx, s[u][r]);
          }
      }
      mx = half_max(mx);
      // lazy running maximum (see attn_fwd32_kernel)
      const float m_cand = fmaxf(m, mx * p.scale_log2);
      const bool grow = m_cand > m + 8.0f;
      float m_new = m;
      if (__builtin_amdgcn_ballot_w64(grow)) {
        m_new = grow ? m_cand : m;
        const float alpha = fast_exp2(m - m_new);
        l *= alpha;
#pragma unroll
        for (int i = 0; i < NDB; ++i)
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[i][r] *= alpha;
        m = m_new;
      }
      // (single-instruction fma / add: a packed f32 instruction beside MFMAs costs more than the two it replaces —
      // MI355X_MICROARCH.md, per-instruction constants; the file is compiled without the SLP vectorizer for the same reason)
      float pa = 0.f, pb = 0.f;
#pragma unroll
      for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int r = 0; r < 16; r += 2) {
          const float e0 = fast_exp2(fmaf(s[u][r], p.scale_log2, -m_new));
          const float e1 = fast_exp2(fmaf(s[u][r + 1], p.scale_log2, -m_new));
          pa += e0;
          pb += e1;
          pf[u][r >> 3][r & 7] = T::from_float(e0);
          pf[u][r >> 3][(r & 7) + 1] = T::from_float(e1);
        }
      l += pa + pb;
#pragma unroll
      for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int k2 = 0; k2 < 2; ++k2) {
          const char* vrd = vt + (32 * u + 16 * k2) * RSV + tr_off;
#pragma unroll
          for (int db = 0; db < NDB; ++db) {
            const u16x4 lo = lds_tr_read(vrd + 64 * (db ^ tr_x));
            const u16x4 hh = lds_tr_read(vrd + 8 * RSV + 64 * (db ^ tr_x));
            u16x8 vf;
            vf[0] = lo[0]; vf[1] = lo[1]; vf[2] = lo[2]; vf[3] = lo[3];
            vf[4] = hh[0]; vf[5] = hh[1]; vf[6] = hh[2]; vf[7] = hh[3];
            acc[db] = Mfma32<T>::mma(vf, pf[u][k2], acc[db]);
          }
        }
    }
    stamp(11);
    if (!last) {
      lookup_page(min(t + 2, t_last));     // the table entry for the next request, in front of the barrier's wait
      tiles_landed();                      // this wave's part of tile t + 1 is in LDS
      stamp(12);
      __syncthreads();                     // tile t + 1 visible; everyone is done with tile t's image
      ++t;
      return false;
    }
    // ---- the seam: the item is complete.  Image 1 - PAR holds the next item's Q rows (this wave's own 32), image PAR
    // the tile just used.
    tiles_landed();
    stamp(20);
    const bool has_next = rn < n_rounds;
    if (has_next) read_q(1 - PAR);
    const int e_q_row0 = q_row0;
    const int64_t e_row = (int64_t)cur.q_start + q_row0;
    const int e_h = cur.h;
    __syncthreads();                       // everybody is done with both images
    stamp(21);
    if (has_next) {
      ri = rn;
      read_item(ri, cur);
      setup(cur);
      page_next = __builtin_amdgcn_readfirstlane(items[ri * REC + 8 + w]);      // looked up when the table was built
      request_tile(0, 1 - PAR);            // in flight while the finished item's rows go out
    }
    // O[query c][dim 32 db + 8 (r >> 2) + 4 hi + (r & 3)] = acc[db][r] / L, through LDS so that a store instruction
    // writes whole rows (see attn_fwd32_kernel); this wave's 32 rows sit in image PAR
    {
      const float lr = half_sum(l);
      const float inv = (lr > 0.f) ? 1.0f / lr : 0.f;
      if (e_q_row0 + 31 >= 0) {
        constexpr int RSO = 2 * D;
        // (the lane's addresses are derived HERE from a lane id the compiler cannot see through: as loop invariants they
        // were computed at kernel entry, spilled, and every reload waited for vmcnt(0) — the first tile's requests)
        int ln = lane;
        asm volatile("" : "+v"(ln));
        const int ec = ln & 31, ehi = ln >> 5, er4 = ln / LPR, ech = ln % LPR;
        char* ob = smem + PAR * IMG + w * 32 * RSO;
#pragma unroll
        for (int db = 0; db < NDB; ++db)
#pragma unroll
          for (int rq = 0; rq < 4; ++rq) {
            u16x4 o;
#pragma unroll
            for (int i = 0; i < 4; ++i) o[i] = T::from_float(acc[db][4 * rq + i] * inv);
            *reinterpret_cast<u16x4*>(ob + ec * RSO + (((8 * db + 2 * rq + ehi) ^ (ec & (LPR - 1))) << 3)) = o;
          }
        c_params* P = kargs();
        const int64_t ors = P->o_row_stride;
        u16* orow = reinterpret_cast<u16*>(P->out) + (int64_t)e_h * D + 8 * ech + (e_row + er4) * ors;
        const char* ord = ob + er4 * RSO;
        auto o_row = [&](int j) __attribute__((always_inline)) {
          const int rl = RPI * j + er4;
          u16x8 v = *reinterpret_cast<const u16x8*>(ord + RPI * j * RSO + 16 * (ech ^ ((rl & (LPR - 1)) >> 1)));
          if (rl & 1) v = u16x8{v[4], v[5], v[6], v[7], v[0], v[1], v[2], v[3]};
          return v;
        };
        if (e_q_row0 >= 0) {             // all 32 rows exist: NQI store instructions, no predicate
#pragma unroll
          for (int j = 0; j < NQI; ++j) *reinterpret_cast<u16x8*>(orow + (int64_t)(RPI * j) * ors) = o_row(j);
        } else {
#pragma unroll
          for (int j = 0; j < NQI; ++j)
            if (e_q_row0 + RPI * j + er4 >= 0) *reinterpret_cast<u16x8*>(orow + (int64_t)(RPI * j) * ors) = o_row(j);
        }
      }
    }
    stamp(22);
    if (!has_next) { stamp_cycles(31); return true; }
    reset_acc();
    t = 0;
    rn = next_item(ri + 1);
    lookup_page(min(1, t_last));
    // The first tile's requests are older than the O stores: when all NQI of them were issued (a wave whose 32 rows all
    // exist), waiting for everything but the NQI youngest operations waits for the tile and not for the stores.
    // The rule this rests on (MI355X_MICROARCH.md, "s_waitcnt vmcnt(N)"; GFX9 ISA, S_WAITCNT): a wave's loads, stores,
    // atomics and LDS-DMA count on ONE counter and complete IN ISSUE ORDER — with the single exception of flat_*
    // instructions, of which this translation unit has none (global_ / buffer_ only: tests/test_kernel_resources.py holds
    // the built code object to that).  The 60-case fuzz of the persistent form against the per-item one
    // (tests/test_gpu_attention.py::test_prefill_persistent_fuzz) runs with every wave shape this seam sees.
    stamp(23);
    if (e_q_row0 >= 0 && !STAMPS) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(NQI) : "memory");
    else tiles_landed();
    stamp(24);
    __syncthreads();
    return false;
  };
  for (;;) {
    if (tile_step(Set0{})) break;
    if (tile_step(Set1{})) break;
  }
}

int fwd_n_cus() {
  static int n = [] {
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return 256;
    return prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
  }();
  return n;
}

int g_fwd_ablate = 0;   // EXPERIMENTS builds: timing ablations of attn_fwd32_kernel (wrong results)

unsigned long long* g_fwd_stamps = nullptr;   // EXPERIMENTS builds: hx_debug_fwd_stamps
int g_fwd_pairing = 1;      // tuning: 0 = the second workgroup of a CU takes its items in list order like the first
int g_fwd_units = -1;       // tuning: -1 = automatic, 0 / 1 = the persistent kernel deals single tiles / units of two
int g_fwd_seq_group = 0;    // tuning: 0 = automatic, 1 / 2 / 4 = sequences per deal group of the persistent kernel
int g_fwd_persistent = 1;   // tuning: 0 = one workgroup per (sequence, query tile, head) item, 2 = persistent for dense launches too
int g_fwd_priority = -1;    // tuning: -1 = automatic, 0 / 1 = the two workgroups of a CU at equal / different priorities

template <typename T, int D, bool PAGED>
int launch_fwd32(const AttnParams& p, int batch, hipStream_t stream) {
  const size_t lds = 2 * 64 * (2 * D + 2 * D) + 16;      // two tiles' K / V images + the workgroup's priority flag
  dim3 grid((unsigned)(p.total_q / 128 + batch), p.n_heads, 1);
  AttnParams pp = p;
  pp.wg_priority = (p.total_q / 128) * (int64_t)p.n_heads > 2 * (int64_t)fwd_n_cus() ? 1 : 0;   // more than two workgroups per CU
  if (g_fwd_priority >= 0) pp.wg_priority = g_fwd_priority;
  if (p.total_q == 0) return HX_OK;
  const int per_item_priority = pp.wg_priority;
  // (persistent workgroups: priority to the workgroup of a CU that holds the longer item — 32 x 704 tokens 227 us against
  // 238 at equal priorities, 2048 of 4096 112.7 against 113.5, three processes each)
  if (g_fwd_priority < 0) pp.wg_priority = 1;
  pp.n_tile_slots = (int32_t)(p.total_q / 128 + batch);
  pp.n_cus = fwd_n_cus();
  const int64_t total = (int64_t)pp.n_tile_slots * p.n_heads;
  const int64_t g = std::min<int64_t>(total, 2 * (int64_t)fwd_n_cus());
  // the workgroup's item table: the slot count in front of every group of 4 sequences, 12 words per round
  pp.seq_group = g_fwd_seq_group > 0 ? g_fwd_seq_group : (batch <= 4 ? 4 : 1);
  pp.cu_pairing = g_fwd_pairing;
  // units (pairs of a sequence's tiles) where there are at least two rounds of them
  const int64_t unit_slots = p.total_q / 256 + batch, total_units = unit_slots * p.n_heads;
  pp.unit_mode = g_fwd_units >= 0 ? g_fwd_units : (pp.seq_group == 1 && total_units >= 2 * g ? 1 : 0);
  if (pp.seq_group != 1) pp.unit_mode = 0;
  if (pp.unit_mode) pp.n_tile_slots = (int32_t)unit_slots;
  const int64_t rounds = pp.unit_mode ? 2 * ((total_units + g - 1) / g) : (total + g - 1) / g;
  const size_t table = 4 * (size_t)((((batch + pp.seq_group - 1) / pp.seq_group + 1 + 3) & ~3) + 12 * rounds);
  // (else one workgroup per item: tables that do not fit; dense launches of equal items — the CLIP tower, 8 x 577:
  // 25.1 us per item against 27.4 — where the static deal puts the second items of a round on the same CUs)
  // (and short launches of short items — up to two rounds of tiles with fewer than 24 key tiles.  Same process, same data,
  // hipGraph of ten launches: 4 x 704 tokens 33.9 / 31.5 us per item (two page permutations) against 34.0 - 35.0 / 32.7 -
  // 33.4 persistent in its four settings, 3 x 683 of 704 27.7 / 26.6 against 28.0 - 28.6 / 27.3 - 27.7; 1 x 704, 192
  // items, 20.2 against 21.0: the table is built for little.  Long items gain from the first round on — 2048 new tokens of
  // 4096, one round: 113.6 us against 123.9 — through the pairing of long with short items on a CU.)
  const bool worth_it = total > 4 * (int64_t)fwd_n_cus() || (total > fwd_n_cus() && p.max_seqlen_k >= 24 * 64);
  if ((g_fwd_persistent == 2 || (g_fwd_persistent && PAGED && worth_it)) && 2 * (lds + table) <= 160 * 1024 &&
      4 * (3 * (size_t)batch + 2 + (size_t)rounds) <= 2 * 64 * (2 * D + 2 * D)) {
    const size_t lds = 2 * 64 * (2 * D + 2 * D) + 16 + table;
    hipError_t e = hipFuncSetAttribute((const void*)attn_fwd32p_kernel<T, D, PAGED>,
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return hip_rc(e);
#if HX_EXPERIMENTS
    if constexpr (D == 128 && PAGED && std::is_same<T, BF16>::value) {
      if (g_fwd_stamps) {
        pp.stamps = g_fwd_stamps;
        e = hipFuncSetAttribute((const void*)attn_fwd32p_kernel<T, D, PAGED, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return hip_rc(e);
        hx::launcher(attn_fwd32p_kernel<T, D, PAGED, true>, dim3((unsigned)g, 1, 1), 256, lds, stream)(pp);
        return check_launch();
      }
    }
#endif
    hx::launcher(attn_fwd32p_kernel<T, D, PAGED>, dim3((unsigned)g, 1, 1), 256, lds, stream)(pp);
    return check_launch();
  }
  pp.wg_priority = per_item_priority;
  if (lds > 48 * 1024) {
    hipError_t e = hipFuncSetAttribute((const void*)attn_fwd32_kernel<T, D, PAGED>,
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return hip_rc(e);
  }
#if HX_EXPERIMENTS
  if constexpr (D == 128 && PAGED && std::is_same<T, BF16>::value) {
    switch (g_fwd_ablate) {
#define HX_ABL(n) case n: hx::launcher(attn_fwd32_kernel<T, D, PAGED, n>, grid, 256, lds, stream)(pp); return check_launch();
      HX_ABL(1) HX_ABL(2) HX_ABL(4) HX_ABL(8) HX_ABL(3) HX_ABL(5) HX_ABL(6) HX_ABL(7) HX_ABL(9) HX_ABL(15) HX_ABL(16) HX_ABL(32) HX_ABL(48) HX_ABL(256) HX_ABL(79) HX_ABL(512) HX_ABL(1039)
#undef HX_ABL
      default: break;
    }
  }
#endif
  hx::launcher(attn_fwd32_kernel<T, D, PAGED>, grid, 256, lds, stream)(pp);
  return check_launch();
}

int g_fwd_mfma32 = 1;   // tuning: 0 = always the 16x16x32 kernel

template <typename T, int D, bool PAGED, int QR, int KU>
int launch_fwd_cfg(const AttnParams& p, int batch, int max_seqlen_q, hipStream_t stream) {
  constexpr int RS = 2 * D + 32;
  const size_t lds = 4 * 32 * KU * RS;   // K[2][KT][RS] + V[2][KT][RS]
  // tile slots: sum_b ceil(q_b / TQ) <= total_q / TQ + batch
  dim3 grid((unsigned)(p.total_q / (64 * QR) + batch), p.n_heads, 1);
  if (p.total_q == 0) return HX_OK;
  if (lds > 48 * 1024) {
    hipError_t e = hipFuncSetAttribute((const void*)attn_fwd_kernel<T, D, PAGED, QR, KU>,
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return hip_rc(e);
  }
  hx::launcher(attn_fwd_kernel<T, D, PAGED, QR, KU>, grid, 256, lds, stream)(p);
  return check_launch();
}

int g_fwd_rows = 0;   // tuning: 0 = automatic, 1 / 2 = row blocks per wave
int g_fwd_keys = 0;   // tuning: 0 = automatic, 1 / 2 = 32-key units per tile

template <typename T, int D, bool PAGED>
int launch_fwd_paged(const AttnParams& p, int batch, int max_seqlen_q, hipStream_t stream) {
  // Tiling choice, measured on MI355X with the XCD-aware numbering on (tools/bench_attn_prefill.py):
  //  * two row blocks per wave (QR = 2) no longer pay once a head's tiles share an L2
  //    (2048 new tokens: 116 us with one row block, 130 with two) — kept as an option only;
  //  * 64-key tiles (KU = 2) win where the launch is a few long dependency chains — up to ~768
  //    workgroups (1 x 704 tokens: 30 vs 35 us) — and for the dense CLIP batches (8 x 577: 34 vs 36);
  //    with more workgroups in flight 32-key tiles and their higher occupancy are ahead
  //    (3 x 683: 53 vs 56, 1 x 2048: 116 vs 119).
  if constexpr (D == 64 || D == 128) {
    // more than 64 query rows in some sequence: prefill / vision tower -> 32 rows per wave
    if (g_fwd_mfma32 && max_seqlen_q > 64 && p.window_left < 0 && p.softcap_scale == 0.f)
      return launch_fwd32<T, D, PAGED>(p, batch, stream);
  }
  bool two = false;
  if (g_fwd_rows == 2 && D <= 128) two = true;
  const int64_t n_wg = (p.total_q / 64 + batch) * (int64_t)p.n_heads;
  bool wide = D <= 128 && max_seqlen_q > 64 && (!PAGED || n_wg <= 768);
  if (g_fwd_keys == 1) wide = false;
  if (g_fwd_keys == 2 && D <= 128) wide = true;
  if constexpr (D <= 128) {
    if (two && wide) return launch_fwd_cfg<T, D, PAGED, 2, 2>(p, batch, max_seqlen_q, stream);
    if (two) return launch_fwd_cfg<T, D, PAGED, 2, 1>(p, batch, max_seqlen_q, stream);
    if (wide) return launch_fwd_cfg<T, D, PAGED, 1, 2>(p, batch, max_seqlen_q, stream);
  }
  return launch_fwd_cfg<T, D, PAGED, 1, 1>(p, batch, max_seqlen_q, stream);
}

template <typename T, int D>
int launch_fwd(const AttnParams& p, int batch, int max_seqlen_q, bool paged, hipStream_t stream) {
  return paged ? launch_fwd_paged<T, D, true>(p, batch, max_seqlen_q, stream)
               : launch_fwd_paged<T, D, false>(p, batch, max_seqlen_q, stream);
}

}  // namespace

namespace hx {

int fwd_set_option(const char* name, int value) {
  if (!strcmp(name, "fwd_row_blocks")) { g_fwd_rows = value; return HX_OK; }
  if (!strcmp(name, "fwd_key_units")) { g_fwd_keys = value; return HX_OK; }
  if (!strcmp(name, "fwd_mfma32")) { g_fwd_mfma32 = value ? 1 : 0; return HX_OK; }
  if (!strcmp(name, "fwd_ablate")) { g_fwd_ablate = value; return HX_OK; }
  if (!strcmp(name, "fwd_persistent")) { g_fwd_persistent = value; return HX_OK; }
  if (!strcmp(name, "fwd_priority")) { g_fwd_priority = value; return HX_OK; }
  if (!strcmp(name, "fwd_pairing")) { g_fwd_pairing = value ? 1 : 0; return HX_OK; }
  if (!strcmp(name, "fwd_units")) { g_fwd_units = value < 0 ? -1 : value > 2 ? 1 : value; return HX_OK; }
  if (!strcmp(name, "fwd_seq_group")) { g_fwd_seq_group = value == 1 || value == 2 || value == 4 ? value : 0; return HX_OK; }
  return HX_ERR_UNSUPPORTED;
}

void fwd_set_stamps(void* buf) { g_fwd_stamps = reinterpret_cast<unsigned long long*>(buf); }
void* fwd_get_stamps() { return g_fwd_stamps; }

bool fwd_supported(int head_dim) {
  return head_dim == 32 || head_dim == 64 || head_dim == 96 || head_dim == 128 || head_dim == 256;
}

int launch_attn_fwd(const AttnParams& p, int batch, int head_dim, int max_seqlen_q, bool paged,
                    int dtype, hipStream_t stream) {
#define HX_FWD_CASE(TT, DD) \
  case DD: return launch_fwd<TT, DD>(p, batch, max_seqlen_q, paged, stream);
  if (dtype == HX_F16) {
    switch (head_dim) {
      HX_FWD_CASE(F16, 32) HX_FWD_CASE(F16, 64) HX_FWD_CASE(F16, 96) HX_FWD_CASE(F16, 128)
      HX_FWD_CASE(F16, 256)
    }
  } else if (dtype == HX_BF16) {
    switch (head_dim) {
      HX_FWD_CASE(BF16, 32) HX_FWD_CASE(BF16, 64) HX_FWD_CASE(BF16, 96) HX_FWD_CASE(BF16, 128)
      HX_FWD_CASE(BF16, 256)
    }
  } else {
    return HX_ERR_DTYPE;
  }
#undef HX_FWD_CASE
  return HX_ERR_SHAPE;
}

}  // namespace hx
