// hx_api.hip — C-ABI entry points that are not tied to one kernel file: status text,
// argument validation + dispatch of hx_mha_varlen_fwd (error behaviour mirrors the
// TORCH_CHECKs of csrc/kernel/flash_attn/flash_api.cpp:236-283 of the reference).
#include <cstring>
#include "attn_common.h"

namespace hx {

int& last_hip_error() {
  static thread_local int e = 0;
  return e;
}

bool decode_supported(int head_dim);
int decode_set_option(const char* name, int value);
int gemm_set_option(const char* name, int value);
int fwd_set_option(const char* name, int value);
int xreg_set_option(const char* name, int value);
void fwd_set_stamps(void* buf);
void* fwd_get_stamps();
#if HX_EXPERIMENTS   // `make EXPERIMENTS=1`: rejected experiments kept measurable (not in the default library)
int decode4_set_option(const char* name, int value);
bool decode4_applies(const AttnParams& p, int batch, int head_dim, int n_cus);
int launch_attn_decode4(const AttnParams& p, int batch, int dtype, hipStream_t stream);
#endif
bool fwd_supported(int head_dim);
int decode_pick_splits(int batch, int n_heads, int max_seqlen_k, int requested);
bool decode_gqa_supported(int head_dim, int group);
int decode_gqa_pick_splits(int batch, int n_kv_heads, int max_seqlen_k, int requested);
int launch_attn_decode_gqa(const AttnParams& p, int batch, int head_dim, int dtype, hipStream_t stream);
int launch_attn_decode(const AttnParams& p, int batch, int head_dim, int dtype,
                       hipStream_t stream);
int launch_attn_fwd(const AttnParams& p, int batch, int head_dim, int max_seqlen_q, bool paged,
                    int dtype, hipStream_t stream);

}  // namespace hx

using namespace hx;

extern "C" int hx_abi_version(void) { return HX_ABI_VERSION; }

extern "C" int hx_last_hip_error(void) { return last_hip_error(); }

static int g_fwd_xcd = 1;      // tuning: XCD-aware workgroup numbering in the prefill kernel
static int g_decode_gqa = 1;   // tuning: 0 routes grouped-query decode through the per-query-head kernel

extern "C" int hx_debug_set_option(const char* name, int value) {
  if (!name) return HX_ERR_NULL;
  if (!strcmp(name, "decode_gqa")) { g_decode_gqa = value ? 1 : 0; return HX_OK; }
  if (!strcmp(name, "fwd_xcd_remap")) { g_fwd_xcd = value ? 1 : 0; return HX_OK; }
  int rc = decode_set_option(name, value);
  if (rc == HX_ERR_UNSUPPORTED) rc = gemm_set_option(name, value);
  if (rc == HX_ERR_UNSUPPORTED) rc = fwd_set_option(name, value);
  if (rc == HX_ERR_UNSUPPORTED) rc = xreg_set_option(name, value);
#if HX_EXPERIMENTS
  if (rc == HX_ERR_UNSUPPORTED) rc = decode4_set_option(name, value);
#endif
  return rc;
}

#if HX_EXPERIMENTS
// time stamps of the persistent prefill kernel: 512 words per workgroup (see attn_fwd.hip, STAMPS); null switches them off
extern "C" int hx_debug_fwd_stamps(void* buf) { fwd_set_stamps(buf); return HX_OK; }
#endif

extern "C" const char* hx_strerror(int status) {
  switch (status) {
    case HX_OK: return "ok";
    case HX_ERR_DTYPE: return "failed to dispatch data type";
    case HX_ERR_SHAPE: return "invalid shape / size argument";
    case HX_ERR_STRIDE: return "unsupported tensor layout (stride / alignment)";
    case HX_ERR_NULL: return "required pointer argument is null";
    case HX_ERR_UNSUPPORTED: return "feature not supported by the MI355X implementation";
    case HX_ERR_WORKSPACE: return "workspace too small";
    case HX_ERR_HIP: return "HIP runtime error";
    case HX_ERR_HANDLE: return "invalid IPC memory handle";
    default: return "unknown hydra_hip status";
  }
}

namespace {

// decode kernel applies when every sequence contributes exactly one query row and the
// cache is paged.
static int device_cus() {
  static int n = [] {
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return 256;
    return prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
  }();
  return n;
}

bool use_decode(const hx_attn_args* a) {
  return a->block_table != nullptr && a->max_seqlen_q == 1 && a->total_q == a->batch &&
         decode_supported(a->head_dim);
}


bool use_gqa(const hx_attn_args* a, bool fused) {
  return g_decode_gqa && !fused && decode_gqa_supported(a->head_dim, a->n_heads / a->n_kv_heads);
}

int pick_splits(const hx_attn_args* a, bool fused) {
  return use_gqa(a, fused) ? decode_gqa_pick_splits(a->batch, a->n_kv_heads, a->max_seqlen_k, a->num_splits)
                           : decode_pick_splits(a->batch, a->n_heads, a->max_seqlen_k, a->num_splits);
}

int validate(const hx_attn_args* a) {
  if (!a) return HX_ERR_NULL;
  if (a->dtype != HX_F16 && a->dtype != HX_BF16) return HX_ERR_DTYPE;
  if (a->batch <= 0) return HX_ERR_SHAPE;
  if (a->n_heads <= 0 || a->n_kv_heads <= 0 || a->n_heads % a->n_kv_heads != 0)
    return HX_ERR_SHAPE;
  if (a->head_dim <= 0 || a->head_dim % 8 != 0 || a->head_dim > 256) return HX_ERR_SHAPE;
  if (!a->cu_seqlens_q || !a->cu_seqlens_k) return HX_ERR_NULL;
  if (a->total_q < 0 || a->max_seqlen_q < 0 || a->max_seqlen_k < 0) return HX_ERR_SHAPE;
  if (a->total_q > 0 && (!a->out || !a->q || !a->k || !a->v)) return HX_ERR_NULL;
  if (a->block_table) {
    if (!a->cu_block_lens) return HX_ERR_NULL;
    if (a->block_size <= 0 || a->block_size % 16 != 0) return HX_ERR_SHAPE;
  }
  if (!fwd_supported(a->head_dim)) return HX_ERR_UNSUPPORTED;
  // 16-byte vector access on q/k/v rows, 8-byte on out
  const int64_t strides[] = {a->q_row_stride, a->k_row_stride, a->v_row_stride, a->k_head_stride,
                             a->v_head_stride};
  for (int64_t s : strides)
    if (s % 8 != 0) return HX_ERR_STRIDE;
  if (a->o_row_stride % 4 != 0) return HX_ERR_STRIDE;
  if (a->block_table && (a->k_block_stride % 8 != 0 || a->v_block_stride % 8 != 0))
    return HX_ERR_STRIDE;
  if (!aligned16(a->q) || !aligned16(a->k) || !aligned16(a->v) ||
      (reinterpret_cast<uintptr_t>(a->out) & 7u))
    return HX_ERR_STRIDE;
  return HX_OK;
}

}  // namespace

static inline bool is_local(const hx_attn_args* a) {
  return (a->flags & HX_ATTN_LOCAL_WINDOW) && (a->window_left >= 0 || a->window_right >= 0);
}

extern "C" int64_t hx_mha_varlen_fwd_workspace_bytes(const hx_attn_args* a) {
  if (validate(a) != HX_OK || !use_decode(a) || a->softcap > 0.f || is_local(a)) return 0;
  // the same query serves hx_mha_varlen_fwd and hx_decode_attention_fused, which may pick
  // different kernels (and split counts) for a grouped-query shape: size for the larger
  const int s1 = pick_splits(a, false), s2 = pick_splits(a, true);
  const int splits = s1 > s2 ? s1 : s2;
  if (splits <= 1) return 0;
  return (int64_t)a->batch * a->n_heads * splits * (a->head_dim + 2) * (int64_t)sizeof(float);
}

static int attn_dispatch(const hx_attn_args* a, const hx_fused_decode_args* fused, hx_stream stream);

extern "C" int hx_mha_varlen_fwd(const hx_attn_args* a, hx_stream stream) {
  return attn_dispatch(a, nullptr, stream);
}

extern "C" int hx_decode_attention_fused(const hx_attn_args* a, const hx_fused_decode_args* f,
                                         hx_stream stream) {
  if (!a || !f) return HX_ERR_NULL;
  if (!f->positions || !f->cos_sin || !f->new_cache_slots) return HX_ERR_NULL;
  if (f->qkv_partial) {
    if (f->qkv_splits < 1 || !aligned16(f->qkv_partial)) return HX_ERR_SHAPE;
    if (a->total_q != a->batch) return HX_ERR_SHAPE;   // slab row b belongs to sequence b
  } else if (!f->k_new || !f->v_new) {
    return HX_ERR_NULL;
  }
  if (!use_decode(a)) return HX_ERR_SHAPE;           // q_len == 1 per sequence, paged cache
  if (f->rotary_dim != a->head_dim || f->interleaved) return HX_ERR_UNSUPPORTED;
  if (!aligned16(f->cos_sin)) return HX_ERR_STRIDE;
  if (!f->qkv_partial && (f->k_new_row_stride % 8 || f->v_new_row_stride % 8 ||
                          !aligned16(f->k_new) || !aligned16(f->v_new)))
    return HX_ERR_STRIDE;
  if (a->k_head_stride != a->head_dim || a->v_head_stride != a->head_dim) return HX_ERR_STRIDE;
  return attn_dispatch(a, f, stream);
}

static int attn_dispatch(const hx_attn_args* a, const hx_fused_decode_args* fused, hx_stream stream) {
  int rc = validate(a);
  if (rc) return rc;
  if (a->total_q == 0) return HX_OK;

  AttnParams p;
  p.out = a->out;
  p.q = a->q;
  p.k = a->k;
  p.v = a->v;
  p.cu_q = a->cu_seqlens_q;
  p.cu_k = a->cu_seqlens_k;
  p.block_table = a->block_table;
  p.cu_block_lens = a->cu_block_lens;
  p.q_row_stride = a->q_row_stride;
  p.o_row_stride = a->o_row_stride;
  p.k_block_stride = a->k_block_stride;
  p.k_row_stride = a->k_row_stride;
  p.k_head_stride = a->k_head_stride;
  p.v_block_stride = a->v_block_stride;
  p.v_row_stride = a->v_row_stride;
  p.v_head_stride = a->v_head_stride;
  p.n_heads = a->n_heads;
  p.batch = a->batch;
  p.total_q = a->total_q;
  p.group = a->n_heads / a->n_kv_heads;
  p.block_size = a->block_table ? a->block_size : 16;
  p.block_shift = (p.block_size & (p.block_size - 1)) == 0 ? __builtin_ctz((unsigned)p.block_size) : -1;
  p.causal = a->causal;
  p.xcd_remap = g_fwd_xcd;
  p.wg_priority = 0;
  p.n_tile_slots = 0;
  p.n_cus = 0;
  p.seq_group = 4;
  p.unit_mode = 0;
  p.cu_pairing = 1;
  p.max_seqlen_k = a->max_seqlen_k;
  p.stamps = nullptr;
#if HX_EXPERIMENTS
  p.stamps = reinterpret_cast<unsigned long long*>(fwd_get_stamps());   // decode kernel: tools/decode_timeline.py
#endif
  p.scale_log2 = a->softmax_scale * 1.4426950408889634f;
  // flash_api.cpp:93-111
  if (a->flags & ~HX_ATTN_LOCAL_WINDOW) return HX_ERR_UNSUPPORTED;
  const bool local = is_local(a);
  p.window_left = local ? a->window_left : -1;
  p.window_right = local ? a->window_right : -1;
  p.softcap_scale = 0.f;
  if (a->softcap > 0.f) {
    p.softcap_scale = a->softmax_scale / a->softcap;
    p.scale_log2 = a->softcap * 1.4426950408889634f;
  }
  if (local && a->causal) return HX_ERR_UNSUPPORTED;
  if ((local || a->softcap > 0.f) && fused) return HX_ERR_UNSUPPORTED;
  if (local) {   // one-sided windows: the open side reaches the end of the sequence
    if (p.window_left < 0) p.window_left = a->max_seqlen_k;
    if (p.window_right < 0) p.window_right = a->max_seqlen_k;
  }
  const bool plain = !local && a->softcap <= 0.f;
  p.n_splits = 1;
  p.ws_o = nullptr;
  p.ws_ml = nullptr;
  p.k_new = nullptr;
  p.v_new = nullptr;
  p.kn_row_stride = p.vn_row_stride = 0;
  p.positions = nullptr;
  p.cos_sin = nullptr;
  p.new_slots = nullptr;
  p.qkv_partial = nullptr;
  p.qkv_splits = 0;
  p.qkv_slab_stride = p.qkv_row = 0;
  p.rank_desc = nullptr;
  if (fused) {
    if (fused->qkv_partial) {
      p.qkv_partial = fused->qkv_partial;
      p.qkv_splits = fused->qkv_splits;
      p.qkv_row = (int64_t)(a->n_heads + 2 * a->n_kv_heads) * a->head_dim;
      p.qkv_slab_stride = (int64_t)a->batch * p.qkv_row;
    }
    p.k_new = fused->k_new;
    p.v_new = fused->v_new;
    p.kn_row_stride = fused->k_new_row_stride;
    p.vn_row_stride = fused->v_new_row_stride;
    p.positions = fused->positions;
    p.cos_sin = fused->cos_sin;
    p.new_slots = fused->new_cache_slots;
    p.rank_desc = fused->rank_desc;
  }

  hipStream_t s = (hipStream_t)stream;
  if (plain && use_decode(a)) {
    // with q_len == 1 the causal mask admits every cached key (bottom-right aligned),
    // so causal and non-causal decode coincide.
    const bool gqa = use_gqa(a, fused != nullptr);
    int splits = pick_splits(a, fused != nullptr);
    if (splits > 1) {
      const int64_t need =
          (int64_t)a->batch * a->n_heads * splits * (a->head_dim + 2) * (int64_t)sizeof(float);
      if (!a->workspace || a->workspace_bytes < need) {
        if (a->num_splits > 1) return HX_ERR_WORKSPACE;
        splits = 1;  // automatic choice degrades gracefully without scratch
      } else {
        p.ws_o = reinterpret_cast<float*>(a->workspace);
        p.ws_ml = p.ws_o + (int64_t)a->batch * a->n_heads * splits * a->head_dim;
      }
    }
    p.n_splits = splits;
    if (gqa) return launch_attn_decode_gqa(p, a->batch, a->head_dim, a->dtype, s);
#if HX_EXPERIMENTS
    // four heads per workgroup (1 KiB contiguous per key row) when the grid still fills the chip
    if (decode4_applies(p, a->batch, a->head_dim, device_cus())) return launch_attn_decode4(p, a->batch, a->dtype, s);
#endif
    return launch_attn_decode(p, a->batch, a->head_dim, a->dtype, s);
  }
  return launch_attn_fwd(p, a->batch, a->head_dim, a->max_seqlen_q, a->block_table != nullptr,
                         a->dtype, s);
}
