// migration.hip — cache-block migration between E/P/D instances.
// Replaces csrc/data_transfer/block_migration.cpp:55-245 of the reference, which issues
// n_layers * n_tokens * n_blocks cudaMemcpyAsync calls (2816 x 128 KiB for one 704-token
// LLaVA-1.5-7B request) and re-opens the peer IPC handle on every call.
//
// MI355X design: the peer pool is mapped once (hipIpcOpenMemHandle, cached by handle
// bytes) and ONE gather-copy kernel moves every (layer, k/v, block) triple.  The block
// pairs travel as kernel arguments (no H2D table copy, capture-safe).  Each workgroup
// copies one 16-byte-vectorised slice of one block; reads cross xGMI from the peer's
// HBM, writes land in local HBM, so a transfer is bound by one xGMI link.
#include <map>
#include <mutex>
#include <string>
#include <cstring>
#include "hx_common.h"

namespace {

using namespace hx;

struct PairTable {
  int32_t src[HX_MIGRATE_MAX_PAIRS];
  int32_t dst[HX_MIGRATE_MAX_PAIRS];
};

// grid = (slices_per_block, n_pairs, n_layers*n_tokens); 256 threads x 16 B x UNROLL
// src_stride_lt / dst_stride_lt: bytes between consecutive (layer, token-kind) planes.
// src_block < 0 / dst_block < 0 mean "staging index = pair index" (pack / unpack).
constexpr int kUnroll = 4;
__global__ __launch_bounds__(256) void gather_copy_blocks(
    const PairTable tbl, const char* __restrict__ src, char* __restrict__ dst,
    int64_t src_plane_bytes, int64_t dst_plane_bytes, int64_t block_bytes, int src_is_staging,
    int dst_is_staging) {
  const int pair = blockIdx.y;
  const int plane = blockIdx.z;
  const int64_t sb = src_is_staging ? pair : tbl.src[pair];
  const int64_t db = dst_is_staging ? pair : tbl.dst[pair];
  const uint4* s = reinterpret_cast<const uint4*>(src + plane * src_plane_bytes + sb * block_bytes);
  uint4* d = reinterpret_cast<uint4*>(dst + plane * dst_plane_bytes + db * block_bytes);
  const int64_t nvec = block_bytes >> 4;
  const int64_t base = (int64_t)blockIdx.x * (256 * kUnroll) + threadIdx.x;
  uint4 r[kUnroll];
#pragma unroll
  for (int u = 0; u < kUnroll; ++u) {
    const int64_t i = base + u * 256;
    if (i < nvec) r[u] = s[i];
  }
#pragma unroll
  for (int u = 0; u < kUnroll; ++u) {
    const int64_t i = base + u * 256;
    if (i < nvec) d[i] = r[u];
  }
}

int launch_copy(const int32_t* src_tbl, const int32_t* dst_tbl, int64_t n_pairs, const void* src,
                void* dst, int64_t n_planes, int64_t src_plane_bytes, int64_t dst_plane_bytes,
                int64_t block_bytes, bool src_staging, bool dst_staging, hipStream_t stream) {
  if (n_pairs == 0 || n_planes == 0) return HX_OK;
  if (block_bytes <= 0 || block_bytes % 16 != 0) return HX_ERR_SHAPE;
  if (!aligned16(src) || !aligned16(dst)) return HX_ERR_STRIDE;
  if (n_planes > 65535) return HX_ERR_SHAPE;
  const int64_t nvec = block_bytes >> 4;
  const unsigned gx = (unsigned)((nvec + 256 * kUnroll - 1) / (256 * kUnroll));
  for (int64_t p0 = 0; p0 < n_pairs; p0 += HX_MIGRATE_MAX_PAIRS) {
    const int64_t np = n_pairs - p0 < HX_MIGRATE_MAX_PAIRS ? n_pairs - p0 : HX_MIGRATE_MAX_PAIRS;
    PairTable tbl;
    for (int64_t i = 0; i < np; ++i) {
      tbl.src[i] = src_tbl ? src_tbl[p0 + i] : 0;
      tbl.dst[i] = dst_tbl ? dst_tbl[p0 + i] : 0;
    }
    // staging buffers are indexed by absolute pair index: shift their base per chunk
    const char* s = (const char*)src + (src_staging ? p0 * block_bytes : 0);
    char* d = (char*)dst + (dst_staging ? p0 * block_bytes : 0);
    hx::launcher(gather_copy_blocks, dim3(gx, (unsigned)np, (unsigned)n_planes), 256, 0, stream)(
        tbl, s, d, src_plane_bytes, dst_plane_bytes, block_bytes, src_staging ? 1 : 0,
        dst_staging ? 1 : 0);
    int rc = check_launch();
    if (rc) return rc;
  }
  return HX_OK;
}

std::mutex g_ipc_mu;
std::map<std::string, void*> g_ipc_open;  // handle bytes -> mapped base

}  // namespace

extern "C" int hx_ipc_get_mem_handle(const void* dev_ptr, uint8_t handle_out[HX_IPC_HANDLE_BYTES],
                                     int64_t* offset_out) {
  if (!dev_ptr || !handle_out) return HX_ERR_NULL;
  static_assert(sizeof(hipIpcMemHandle_t) == HX_IPC_HANDLE_BYTES, "handle size");
  // the handle names the whole allocation; report where dev_ptr sits inside it
  void* base = nullptr;
  size_t size = 0;
  int rc = hip_rc(hipMemGetAddressRange((hipDeviceptr_t*)&base, &size, (hipDeviceptr_t)dev_ptr));
  if (rc) return rc;
  hipIpcMemHandle_t h;
  rc = hip_rc(hipIpcGetMemHandle(&h, base));
  if (rc) return rc;
  memcpy(handle_out, &h, HX_IPC_HANDLE_BYTES);
  if (offset_out) *offset_out = (const char*)dev_ptr - (const char*)base;
  return HX_OK;
}

extern "C" int hx_ipc_open_mem_handle(const uint8_t handle[HX_IPC_HANDLE_BYTES],
                                      void** dev_ptr_out) {
  if (!handle || !dev_ptr_out) return HX_ERR_NULL;
  std::string key(reinterpret_cast<const char*>(handle), HX_IPC_HANDLE_BYTES);
  std::lock_guard<std::mutex> lk(g_ipc_mu);
  auto it = g_ipc_open.find(key);
  if (it != g_ipc_open.end()) {
    *dev_ptr_out = it->second;
    return HX_OK;
  }
  hipIpcMemHandle_t h;
  memcpy(&h, handle, HX_IPC_HANDLE_BYTES);
  void* p = nullptr;
  hipError_t e = hipIpcOpenMemHandle(&p, h, hipIpcMemLazyEnablePeerAccess);
  if (e != hipSuccess) {
    last_hip_error() = (int)e;
    (void)hipGetLastError();
    return e == hipErrorInvalidValue ? HX_ERR_HANDLE : HX_ERR_HIP;
  }
  g_ipc_open[key] = p;
  *dev_ptr_out = p;
  return HX_OK;
}

extern "C" int hx_ipc_close_all(void) {
  std::lock_guard<std::mutex> lk(g_ipc_mu);
  int rc = HX_OK;
  for (auto& kv : g_ipc_open) {
    int r = hip_rc(hipIpcCloseMemHandle(kv.second));
    if (r) rc = r;
  }
  g_ipc_open.clear();
  return rc;
}

extern "C" int hx_migrate_blocks_planes(const int32_t* src_table_host, const int32_t* dst_table_host,
                                        int64_t n_pairs, const void* src_pool, void* dst_pool, int64_t n_planes,
                                        int64_t src_n_blocks, int64_t dst_n_blocks, int64_t src_plane_bytes,
                                        int64_t dst_plane_bytes, int64_t block_bytes, hx_stream stream) {
  if (n_pairs < 0 || n_planes < 0) return HX_ERR_SHAPE;
  if (n_pairs == 0) return HX_OK;
  if (!src_table_host || !dst_table_host || !src_pool || !dst_pool) return HX_ERR_NULL;
  // a plane holds its pool's blocks back to back; what lies between two planes is the pool's own business
  if (src_plane_bytes < src_n_blocks * block_bytes || dst_plane_bytes < dst_n_blocks * block_bytes) return HX_ERR_STRIDE;
  if (src_plane_bytes % 16 != 0 || dst_plane_bytes % 16 != 0) return HX_ERR_STRIDE;
  for (int64_t i = 0; i < n_pairs; ++i) {
    if (src_table_host[i] < 0 || src_table_host[i] >= src_n_blocks) return HX_ERR_SHAPE;
    if (dst_table_host[i] < 0 || dst_table_host[i] >= dst_n_blocks) return HX_ERR_SHAPE;
  }
  return launch_copy(src_table_host, dst_table_host, n_pairs, src_pool, dst_pool, n_planes, src_plane_bytes,
                     dst_plane_bytes, block_bytes, false, false, (hipStream_t)stream);
}

extern "C" int hx_migrate_blocks(const int32_t* src_table_host, const int32_t* dst_table_host,
                                 int64_t n_pairs, const void* src_pool, void* dst_pool,
                                 int64_t n_layers, int64_t n_tokens, int64_t src_n_blocks,
                                 int64_t dst_n_blocks, int64_t block_bytes, hx_stream stream) {
  if (n_layers < 0 || n_tokens < 0) return HX_ERR_SHAPE;
  return hx_migrate_blocks_planes(src_table_host, dst_table_host, n_pairs, src_pool, dst_pool, n_layers * n_tokens,
                                  src_n_blocks, dst_n_blocks, src_n_blocks * block_bytes, dst_n_blocks * block_bytes,
                                  block_bytes, stream);
}

extern "C" int hx_pack_blocks_planes(const int32_t* table_host, int64_t n_pairs, const void* pool, void* staging,
                                     int64_t n_planes, int64_t n_blocks, int64_t pool_plane_bytes,
                                     int64_t block_bytes, hx_stream stream) {
  if (n_pairs < 0 || n_planes < 0) return HX_ERR_SHAPE;
  if (n_pairs == 0) return HX_OK;
  if (!table_host || !pool || !staging) return HX_ERR_NULL;
  if (pool_plane_bytes < n_blocks * block_bytes || pool_plane_bytes % 16 != 0) return HX_ERR_STRIDE;
  for (int64_t i = 0; i < n_pairs; ++i)
    if (table_host[i] < 0 || table_host[i] >= n_blocks) return HX_ERR_SHAPE;
  return launch_copy(table_host, nullptr, n_pairs, pool, staging, n_planes, pool_plane_bytes,
                     n_pairs * block_bytes, block_bytes, false, true, (hipStream_t)stream);
}

extern "C" int hx_pack_blocks(const int32_t* table_host, int64_t n_pairs, const void* pool,
                              void* staging, int64_t n_layers, int64_t n_tokens, int64_t n_blocks,
                              int64_t block_bytes, hx_stream stream) {
  return hx_pack_blocks_planes(table_host, n_pairs, pool, staging, n_layers * n_tokens, n_blocks,
                               n_blocks * block_bytes, block_bytes, stream);
}

extern "C" int hx_unpack_blocks_planes(const int32_t* table_host, int64_t n_pairs, const void* staging, void* pool,
                                       int64_t n_planes, int64_t n_blocks, int64_t pool_plane_bytes,
                                       int64_t block_bytes, hx_stream stream) {
  if (n_pairs < 0 || n_planes < 0) return HX_ERR_SHAPE;
  if (n_pairs == 0) return HX_OK;
  if (!table_host || !staging || !pool) return HX_ERR_NULL;
  if (pool_plane_bytes < n_blocks * block_bytes || pool_plane_bytes % 16 != 0) return HX_ERR_STRIDE;
  for (int64_t i = 0; i < n_pairs; ++i)
    if (table_host[i] < 0 || table_host[i] >= n_blocks) return HX_ERR_SHAPE;
  return launch_copy(nullptr, table_host, n_pairs, staging, pool, n_planes, n_pairs * block_bytes,
                     pool_plane_bytes, block_bytes, true, false, (hipStream_t)stream);
}

extern "C" int hx_unpack_blocks(const int32_t* table_host, int64_t n_pairs, const void* staging,
                                void* pool, int64_t n_layers, int64_t n_tokens, int64_t n_blocks,
                                int64_t block_bytes, hx_stream stream) {
  return hx_unpack_blocks_planes(table_host, n_pairs, staging, pool, n_layers * n_tokens, n_blocks,
                                 n_blocks * block_bytes, block_bytes, stream);
}
