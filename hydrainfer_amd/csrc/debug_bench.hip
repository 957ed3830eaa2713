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
  const int64_t wave = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  const int64_t n_waves = (int64_t)gridDim.x * (blockDim.x >> 6);
  const int64_t chunk = (int64_t)U * 1024;                 // bytes per wave per step
  const int64_t n_chunks = bytes / chunk;
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  for (int64_t c = wave; c < n_chunks; c += n_waves) {
    f32x4 v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      int64_t off;
      if (variant == 0) {
        off = c * chunk + (int64_t)u * 1024 + lane * 16;
      } else {
        // the chunk is a [rows_per_chunk][seg] block of a row-major matrix with row pitch `pitch`:
        // one instruction covers R rows x S bytes (R*S = 1024), U instructions walk along the row
        const int S = variant == 1 ? 128 : variant == 2 ? 256 : variant == 3 ? 512 : 1024;
        const int R = 1024 / S;
        const int lanes_per_row = S / 16;
        const int r = lane / lanes_per_row, piece = lane % lanes_per_row;
        // chunk c -> (row block, column block): a row block of R rows spans pitch bytes = pitch / (S*U) chunks
        const int64_t per_rowblock = pitch / ((int64_t)S * U);
        const int64_t rb = c / per_rowblock, cb = c % per_rowblock;
        off = (rb * R + r) * pitch + cb * ((int64_t)S * U) + (int64_t)u * S + piece * 16;
      }
      const f32x4* p = reinterpret_cast<const f32x4*>(base + off);
      if (POLICY == 1) v[u] = __builtin_nontemporal_load(p);
      else v[u] = *p;
    }
#pragma unroll
    for (int u = 0; u < U; ++u) acc += v[u];
  }
  if (acc[0] + acc[1] + acc[2] + acc[3] == 123.456f) sink[0] = acc[0];
}

template <int U>
int launch_u(const void* p, int64_t bytes, int variant, int64_t pitch, int policy, int wgs, float* sink, hipStream_t s) {
  if (policy) stream_read_kernel<U, 1><<<wgs, 256, 0, s>>>((const char*)p, bytes, variant, pitch, sink);
  else stream_read_kernel<U, 0><<<wgs, 256, 0, s>>>((const char*)p, bytes, variant, pitch, sink);
  return check_launch();
}

}  // namespace

extern "C" int hx_debug_stream_read(const void* p, int64_t bytes, int variant, int64_t pitch, int unroll,
                                    int policy, int wgs, float* sink, hx_stream stream) {
  if (!p || !sink || bytes <= 0 || wgs <= 0) return HX_ERR_NULL;
  if (variant < 0 || variant > 4 || (variant && (pitch <= 0 || pitch % 1024))) return HX_ERR_SHAPE;
  hipStream_t s = (hipStream_t)stream;
  switch (unroll) {
    case 4: return launch_u<4>(p, bytes, variant, pitch, policy, wgs, sink, s);
    case 8: return launch_u<8>(p, bytes, variant, pitch, policy, wgs, sink, s);
    case 16: return launch_u<16>(p, bytes, variant, pitch, policy, wgs, sink, s);
    case 32: return launch_u<32>(p, bytes, variant, pitch, policy, wgs, sink, s);
    default: return HX_ERR_SHAPE;
  }
}
