// launch_plan.hip — record / replay of a fixed launch sequence (the decode step).
//
// The role a hipGraph plays in the reference's unfinished cuda_graph_model_runner.py
// (hydrainfer/model_runner/cuda_graph_model_runner.py:1-72) and in engine/graph_decode.py: a decode step is
// ~170 launches of 5-70 us kernels whose arguments never change, so they are issued by one native loop instead of
// one Python call each.  A plan is recorded on one thread: between hx_plan_begin and hx_plan_end every hx_* entry
// point called on that thread appends its launches (hx::launcher) to the plan instead of executing them;
// hx_plan_launch issues them in stream order.  Measured against a captured hipGraph of the same launches
// (profiles/r3_launch_chain_experiment.md): equal inside one process, 0.5-1 % faster in a process that has only this
// one replay mechanism.  (Round 3 also tried launching the plan's kernels WITHOUT the AQL barrier bit —
// hipExtAnyOrderLaunch, which a captured hipGraph cannot express — with the dependencies taken inside the kernels:
// correct, real overlap, no net gain; same file.)
#include <cstring>
#include <vector>
#include "hx_common.h"

namespace hx {

struct PlanItem {
  const void* func;
  dim3 grid, block;
  size_t lds;
  std::unique_ptr<ArgHolderBase> args;
};

struct PlanRecorder {
  std::vector<PlanItem> items;
};

static thread_local PlanRecorder* t_recording = nullptr;

PlanRecorder* recording() { return t_recording; }

void record_launch(PlanRecorder* r, const void* func, dim3 grid, dim3 block, size_t lds,
                   std::unique_ptr<ArgHolderBase> args) {
  PlanItem it;
  it.func = func; it.grid = grid; it.block = block; it.lds = lds; it.args = std::move(args);
  r->items.push_back(std::move(it));
}

}  // namespace hx

using namespace hx;

struct hx_plan {
  PlanRecorder rec;
  bool finished = false;
};

extern "C" int hx_plan_begin(hx_plan** plan) {
  if (!plan) return HX_ERR_NULL;
  if (t_recording) return HX_ERR_UNSUPPORTED;          // one recording per thread
  hx_plan* p = new hx_plan();
  *plan = p;
  t_recording = &p->rec;
  return HX_OK;
}

extern "C" int hx_plan_end(hx_plan* plan) {
  if (!plan) return HX_ERR_NULL;
  if (t_recording != &plan->rec) return HX_ERR_UNSUPPORTED;
  t_recording = nullptr;
  plan->finished = true;
  return HX_OK;
}

extern "C" int hx_plan_size(const hx_plan* plan) {
  return plan ? (int)plan->rec.items.size() : HX_ERR_NULL;
}

extern "C" int hx_plan_launch(const hx_plan* plan, hx_stream stream) {
  if (!plan) return HX_ERR_NULL;
  if (!plan->finished || t_recording) return HX_ERR_UNSUPPORTED;
  hipStream_t s = (hipStream_t)stream;
  for (const PlanItem& it : plan->rec.items) {
    hipError_t e = hipLaunchKernel(it.func, it.grid, it.block, it.args->argv(), it.lds, s);
    if (e != hipSuccess) return hip_rc(e);
  }
  return HX_OK;
}

extern "C" int hx_plan_destroy(hx_plan* plan) {
  if (!plan) return HX_OK;
  if (t_recording == &plan->rec) t_recording = nullptr;
  delete plan;
  return HX_OK;
}

namespace {
__global__ __launch_bounds__(256) void zero_kernel(uint32_t* __restrict__ p, int64_t n_words) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n_words) p[i] = 0u;
}
}  // namespace

// memset(p, 0, bytes) on the stream as a KERNEL of this library: recordable in a plan (a torch.zeros inside a
// recorded region would run once, at recording time, and never again) and a plain kernel node under hipGraph
// capture (a captured hipMemsetAsync node was seen to leave garbage from the second replay on).  4-byte granularity.
// hx_measure_read_stream: the read rate of this GPU for the access shape of the weight-streaming kernels (1 KiB
// contiguous per wave instruction, non-temporal), nothing computed.  Eight loads in flight per wave, consumed before the
// next eight are requested, 512 workgroups: the best of the shapes of tools/bench_stream.py / tools/probes/stream_lds_dma.py
// (6.8 - 7.0 TB/s; sixteen in flight from 1024 workgroups, the round's first version of this kernel: 6.3 - 6.5; without
// the non-temporal hint 5.7 - 6.3; the same stream by LDS-DMA: no different).
namespace {
__global__ __launch_bounds__(256) void read_stream_kernel(const char* __restrict__ base, int64_t n_chunks, float* sink) {
  const int lane = threadIdx.x & 63;
  const int64_t wave = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6), n_waves = (int64_t)gridDim.x * 4;
  hx::f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  for (int64_t c = wave; c < n_chunks; c += n_waves) {
    const char* p0 = base + c * 8192 + lane * 16;
    hx::f32x4 v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = __builtin_nontemporal_load(reinterpret_cast<const hx::f32x4*>(p0 + u * 1024));
#pragma unroll
    for (int u = 0; u < 8; ++u) acc += v[u];
  }
  if (acc[0] + acc[1] + acc[2] + acc[3] == 123.456f) sink[0] = acc[0];
}
}  // namespace

extern "C" int hx_measure_read_stream(const void* p, int64_t bytes, float* sink, hx_stream stream) {
  if (!p || !sink) return HX_ERR_NULL;
  if (bytes <= 0 || bytes % 8192) return HX_ERR_SHAPE;
  if (reinterpret_cast<uintptr_t>(p) & 15u) return HX_ERR_STRIDE;
  hx::launcher(read_stream_kernel, 512, 256, 0, (hipStream_t)stream)((const char*)p, bytes / 8192, sink);
  return check_launch();
}

extern "C" int hx_measure_read_grid(const void* p, int64_t bytes, int n_workgroups, float* sink, hx_stream stream) {
  if (!p || !sink) return HX_ERR_NULL;
  if (bytes <= 0 || bytes % 8192 || n_workgroups <= 0 || n_workgroups > 65535) return HX_ERR_SHAPE;
  if (reinterpret_cast<uintptr_t>(p) & 15u) return HX_ERR_STRIDE;
  hx::launcher(read_stream_kernel, (unsigned)n_workgroups, 256, 0, (hipStream_t)stream)((const char*)p, bytes / 8192, sink);
  return check_launch();
}

// The decode attention kernel's access pattern with the arithmetic removed (the round-4 probe of
// tools/bench_attn_ceiling.py, moved into the default library for the null layer): grid (head, sequence), 4 waves,
// wave w owns tiles w, w + 4, ...; one wave instruction = 4 key rows x 256 B; a tile = 4 K + 4 V instructions; register
// double buffer (tile t + 4 requested before tile t is consumed) like attn_decode_kernel.
namespace {
typedef __amdgpu_buffer_rsrc_t prsrc_t;
typedef unsigned int pu32x4 __attribute__((__vector_size__(16)));
__global__ __launch_bounds__(256) void paged_read_kernel(const char* __restrict__ kbase, const char* __restrict__ vbase,
                                                         const int32_t* __restrict__ table, int64_t table_stride, int tiles,
                                                         int64_t page_bytes, int64_t row_bytes, float* sink) {
  // (round 5: like attn_decode_kernel, every tile load is ISSUED — a tile past the end through a buffer resource with
  // zero records, which touches no memory — so that the loop has no branch around its loads and two tiles per wave are
  // really in flight; the probe must not be held back by something the kernel it stands in for no longer does)
  const int h = blockIdx.x, b = blockIdx.y;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const uint32_t lane_off = (uint32_t)((lane >> 4) * row_bytes + h * 256 + (lane & 15) * 16);
  const int32_t* tb = table + (int64_t)b * table_stride;
  const int n_my = (tiles - w + 3) / 4;                 // tiles w, w + 4, ...
  const int tj = w + 4 * lane;
  const int my_page = (lane < n_my && tj < tiles) ? tb[tj] : 0;     // one coalesced load of the wave's page ids (<= 64 tiles per wave)
  pu32x4 acc = {0u, 0u, 0u, 0u};
  pu32x4 bufA[8], bufB[8];
  auto issue = [&](pu32x4 (&bf)[8], int jj) {
    const int page = __builtin_amdgcn_readlane(my_page, min(jj, 63));
    const uint32_t nrec = jj < n_my ? 0x7fffffffu : 0u;
    const prsrc_t kr = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(kbase + (int64_t)page * page_bytes), 0, nrec, 0x00020000);
    const prsrc_t vr = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(vbase + (int64_t)page * page_bytes), 0, nrec, 0x00020000);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      bf[i] = __builtin_amdgcn_raw_buffer_load_b128(kr, lane_off + (uint32_t)(i * 4 * row_bytes), 0, 2);
      bf[4 + i] = __builtin_amdgcn_raw_buffer_load_b128(vr, lane_off + (uint32_t)(i * 4 * row_bytes), 0, 2);
    }
  };
  issue(bufA, 0);
  for (int j = 0; j < n_my; j += 2) {
    issue(bufB, j + 1);
#pragma unroll
    for (int i = 0; i < 8; ++i) acc += bufA[i];
    issue(bufA, j + 2);
    if (j + 1 < n_my) {
#pragma unroll
      for (int i = 0; i < 8; ++i) acc += bufB[i];
    }
  }
  if (acc[0] + acc[1] + acc[2] + acc[3] == 0x12345678u) sink[0] = 1.f;
}
}  // namespace

extern "C" int hx_measure_paged_read(const void* kbase, const void* vbase, const int32_t* table, int64_t table_stride,
                                     int n_seq, int n_heads, int tiles, int64_t page_bytes, int64_t row_bytes,
                                     int head_bytes, float* sink, hx_stream stream) {
  if (!kbase || !vbase || !table || !sink) return HX_ERR_NULL;
  if (n_seq <= 0 || n_heads <= 0 || tiles <= 0 || tiles > 256 || table_stride < tiles || head_bytes != 256 ||
      row_bytes < (int64_t)n_heads * 256 || page_bytes < 16 * row_bytes || page_bytes > 0x7fffffff || n_seq > 65535)
    return HX_ERR_SHAPE;
  if ((reinterpret_cast<uintptr_t>(kbase) | reinterpret_cast<uintptr_t>(vbase) | (uintptr_t)row_bytes | (uintptr_t)page_bytes) & 15u) return HX_ERR_STRIDE;
  hx::launcher(paged_read_kernel, dim3((unsigned)n_heads, (unsigned)n_seq), 256, 0, (hipStream_t)stream)(
      (const char*)kbase, (const char*)vbase, table, table_stride, tiles, page_bytes, row_bytes, sink);
  return check_launch();
}

extern "C" int hx_memset_zero(void* p, int64_t bytes, hx_stream stream) {
  if (!p || bytes < 0) return HX_ERR_NULL;
  if (bytes == 0) return HX_OK;
  if ((bytes & 3) || (reinterpret_cast<uintptr_t>(p) & 3u)) return HX_ERR_STRIDE;
  const int64_t n = bytes >> 2;
  hx::launcher(zero_kernel, (unsigned)((n + 255) / 256), 256, 0, (hipStream_t)stream)((uint32_t*)p, n);
  return check_launch();
}
