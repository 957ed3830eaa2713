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

extern "C" int hx_memset_zero(void* p, int64_t bytes, hx_stream stream) {
  if (!p || bytes < 0) return HX_ERR_NULL;
  if (bytes == 0) return HX_OK;
  if ((bytes & 3) || (reinterpret_cast<uintptr_t>(p) & 3u)) return HX_ERR_STRIDE;
  const int64_t n = bytes >> 2;
  hx::launcher(zero_kernel, (unsigned)((n + 255) / 256), 256, 0, (hipStream_t)stream)((uint32_t*)p, n);
  return check_launch();
}
