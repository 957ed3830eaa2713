// launch_plan.hip — record / replay of a fixed launch sequence (the decode step), with launch chains.
//
// The role a hipGraph plays in the reference's unfinished cuda_graph_model_runner.py
// (hydrainfer/model_runner/cuda_graph_model_runner.py:1-72) and in engine/graph_decode.py: a decode step is
// ~170 launches of 5-70 us kernels whose arguments never change, so they are issued by one native loop instead of
// one Python call each.  What a plan can do and a captured hipGraph cannot is to launch a kernel WITHOUT the AQL
// barrier bit (hipExtAnyOrderLaunch): its workgroups are dispatched, in queue order, while the previous kernel is
// still draining; the data dependency is taken inside the kernel (hx::ChainLink, hx_common.h).  hipGraph capture
// drops that flag (tools/probes/probe_chain.hip: captured any-order launches start 1.8-3.7 us after their
// predecessor's end like ordinary ones; launched eagerly they start 0.3-37 us BEFORE it).
//
// A plan is recorded on one thread: between hx_plan_begin and hx_plan_end every hx_* entry point called on that
// thread appends its launches (hx::launcher) to the plan instead of executing them.  Chained recording
// (chain != 0): consecutive launches of chain-capable kernels are linked — launch i+1 waits, in the kernel, for
// launch i's done flag and is launched any-order; any other launch ends the chain (it is launched in stream order,
// i.e. after everything before it has completed) and the next capable launch starts a new one.  The sync areas of
// the links live in a caller-provided device buffer that the plan zeroes at the start of every replay.
#include <cstring>
#include <vector>
#include "hx_common.h"

namespace hx {

struct PlanItem {
  int kind;                 // 0 kernel, 1 memset
  const void* func;
  dim3 grid, block;
  size_t lds;
  uint32_t flags;
  bool chained;
  std::unique_ptr<ArgHolderBase> args;
  void* ptr;                // memset
  size_t bytes;
};

struct PlanRecorder {
  std::vector<PlanItem> items;
  bool chain = false;
  uint32_t* sync = nullptr;      // device buffer of link areas (kChainWords each)
  uint32_t* err = nullptr;       // device word: set to 1 by a wait that gives up
  int64_t sync_words = 0, used_words = 0;
  uint32_t* prev_signal = nullptr;
  int n_chained = 0;
  bool overflow = false;
};

static thread_local PlanRecorder* t_recording = nullptr;
static int g_chain_stamps = 0;   // diagnostic (hx_debug_set_option("chain_stamps", 1)): links recorded from now on carry time stamps

int plan_set_option(const char* name, int value) {
  if (!strcmp(name, "chain_stamps")) { g_chain_stamps = value & 3; return HX_OK; }
  return HX_ERR_UNSUPPORTED;
}

PlanRecorder* recording() { return t_recording; }

void record_launch(PlanRecorder* r, const void* func, dim3 grid, dim3 block, size_t lds, uint32_t flags,
                   std::unique_ptr<ArgHolderBase> args, bool chained) {
  PlanItem it;
  it.kind = 0; it.func = func; it.grid = grid; it.block = block; it.lds = lds; it.flags = flags; it.chained = chained;
  it.args = std::move(args); it.ptr = nullptr; it.bytes = 0;
  if (!chained) r->prev_signal = nullptr;      // a launch outside the protocol ends the chain
  else if (flags) ++r->n_chained;
  r->items.push_back(std::move(it));
}

ChainLink chain_next(uint32_t n_workgroups, uint32_t* flags) {
  ChainLink lk{nullptr, nullptr, nullptr, 0u, 0u};
  *flags = 0;
  PlanRecorder* r = t_recording;
  if (!r || !r->chain) return lk;
  if (r->used_words + kChainWords > r->sync_words) {   // out of link areas: this launch runs unchained
    r->overflow = true;
    r->prev_signal = nullptr;
    return lk;
  }
  lk.wait = r->prev_signal;
  lk.signal = r->sync + r->used_words;
  lk.err = r->err;
  lk.signal_total = n_workgroups;
  lk.opts = (uint32_t)g_chain_stamps;
  r->used_words += kChainWords;
  r->prev_signal = lk.signal;
  if (lk.wait) *flags = hipExtAnyOrderLaunch;
  return lk;
}

}  // namespace hx

using namespace hx;

struct hx_plan {
  PlanRecorder rec;
  bool finished = false;
};

extern "C" int hx_plan_begin(hx_plan** plan, void* sync, int64_t sync_bytes, uint32_t* error_word, int chain) {
  if (!plan) return HX_ERR_NULL;
  if (t_recording) return HX_ERR_UNSUPPORTED;          // one recording per thread
  if (chain && (!sync || !error_word || sync_bytes < (int64_t)kChainWords * 4 || (reinterpret_cast<uintptr_t>(sync) & 127u)))
    return HX_ERR_WORKSPACE;
  hx_plan* p = new hx_plan();
  p->rec.chain = chain != 0;
  p->rec.sync = reinterpret_cast<uint32_t*>(sync);
  p->rec.err = error_word;
  p->rec.sync_words = chain ? sync_bytes / 4 : 0;
  *plan = p;
  t_recording = &p->rec;
  return HX_OK;
}

extern "C" int hx_plan_end(hx_plan* plan) {
  if (!plan) return HX_ERR_NULL;
  if (t_recording != &plan->rec) return HX_ERR_UNSUPPORTED;
  t_recording = nullptr;
  plan->finished = true;
  return plan->rec.overflow ? HX_ERR_WORKSPACE : HX_OK;
}

extern "C" int hx_plan_info(const hx_plan* plan, int32_t* n_launches, int32_t* n_any_order, int64_t* sync_bytes_used) {
  if (!plan) return HX_ERR_NULL;
  if (n_launches) *n_launches = (int32_t)plan->rec.items.size();
  if (n_any_order) *n_any_order = plan->rec.n_chained;
  if (sync_bytes_used) *sync_bytes_used = plan->rec.chain ? plan->rec.used_words * 4 : 0;
  return HX_OK;
}

extern "C" int hx_plan_launch(const hx_plan* plan, hx_stream stream) {
  if (!plan) return HX_ERR_NULL;
  if (!plan->finished || t_recording) return HX_ERR_UNSUPPORTED;
  hipStream_t s = (hipStream_t)stream;
  const PlanRecorder& r = plan->rec;
  if (r.chain && r.used_words > 0) {
    // flags and counts of the previous replay -> 0; stream-ordered: every earlier launch has completed
    hipError_t e = hipMemsetAsync(r.sync, 0, (size_t)r.used_words * 4, s);
    if (e != hipSuccess) return hip_rc(e);
  }
  for (const PlanItem& it : r.items) {
    hipError_t e;
    if (it.kind == 1) e = hipMemsetAsync(it.ptr, 0, it.bytes, s);
    else if (it.flags) e = hipExtLaunchKernel(it.func, it.grid, it.block, it.args->argv(), it.lds, s, nullptr, nullptr, (int)it.flags);
    else e = hipLaunchKernel(it.func, it.grid, it.block, it.args->argv(), it.lds, s);
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
extern "C" int hx_memset_zero(void* p, int64_t bytes, hx_stream stream) {
  if (!p || bytes < 0) return HX_ERR_NULL;
  if (bytes == 0) return HX_OK;
  if ((bytes & 3) || (reinterpret_cast<uintptr_t>(p) & 3u)) return HX_ERR_STRIDE;
  const int64_t n = bytes >> 2;
  hx::launcher(zero_kernel, (unsigned)((n + 255) / 256), 256, 0, (hipStream_t)stream)((uint32_t*)p, n);
  return check_launch();
}

static_assert(HX_PLAN_SYNC_BYTES_PER_LAUNCH == hx::kChainWords * 4, "header and kernel disagree on the size of a link area");
static_assert(hx::kChainFlagWord + 8 * 32 <= hx::kChainWords, "link area too small for its lines");
