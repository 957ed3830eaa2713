// hx_common.h — shared device/host helpers for libhydra_hip (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdint.h>
#include <memory>
#include <tuple>
#include <utility>
#include "../../include/hydra_hip.h"
#include "../../include/hydra_hip_experimental.h"   // declarations only; defined in EXPERIMENTS=1 builds
#ifndef HX_EXPERIMENTS
#define HX_EXPERIMENTS 0
#endif

#define HX_WAVE 64

namespace hx {

// thread-local last hipError_t for hx_last_hip_error()
int& last_hip_error();

inline int check_launch() {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    last_hip_error() = (int)e;
    return HX_ERR_HIP;
  }
  return HX_OK;
}

inline int hip_rc(hipError_t e) {
  if (e != hipSuccess) {
    last_hip_error() = (int)e;
    return HX_ERR_HIP;
  }
  return HX_OK;
}

// ---------------------------------------------------------------------------------------------------------
// Kernel launches.  Every launch of the library goes through hx::launcher(kernel, grid, block, lds, stream)(args...)
// instead of kernel<<<...>>>(args...): outside a recording it is the same hipLaunchKernel call; while a launch plan is
// being recorded on this thread (hx_plan_begin .. hx_plan_end, csrc/launch_plan.hip) the launch is appended to the
// plan — kernel, geometry and a private copy of its arguments — and NOT executed; hx_plan_launch replays the list.
//
// Why the library has its own replay mechanism next to hipGraph: a captured hipGraph always puts the AQL barrier
// bit between consecutive kernels, so a kernel's workgroups are dispatched only after its predecessor has drained
// completely (~2 us of idle HBM plus the ramp, five times per decoder layer).  A plan can launch a kernel with
// hipExtAnyOrderLaunch (barrier bit cleared: the command processor dispatches its workgroups, in queue order, as
// soon as the predecessor's LAST workgroup has been dispatched, i.e. while that one is still running); the data
// dependency is then taken inside the kernel (ChainLink below): weight / KV prefetch first, then wait for the
// predecessor's done flag, then consume its output.  Measured: tools/probes/probe_chain2.hip, DESIGN.md §6d.
// ---------------------------------------------------------------------------------------------------------
struct ChainLink {
  const uint32_t* wait;    // sync area of the predecessor launch (nullptr: nothing to wait for)
  uint32_t* signal;        // sync area of this launch (nullptr: nobody waits for it)
  uint32_t* err;           // word set to 1 if a wait gives up (1 s): the step's results are invalid
  uint32_t signal_total;   // workgroups of this launch
  uint32_t opts;           // diagnostics (hx_debug_set_option("chain_stamps", v)): bit 0 time stamps of workgroup 0 and of the
                           // flag raiser into the link area; bit 1 earliest / latest workgroup entry (one atomic pair per workgroup)
};
// sync area of one chained launch, in 32-word (128-byte) lines: 16 arrival-count shards (workgroup id mod 16), one
// line counting completed shards, 8 flag lines (one per XCD, polled by that XCD's waiters)
constexpr int kChainShards = 16;
constexpr int kChainTopWord = 32 * kChainShards;
constexpr int kChainFlagWord = kChainTopWord + 32;
constexpr int kChainStampWord = kChainFlagWord + 8 * 32;   // 4 x uint64 (100 MHz clock): workgroup 0 at its wait, workgroup
                                                           // 0 past its wait, the flag raiser at the end, workgroup 0 at its signal
constexpr int kChainWords = 1024;

struct ArgHolderBase {
  virtual ~ArgHolderBase() = default;
  virtual void** argv() = 0;
};
template <typename... P>
struct ArgHolder final : ArgHolderBase {
  std::tuple<P...> t;
  void* ptrs[sizeof...(P) ? sizeof...(P) : 1];
  explicit ArgHolder(const P&... a) : t(a...) { fill(std::index_sequence_for<P...>{}); }
  template <size_t... I> void fill(std::index_sequence<I...>) { ((ptrs[I] = (void*)&std::get<I>(t)), ...); }
  void** argv() override { return ptrs; }
};

struct PlanRecorder;
PlanRecorder* recording();       // the plan being recorded on this thread, or nullptr
void record_launch(PlanRecorder* r, const void* func, dim3 grid, dim3 block, size_t lds, uint32_t flags,
                   std::unique_ptr<ArgHolderBase> args, bool chained);
// Next link of the launch chain being recorded (zeros outside a chained recording).  *flags gets
// hipExtAnyOrderLaunch when the launch has a predecessor to wait for.
ChainLink chain_next(uint32_t n_workgroups, uint32_t* flags);

template <typename... P>
struct Launcher {
  void (*kernel)(P...);
  dim3 grid, block;
  size_t lds;
  hipStream_t stream;
  uint32_t flags;
  bool chained;
  void operator()(P... a) const {
    if (PlanRecorder* r = recording()) {
      record_launch(r, (const void*)kernel, grid, block, lds, flags,
                    std::unique_ptr<ArgHolderBase>(new ArgHolder<P...>(a...)), chained);
      return;
    }
    void* argv[sizeof...(P) ? sizeof...(P) : 1] = {(void*)&a...};
    (void)hipLaunchKernel((const void*)kernel, grid, block, argv, lds, stream);   // errors: hipGetLastError (check_launch)
  }
};
// a launch that takes no part in a chain (it ends one: its successor is launched in stream order again)
template <typename... P>
inline Launcher<P...> launcher(void (*kernel)(P...), dim3 grid, dim3 block, size_t lds, hipStream_t stream) {
  return Launcher<P...>{kernel, grid, block, lds, stream, 0u, false};
}
// a launch whose kernel implements the ChainLink protocol (flags from chain_next)
template <typename... P>
inline Launcher<P...> launcher_chained(void (*kernel)(P...), dim3 grid, dim3 block, size_t lds, hipStream_t stream,
                                       uint32_t flags) {
  return Launcher<P...>{kernel, grid, block, lds, stream, flags, true};
}

inline int64_t dtype_size(int dtype) {
  switch (dtype) {
    case HX_F32: return 4;
    case HX_F16: return 2;
    case HX_BF16: return 2;
    default: return 0;
  }
}

inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

// ---------------------------------------------------------------------------
// 16-bit float element traits.  Storage is always raw uint16_t bits; arithmetic
// in "T precision" means: compute in fp32, round to T after every operation
// (exact emulation of IEEE T arithmetic for + - * since 24 >= 2*p+2).
// ---------------------------------------------------------------------------
typedef uint16_t u16;
typedef u16 u16x4 __attribute__((ext_vector_type(4)));
typedef u16 u16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));

struct F16 {
  static constexpr int kDtype = HX_F16;
  typedef u16 storage;
  static __device__ __forceinline__ float to_float(u16 b) {
    return (float)__builtin_bit_cast(_Float16, b);
  }
  static __device__ __forceinline__ u16 from_float(float f) {
    return __builtin_bit_cast(u16, (_Float16)f);  // RNE
  }
};

struct BF16 {
  static constexpr int kDtype = HX_BF16;
  typedef u16 storage;
  static __device__ __forceinline__ float to_float(u16 b) {
    return __builtin_bit_cast(float, ((uint32_t)b) << 16);
  }
  static __device__ __forceinline__ u16 from_float(float f) {
    return __builtin_bit_cast(u16, (__bf16)f);  // v_cvt_pk_bf16_f32: RNE, NaN-preserving
  }
};

struct F32 {
  static constexpr int kDtype = HX_F32;
  typedef float storage;
  static __device__ __forceinline__ float to_float(float b) { return b; }
  static __device__ __forceinline__ float from_float(float f) { return f; }
};

// round fp32 value to T precision and come back (one "T arithmetic" rounding)
template <typename T>
__device__ __forceinline__ float round_to(float f) {
  return T::to_float(T::from_float(f));
}
template <>
__device__ __forceinline__ float round_to<F32>(float f) { return f; }

// wave-level reductions (64 lanes)
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v = fmaxf(v, __shfl_xor(v, off, 64));
  return v;
}

// ---------------------------------------------------------------------------------------------------------
// Device side of a launch chain (see ChainLink).  Protocol of a chained kernel:
//   1. issue every load that does not depend on the predecessor (weights, KV pages, block tables);
//   2. chain_wait(link)  — all threads; the first wave polls this XCD's flag line of the predecessor's area;
//   3. read the predecessor's output with chain_load* (sc1: served past the CU's L1 and coherent across the XCDs'
//      L2s), write its own output with chain_store* (sc1: written through);
//   4. chain_signal(link) — all threads: drain the stores, then thread 0 counts the workgroup in on its shard of the
//      arrival counter; the workgroup that completes the last shard raises the eight flag lines.
// Progress: a chained kernel's workgroups are dispatched only after ALL of its predecessor's have been (one
// in-order queue), so a waiter never holds a resource its producer still needs.  A wait is bounded (1 s of the
// 100 MHz clock): on give-up the error word is set and the kernel continues — never a hung GPU; the host
// checks the word with the step's tokens.
// ---------------------------------------------------------------------------------------------------------
typedef __amdgpu_buffer_rsrc_t chain_rsrc_t;
__device__ __forceinline__ chain_rsrc_t chain_rsrc(const void* p) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, 0x7fffffff, 0x00020000);
}
constexpr int kSc1 = 16;   // aux bit of the buffer intrinsics on gfx94x/gfx950: sc1 (agent-coherent, write-through)
__device__ __forceinline__ u32x4 chain_load_b128(chain_rsrc_t r, uint32_t byte_off) {
  return __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(r, byte_off, 0, kSc1));
}
__device__ __forceinline__ void chain_store_b128(chain_rsrc_t r, uint32_t byte_off, u32x4 v) {
  typedef unsigned int bu32x4_ __attribute__((__vector_size__(16)));
  __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(bu32x4_, v), r, byte_off, 0, kSc1);
}
__device__ __forceinline__ void chain_store_b64(chain_rsrc_t r, uint32_t byte_off, u32x2 v) {
  typedef unsigned int bu32x2_ __attribute__((__vector_size__(8)));
  __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(bu32x2_, v), r, byte_off, 0, kSc1);
}
__device__ __forceinline__ void chain_store_b16(chain_rsrc_t r, uint32_t byte_off, u16 v) {
  __builtin_amdgcn_raw_buffer_store_b16((short)v, r, byte_off, 0, kSc1);
}
__device__ __forceinline__ float chain_load_f32(chain_rsrc_t r, uint32_t byte_off) {
  return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, byte_off, 0, kSc1));
}
__device__ __forceinline__ int xcc_id() { return __builtin_amdgcn_s_getreg(20 | (3 << 11)) & 7; }   // HW_REG_XCC_ID

__device__ __forceinline__ void chain_stamp(const ChainLink& ch, int which, bool only_wg0) {
  if ((ch.opts & 1u) && ch.signal && threadIdx.x == 0 &&
      (!only_wg0 || (blockIdx.x | blockIdx.y | blockIdx.z) == 0))
    reinterpret_cast<unsigned long long*>(ch.signal + kChainStampWord)[which] = __builtin_amdgcn_s_memrealtime();
}

// First thing in a chained kernel (all threads): request the predecessor's flag BEFORE any prefetch load is issued.
// Loads return in order, so this one comes back after one round trip (~1 us) however many prefetch loads queue up
// behind it; in the common case — the predecessor had already finished when this workgroup was dispatched — the
// later chain_wait then costs nothing (measured: polled only after the prefetch had been issued, the first flag
// load came back behind 32 KiB of HBM loads, 2.2 us per launch on the critical path).
__device__ __forceinline__ uint32_t chain_peek(const ChainLink& ch) {
  if ((ch.opts & 2u) && ch.signal && threadIdx.x == 0) {   // diagnostic: first / last workgroup entry of this launch
    unsigned long long* st = reinterpret_cast<unsigned long long*>(ch.signal + kChainStampWord);
    const unsigned long long t = __builtin_amdgcn_s_memrealtime();
    atomicMax(st + 4, ~t);      // earliest entry, complemented (the area starts zeroed)
    atomicMax(st + 5, t);       // latest entry
  }
  uint32_t v = 1u;
  if (ch.wait && threadIdx.x < 64)
    v = __hip_atomic_load(ch.wait + kChainFlagWord + 32 * xcc_id(), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  return v;
}

__device__ __forceinline__ void chain_wait(const ChainLink& ch, uint32_t peeked = 0u) {
  chain_stamp(ch, 0, true);
  if (ch.wait) {
    if (threadIdx.x < 64 && !__builtin_amdgcn_readfirstlane(peeked)) {
      const uint32_t* fl = ch.wait + kChainFlagWord + 32 * xcc_id();
      const uint64_t t0 = __builtin_amdgcn_s_memrealtime();
      while (!__builtin_amdgcn_readfirstlane(__hip_atomic_load(fl, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))) {
        __builtin_amdgcn_s_sleep(2);
        if (__builtin_amdgcn_s_memrealtime() - t0 > 100000000ull) {   // 1 s: report, never hang
          if (threadIdx.x == 0) __hip_atomic_fetch_or(ch.err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          break;
        }
      }
    }
    __syncthreads();
  }
  chain_stamp(ch, 1, true);
  asm volatile("" ::: "memory");
}

__device__ __forceinline__ void chain_signal(const ChainLink& ch) {
  if ((ch.opts & 2u) && ch.signal && threadIdx.x == 0) {   // diagnostic: histogram of workgroup END times, 2 us buckets
    // relative to the launch's earliest workgroup entry (words kChainStampWord + 16 .. + 79)
    const unsigned long long t = __builtin_amdgcn_s_memrealtime();
    const unsigned long long first = ~__hip_atomic_load(reinterpret_cast<unsigned long long*>(ch.signal + kChainStampWord) + 4,
                                                         __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    unsigned long long b = (t - first) / 200ull;
    if (b > 63ull) b = 63ull;
    atomicAdd(ch.signal + kChainStampWord + 16 + (int)b, 1u);
    // ... and the summed end time (us) per blockIdx.x and per blockIdx.y (both mod 32): who finishes late?
    atomicAdd(ch.signal + kChainStampWord + 80 + (blockIdx.x & 31), (uint32_t)((t - first) / 100ull));
    atomicAdd(ch.signal + kChainStampWord + 112 + (blockIdx.y & 31), (uint32_t)((t - first) / 100ull));
  }
  if (ch.signal) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this thread's write-through stores have reached memory
    __syncthreads();
    chain_stamp(ch, 3, true);
    if (threadIdx.x == 0) {
      // ONE memory-side operation per workgroup (1024 workgroups reading eight count lines each cost the decode
      // attention launch 10 us): the arrival counter is sharded by workgroup id, so every shard knows its own total;
      // whoever completes a shard counts it in on the top line, whoever completes that raises the eight flags
      const uint32_t id = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);
      const uint32_t shard = id & (kChainShards - 1);
      const uint32_t expect = (ch.signal_total + (kChainShards - 1) - shard) / kChainShards;
      if (__hip_atomic_fetch_add(ch.signal + 32 * shard, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1 == expect) {
        const uint32_t n_shards = ch.signal_total < (uint32_t)kChainShards ? ch.signal_total : (uint32_t)kChainShards;
        if (__hip_atomic_fetch_add(ch.signal + kChainTopWord, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1 == n_shards) {
#pragma unroll
          for (int i = 0; i < 8; ++i)
            __hip_atomic_store(ch.signal + kChainFlagWord + 32 * i, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          chain_stamp(ch, 2, false);
        }
      }
    }
  }
}

}  // namespace hx
