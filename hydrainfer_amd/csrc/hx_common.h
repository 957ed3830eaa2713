// hx_common.h — shared device/host helpers for libhydra_hip (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
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
// ---------------------------------------------------------------------------------------------------------
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
void record_launch(PlanRecorder* r, const void* func, dim3 grid, dim3 block, size_t lds,
                   std::unique_ptr<ArgHolderBase> args);

template <typename... P>
struct Launcher {
  void (*kernel)(P...);
  dim3 grid, block;
  size_t lds;
  hipStream_t stream;
  void operator()(P... a) const {
    if (PlanRecorder* r = recording()) {
      record_launch(r, (const void*)kernel, grid, block, lds, std::unique_ptr<ArgHolderBase>(new ArgHolder<P...>(a...)));
      return;
    }
    void* argv[sizeof...(P) ? sizeof...(P) : 1] = {(void*)&a...};
    (void)hipLaunchKernel((const void*)kernel, grid, block, argv, lds, stream);   // errors: hipGetLastError (check_launch)
  }
};
template <typename... P>
inline Launcher<P...> launcher(void (*kernel)(P...), dim3 grid, dim3 block, size_t lds, hipStream_t stream) {
  return Launcher<P...>{kernel, grid, block, lds, stream};
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


// One decode step's metadata advance for `batch` sequences by ONE 256-thread workgroup (hx_decode_advance, and the
// advance role of hx_decode_step_head): positions / kv lengths += stride, cu_seqlens_k = their prefix sum, the new
// token's cache slot through the block table (hydrainfer/layer/causal_attention.py:147-168).  scan: 256 ints of LDS.
__device__ __forceinline__ void decode_advance_block(int32_t* __restrict__ positions, int32_t* __restrict__ kv_lens,
                                                     int32_t* __restrict__ cu_seqlens_k, int32_t* __restrict__ new_cache_slots,
                                                     const int32_t* __restrict__ block_table,
                                                     const int32_t* __restrict__ cu_block_lens, int32_t batch,
                                                     int32_t block_size, int32_t stride, int32_t* scan) {
  int32_t carry = 0;
  if (threadIdx.x == 0) cu_seqlens_k[0] = 0;
  for (int base = 0; base < batch; base += 256) {
    const int b = base + threadIdx.x;
    int32_t len = 0;
    if (b < batch) {
      const int32_t pos = positions[b] + stride;
      positions[b] = pos;
      len = kv_lens[b] + stride;
      kv_lens[b] = len;
      const int32_t page = block_table[cu_block_lens[b] + pos / block_size];
      new_cache_slots[b] = page * block_size + pos % block_size;
    }
    scan[threadIdx.x] = len;
    __syncthreads();
    // inclusive Hillis-Steele scan over 256 entries
    for (int off = 1; off < 256; off <<= 1) {
      int32_t add = (threadIdx.x >= off) ? scan[threadIdx.x - off] : 0;
      __syncthreads();
      scan[threadIdx.x] += add;
      __syncthreads();
    }
    if (b < batch) cu_seqlens_k[b + 1] = carry + scan[threadIdx.x];
    carry += scan[255];
    __syncthreads();
  }
}

// The RANK DESCRIPTOR of a decode batch (attn_decode.hip, RANKED form): rank_desc[0] = 1 when the batch is ragged — some
// sequence holds more than 1.125 x the mean + 16 keys — else 0; rank_desc[1 + r] = the sequence with the r-th most keys
// (ties: the lower sequence number first).  One 256-thread workgroup, `batch` <= 256 (a larger batch is declared even).
// lens: this workgroup's LDS scratch of 256 ints, filled here from kv_lens (global, already current).
#define HX_RANKED_THR 1.125f
__device__ __forceinline__ void decode_rank_block(const int32_t* __restrict__ kv_lens, int32_t batch, int32_t* __restrict__ rank_desc,
                                                  int32_t* lens) {
  if (batch > 256) {
    if (threadIdx.x == 0) rank_desc[0] = 0;
    return;
  }
  const int b = threadIdx.x;
  const int32_t mine = b < batch ? kv_lens[b] : 0;
  __syncthreads();      // (lens may be the scratch of a scan that has just finished)
  lens[b] = mine;
  __syncthreads();
  int64_t total = 0;
  int rank = 0;
  for (int i = 0; i < batch; ++i) {
    const int32_t ln = lens[i];
    total += ln;
    rank += (ln > mine || (ln == mine && i < b)) ? 1 : 0;
  }
  const float mean = (float)total / (float)batch;
  const int ragged = __syncthreads_or(b < batch && (float)mine > mean * HX_RANKED_THR + 16.f);
  if (b < batch) rank_desc[1 + rank] = b;
  if (b == 0) rank_desc[0] = ragged ? 1 : 0;
}

}  // namespace hx
