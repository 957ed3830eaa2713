// kernels of tools/probes/fence_cost.cpp (compiled with hipcc --genco to a code object that the probe loads through HSA)
#include <hip/hip_runtime.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
extern "C" __global__ __launch_bounds__(256) void k_empty(const f32x4* __restrict__ src, float* __restrict__ dst, long n_chunks, long n_waves) {}
// reads n_chunks x 8 KiB the way the weight-streaming kernels do (one 8 KiB chunk per wave in flight), writes 16 bytes per
// thread of the first `n_write` threads (dirty lines for the release at the kernel's end to write back)
extern "C" __global__ __launch_bounds__(256) void k_stream(const f32x4* __restrict__ src, float* __restrict__ dst, long n_chunks, long n_waves) {
  // (n_waves is an argument: gridDim would come from the hidden kernel arguments, which only the HIP runtime fills in)
  const int lane = threadIdx.x & 63;
  const long wave = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  for (long c = wave; c < n_chunks; c += n_waves) {
    const f32x4* p0 = src + c * 512 + lane;
    f32x4 v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = __builtin_nontemporal_load(p0 + u * 64);
#pragma unroll
    for (int u = 0; u < 8; ++u) acc += v[u];
  }
  // 256 workgroups x 256 threads x 16 B = 1 MiB of results (a split-K slab's worth)
  reinterpret_cast<f32x4*>(dst)[(long)blockIdx.x * 256 + threadIdx.x] = acc;
}
