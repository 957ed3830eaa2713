// Probe 3: a decode-layer-shaped chain of weight-streaming kernels ([396, 33, 180, 90, 100] MB x layers), each
// consuming a small result of its predecessor.  (a) hipGraph in stream order (today's decode step),
// (b) eager stream order, (c) eager hipExtAnyOrderLaunch (AQL barrier bit cleared: the next kernel's workgroups are
// dispatched while the previous one drains) + first loads issued BEFORE an in-kernel acquire-wait on the
// predecessor's per-XCD done counters (release increments).  Reports time per kernel, host time per launch and
// whether every consumer saw its predecessor's result (plain stores + agent-scope release/acquire).
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <chrono>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

typedef float f4 __attribute__((ext_vector_type(4)));
constexpr int U = 8;
constexpr int RES_FLOATS = 128;      // result floats per workgroup (512 B)

struct Sync { unsigned int done[8][32]; unsigned int bad, timeout, pad[30]; };   // one 128-B line per XCD counter

__global__ __launch_bounds__(256) void stream_k(Sync* sync, int idx, const char* __restrict__ w, long bytes,
                                                float* res_prev, float* res_mine, int wait_prev, unsigned int prev_grid,
                                                unsigned long long* stamps, int rel, int acq) {
    const int lane = threadIdx.x & 63;
    const int wave = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int n_waves = gridDim.x * 4;
    const long chunk = U * 1024;
    const long n_chunks = bytes / chunk;
    unsigned long long t0 = wall_clock64();
    f4 v[U];
    long c = wave;
    if (c < n_chunks) {
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = __builtin_nontemporal_load((const f4*)(w + c * chunk + u * 1024 + lane * 16));
    }
    if (wait_prev && idx > 0) {
        if (threadIdx.x < 64) {
            const Sync* ps = sync + (idx - 1);
            for (;;) {
                unsigned int x = lane < 8 ? __hip_atomic_load(&ps->done[lane][0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0u;
                for (int o = 4; o; o >>= 1) x += __shfl_xor(x, o);
                if (__builtin_amdgcn_readfirstlane(x) >= prev_grid) break;
                if (wall_clock64() - t0 > 1000000ull) { if (lane == 0) atomicAdd(&sync[idx].timeout, 1u); break; }   // 10 ms
                __builtin_amdgcn_s_sleep(4);
            }
            if (acq == 0) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        }
        __syncthreads();
    }
    // consume the predecessor's result (every workgroup reads one line of it)
    float r = 0.f;
    if (idx > 0) {
        const unsigned int o = ((blockIdx.x % prev_grid) * RES_FLOATS + (threadIdx.x & (RES_FLOATS - 1))) * 4;
        if (acq == 1) r = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(__builtin_amdgcn_make_buffer_rsrc(res_prev, 0, 0x7fffffff, 0x00020000), o, 0, 16));
        else r = res_prev[o / 4];
        if (r != (float)(idx - 1) && threadIdx.x == 0) atomicAdd(&sync[idx].bad, 1u);
    }
    f4 acc = {r, 0, 0, 0};
    for (; c < n_chunks; c += n_waves) {
        const long cn = c + n_waves;
        f4 v2[U];
        if (cn < n_chunks) {
#pragma unroll
            for (int u = 0; u < U; ++u) v2[u] = __builtin_nontemporal_load((const f4*)(w + cn * chunk + u * 1024 + lane * 16));
        }
#pragma unroll
        for (int u = 0; u < U; ++u) { acc += v[u]; v[u] = v2[u]; }
    }
    float s = acc.x + acc.y + acc.z + acc.w;
    if (threadIdx.x < RES_FLOATS) {
        const float val = (s == 123.456f) ? s : (float)idx;
        if (rel >= 1) {
            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned int, val), __builtin_amdgcn_make_buffer_rsrc(res_mine, 0, 0x7fffffff, 0x00020000),
                                                  (blockIdx.x * RES_FLOATS + threadIdx.x) * 4, 0, 16);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        } else res_mine[blockIdx.x * RES_FLOATS + threadIdx.x] = val;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        const int xcc = __builtin_amdgcn_s_getreg(20 | (3 << 11)) & 7;     // XCC_ID
        if (rel == 1) __hip_atomic_fetch_add(&sync[idx].done[xcc][0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        else if (rel == 3) { if (blockIdx.x == 0) __hip_atomic_fetch_add(&sync[idx].done[xcc][0], prev_grid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
        else __hip_atomic_fetch_add(&sync[idx].done[xcc][0], 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        if (stamps && blockIdx.x == 0) { stamps[2 * idx] = t0; stamps[2 * idx + 1] = wall_clock64(); }
    }
}

int main(int argc, char** argv) {
    const int layers = 6;
    const long sizes_mb[5] = {396, 33, 180, 90, 100};
    const int grid = argc > 1 ? atoi(argv[1]) : 1024;
    const int N = layers * 5;
    long total = 0; std::vector<long> off(N), len(N);
    for (int i = 0; i < N; ++i) { off[i] = total; len[i] = sizes_mb[i % 5] << 20; total += len[i]; }
    char* w; CK(hipMalloc(&w, total)); CK(hipMemset(w, 0, total));
    float* res; CK(hipMalloc(&res, (size_t)N * grid * RES_FLOATS * 4));
    Sync* sync; CK(hipMalloc(&sync, sizeof(Sync) * N));
    unsigned long long* stamps; CK(hipMalloc(&stamps, 16 * N));
    hipStream_t st; CK(hipStreamCreate(&st));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto reset = [&]() { CK(hipMemsetAsync(sync, 0, sizeof(Sync) * N, st)); CK(hipMemsetAsync(res, 0xff, (size_t)N * grid * RES_FLOATS * 4, st)); CK(hipStreamSynchronize(st)); };
    double host_us = 0;
    int rel = 0, acq = 0;
    auto launch_chain = [&](int mode) {   // 0 normal, 1 any-order + wait
        auto h0 = std::chrono::steady_clock::now();
        for (int i = 0; i < N; ++i) {
            const char* wp = w + off[i]; long b = len[i];
            float* rp = res + (size_t)(i ? i - 1 : 0) * grid * RES_FLOATS; float* rm = res + (size_t)i * grid * RES_FLOATS;
            int idx = i, wait = mode == 1; unsigned int pg = grid;
            void* args[] = {&sync, &idx, &wp, &b, &rp, &rm, &wait, &pg, &stamps, &rel, &acq};
            CK(hipExtLaunchKernel((void*)stream_k, dim3(grid), dim3(256), args, 0, st, nullptr, nullptr,
                                  (mode == 1 && i > 0) ? hipExtAnyOrderLaunch : 0));
        }
        host_us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - h0).count();
    };
    auto report = [&](const char* name, float ms) {
        std::vector<Sync> h(N); CK(hipMemcpy(h.data(), sync, sizeof(Sync) * N, hipMemcpyDeviceToHost));
        std::vector<unsigned long long> s(2 * N); CK(hipMemcpy(s.data(), stamps, 16 * N, hipMemcpyDeviceToHost));
        unsigned bad = 0, to = 0; for (auto& x : h) { bad += x.bad; to += x.timeout; }
        printf("%-44s %8.1f us total = %6.2f us per 5-kernel layer (pure read at 6.9 TB/s: %.1f) | %.2f TB/s | host %.1f us/launch | stale results %u, timeouts %u\n",
               name, ms * 1e3, ms * 1e3 / layers, total / layers / 6.9e6, total / (ms * 1e-3) / 1e12, host_us / N, bad, to);
    };
    printf("chain of %d kernels (%d layers x [396,33,180,90,100] MB), grid %d x 256\n", N, layers, grid);
    float ms;
    const int combos[][2] = {{0, 0}, {1, 1}, {2, 1}, {1, 0}, {3, 1}};
    for (auto& cb : combos) {
      rel = cb[0]; acq = cb[1];
      printf("-- publish: %s; consume: %s\n", rel == 0 ? "plain stores + RELEASE atomic" : rel == 1 ? "sc1 stores + vmcnt(0) + relaxed atomic" : rel == 2 ? "sc1 stores + RELEASE atomic" : "sc1 stores, ONE relaxed atomic per kernel (invalid sync: cost probe only)",
             acq == 0 ? "acquire fence + plain loads" : "sc1 loads");
      for (int rep = 0; rep < 2; ++rep) {
        reset(); CK(hipEventRecord(e0, st)); launch_chain(0); CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
        CK(hipEventElapsedTime(&ms, e0, e1)); report("eager, stream order", ms);
        reset(); CK(hipEventRecord(e0, st)); launch_chain(1); CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
        CK(hipEventElapsedTime(&ms, e0, e1)); report("eager, any-order + in-kernel wait", ms);
      }
    }
    rel = 1; acq = 1;
    {
        hipGraph_t g; hipGraphExec_t ge;
        CK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
        launch_chain(0);
        CK(hipStreamEndCapture(st, &g));
        CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        for (int rep = 0; rep < 3; ++rep) {
            reset(); CK(hipEventRecord(e0, st)); CK(hipGraphLaunch(ge, st)); CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
            CK(hipEventElapsedTime(&ms, e0, e1)); host_us = 0; report("hipGraph, stream order", ms);
        }
    }
    return 0;
}
