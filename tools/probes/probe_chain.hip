// Probe 2: a chain of N weight-streaming kernels (each reads its own `bytes` of HBM, then needs its predecessor's
// result).  How long does the chain take when launched (a) as a captured hipGraph, (b) eagerly in stream order,
// (c) eagerly with hipExtAnyOrderLaunch + an in-kernel wait on the predecessor's done-counter (loads issued
// BEFORE the wait)?  Per-kernel start/end stamps (s_memrealtime, 100 MHz) show the boundaries.
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <algorithm>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

struct Node { unsigned long long start, end, first_data, pad; unsigned int done, timeout, pad2[6]; };   // 64 B

typedef float f4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256) void stream_k(Node* nodes, int idx, const f4* __restrict__ w, long n_vec_per_wg,
                                                float* out, int wait_prev, unsigned int prev_grid) {
    Node* me = nodes + idx;
    unsigned long long t0 = wall_clock64();
    if (threadIdx.x == 0) atomicMin(&me->start, t0);
    const f4* p = w + (long)blockIdx.x * n_vec_per_wg + threadIdx.x;
    f4 acc = {0, 0, 0, 0};
    // first batch of loads: issued before the dependency is checked
    f4 r[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) r[j] = __builtin_nontemporal_load(p + j * 256);
    if (wait_prev && idx > 0) {
        if (threadIdx.x == 0) {
            const unsigned int* d = &nodes[idx - 1].done;
            while (__hip_atomic_load(d, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < prev_grid) {
                if (wall_clock64() - t0 > 200000ull) { atomicAdd(&me->timeout, 1u); break; }
                __builtin_amdgcn_s_sleep(4);
            }
        }
        __syncthreads();
    }
    long i = 0;
    for (; i + 16 * 256 <= n_vec_per_wg; i += 8 * 256) {
        f4 r2[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) r2[j] = __builtin_nontemporal_load(p + i + (8 + j) * 256);
#pragma unroll
        for (int j = 0; j < 8; ++j) { acc += r[j]; r[j] = r2[j]; }
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) acc += r[j];
    float s = acc.x + acc.y + acc.z + acc.w;
    if (s == 123.456f) out[blockIdx.x * 256 + threadIdx.x] = s;     // keep the loads
    __syncthreads();
    if (threadIdx.x == 0) {
        out[blockIdx.x] = (float)idx;                                   // the "result"
        __threadfence();
        atomicMax(&me->end, wall_clock64());
        __hip_atomic_fetch_add(&me->done, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    }
}

int main(int argc, char** argv) {
    const int N = 24;
    const long mb = argc > 1 ? atol(argv[1]) : 64;          // MB per kernel
    const int grid = argc > 2 ? atoi(argv[2]) : 1024;
    const long bytes = mb << 20;
    const long n_vec_per_wg = bytes / 16 / grid;             // multiple of 2048 for the sizes used
    char* w; CK(hipMalloc(&w, bytes * N));
    CK(hipMemset(w, 0, bytes * N));
    float* out; CK(hipMalloc(&out, 4 * 256 * 4096));
    Node* nodes; CK(hipMalloc(&nodes, sizeof(Node) * N));
    hipStream_t st; CK(hipStreamCreate(&st));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    std::vector<Node> h(N);
    auto reset = [&]() {
        for (auto& n : h) { memset(&n, 0, sizeof n); n.start = ~0ull; }
        CK(hipMemcpyAsync(nodes, h.data(), sizeof(Node) * N, hipMemcpyHostToDevice, st));
        CK(hipStreamSynchronize(st));
    };
    auto launch_chain = [&](int mode) {   // 0 normal, 1 any-order + wait
        for (int i = 0; i < N; ++i) {
            const f4* wp = (const f4*)(w + bytes * i);
            int idx = i, wait = mode == 1; unsigned int pg = grid; long nv = n_vec_per_wg;
            void* args[] = {&nodes, &idx, &wp, &nv, &out, &wait, &pg};
            CK(hipExtLaunchKernel((void*)stream_k, dim3(grid), dim3(256), args, 0, st, nullptr, nullptr,
                                  (mode == 1 && i > 0) ? hipExtAnyOrderLaunch : 0));
        }
    };
    auto report = [&](const char* name, float ms) {
        CK(hipMemcpy(h.data(), nodes, sizeof(Node) * N, hipMemcpyDeviceToHost));
        double gaps = 0, dur = 0; unsigned int to = 0;
        for (int i = 1; i < N; ++i) gaps += ((double)h[i].start - (double)h[i - 1].end) * 0.01;
        for (int i = 0; i < N; ++i) { dur += (h[i].end - h[i].start) * 0.01; to += h[i].timeout; }
        double span = (h[N - 1].end - h[0].start) * 0.01;
        printf("%-46s events %8.1f us total, %6.2f us/kernel | stamps: span %8.1f us, mean kernel %6.2f us, mean (start[i]-end[i-1]) %+6.2f us, timeouts %u | %.2f TB/s\n",
               name, ms * 1e3, ms * 1e3 / N, span, dur / N, gaps / (N - 1), to, bytes * N / (span * 1e-6) / 1e12);
    };
    printf("chain of %d kernels, %ld MB each, grid %d x 256\n", N, mb, grid);
    for (int rep = 0; rep < 3; ++rep) {
        float ms;
        // (b) eager, stream order
        reset(); CK(hipEventRecord(e0, st)); launch_chain(0); CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
        CK(hipEventElapsedTime(&ms, e0, e1)); report("eager, stream order", ms);
        // (c) eager, any-order + in-kernel wait
        reset(); CK(hipEventRecord(e0, st)); launch_chain(1); CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
        CK(hipEventElapsedTime(&ms, e0, e1)); report("eager, any-order + in-kernel wait", ms);
    }
    // (a) graph, stream order
    for (int mode = 0; mode <= 1; ++mode) {
        hipGraph_t g; hipGraphExec_t ge;
        CK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
        launch_chain(mode);
        CK(hipStreamEndCapture(st, &g));
        CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        for (int rep = 0; rep < 3; ++rep) {
            float ms;
            reset(); CK(hipEventRecord(e0, st)); CK(hipGraphLaunch(ge, st)); CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
            CK(hipEventElapsedTime(&ms, e0, e1));
            report(mode ? "hipGraph, captured any-order + in-kernel wait" : "hipGraph, stream order", ms);
        }
        CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
    }
    return 0;
}
