// Probe: can a kernel launched with hipExtAnyOrderLaunch start while its predecessor in the same
// stream is still draining (AQL barrier bit cleared), on the stream and inside a captured hipGraph?
// Times are s_memrealtime ticks (100 MHz).  Build: hipcc --offload-arch=gfx950 -O2 -o probe_overlap probe_overlap.hip
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

struct Stamps { unsigned long long a_start, a_end, b_start, b_end, b_saw; unsigned int a_done, b_timeout, pad; };

__global__ void kA(Stamps* s, int base_ticks, int spread) {
    unsigned long long t0 = wall_clock64();
    if (threadIdx.x == 0) atomicMin(&s->a_start, t0);
    unsigned long long dur = base_ticks + (blockIdx.x % 16) * spread;
    while (wall_clock64() - t0 < dur) __builtin_amdgcn_s_sleep(2);
    __syncthreads();
    if (threadIdx.x == 0) {
        atomicMax(&s->a_end, wall_clock64());
        __threadfence();
        atomicAdd(&s->a_done, 1u);
    }
}

__global__ void kB(Stamps* s, unsigned int a_grid, int wait_for_a) {
    unsigned long long t0 = wall_clock64();
    if (threadIdx.x == 0) {
        atomicMin(&s->b_start, t0);
        if (wait_for_a) {
            unsigned int v;
            while ((v = __hip_atomic_load(&s->a_done, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) < a_grid) {
                if (wall_clock64() - t0 > 100000ull) { atomicAdd(&s->b_timeout, 1u); break; }   // 1 ms
                __builtin_amdgcn_s_sleep(8);
            }
            atomicMax(&s->b_saw, wall_clock64());
        }
    }
    __syncthreads();
    if (threadIdx.x == 0) atomicMax(&s->b_end, wall_clock64());
}

static void reset(Stamps* d, hipStream_t st) {
    Stamps h; memset(&h, 0, sizeof h); h.a_start = ~0ull; h.b_start = ~0ull;
    CK(hipMemcpyAsync(d, &h, sizeof h, hipMemcpyHostToDevice, st));
}
static void report(const char* name, Stamps* d) {
    Stamps h; CK(hipMemcpy(&h, d, sizeof h, hipMemcpyDeviceToHost));
    double us = 0.01;
    printf("%-44s A: 0 .. %7.2f us | B start %+8.2f us after A's last WG ended (negative = overlap), B end %+8.2f, saw-done %+8.2f, a_done %u, timeouts %u\n",
           name, (h.a_end - h.a_start) * us, ((double)h.b_start - (double)h.a_end) * us, ((double)h.b_end - (double)h.a_end) * us,
           h.b_saw ? ((double)h.b_saw - (double)h.a_end) * us : 0.0, h.a_done, h.b_timeout);
}

int main(int argc, char** argv) {
    int a_grid = argc > 1 ? atoi(argv[1]) : 2048;
    int b_grid = 256;
    Stamps* d; CK(hipMalloc(&d, sizeof(Stamps)));
    hipStream_t st; CK(hipStreamCreate(&st));
    int base = 2000, spread = 100;      // 20 us + 0..15 us stagger
    void* argsA[] = {&d, &base, &spread};
    for (int wait = 0; wait <= 1; ++wait) {
        unsigned int ag = a_grid; int w = wait;
        void* argsB[] = {&d, &ag, &w};
        for (int rep = 0; rep < 2; ++rep) {
            // 1. plain stream order
            reset(d, st);
            CK(hipExtLaunchKernel((void*)kA, dim3(a_grid), dim3(256), argsA, 0, st, nullptr, nullptr, 0));
            CK(hipExtLaunchKernel((void*)kB, dim3(b_grid), dim3(256), argsB, 0, st, nullptr, nullptr, 0));
            CK(hipStreamSynchronize(st));
            report(wait ? "stream, B normal, B waits on A's counter" : "stream, B normal", d);
            // 2. any-order B
            reset(d, st);
            CK(hipExtLaunchKernel((void*)kA, dim3(a_grid), dim3(256), argsA, 0, st, nullptr, nullptr, 0));
            CK(hipExtLaunchKernel((void*)kB, dim3(b_grid), dim3(256), argsB, 0, st, nullptr, nullptr, hipExtAnyOrderLaunch));
            CK(hipStreamSynchronize(st));
            report(wait ? "stream, B any-order, B waits on A's counter" : "stream, B any-order", d);
        }
        // 3. captured into a graph
        hipGraph_t g; hipGraphExec_t ge;
        CK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
        CK(hipExtLaunchKernel((void*)kA, dim3(a_grid), dim3(256), argsA, 0, st, nullptr, nullptr, 0));
        hipError_t e = hipExtLaunchKernel((void*)kB, dim3(b_grid), dim3(256), argsB, 0, st, nullptr, nullptr, hipExtAnyOrderLaunch);
        CK(hipStreamEndCapture(st, &g));
        if (e != hipSuccess) { printf("capture of any-order launch failed: %s\n", hipGetErrorString(e)); }
        else {
            CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
            for (int rep = 0; rep < 2; ++rep) {
                reset(d, st); CK(hipStreamSynchronize(st));
                CK(hipGraphLaunch(ge, st)); CK(hipStreamSynchronize(st));
                report(wait ? "graph (captured), B any-order, B waits" : "graph (captured), B any-order", d);
            }
            CK(hipGraphExecDestroy(ge));
        }
        CK(hipGraphDestroy(g));
        // 4. manual graph, A and B without an edge (parallel branches)
        CK(hipGraphCreate(&g, 0));
        hipKernelNodeParams pa = {}, pb = {};
        pa.func = (void*)kA; pa.gridDim = dim3(a_grid); pa.blockDim = dim3(256); pa.kernelParams = argsA;
        pb.func = (void*)kB; pb.gridDim = dim3(b_grid); pb.blockDim = dim3(256); pb.kernelParams = argsB;
        hipGraphNode_t na, nb;
        CK(hipGraphAddKernelNode(&na, g, nullptr, 0, &pa));
        CK(hipGraphAddKernelNode(&nb, g, nullptr, 0, &pb));
        CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        for (int rep = 0; rep < 2; ++rep) {
            reset(d, st); CK(hipStreamSynchronize(st));
            CK(hipGraphLaunch(ge, st)); CK(hipStreamSynchronize(st));
            report(wait ? "graph, A || B (no edge), B waits" : "graph, A || B (no edge)", d);
        }
        CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
    }
    // 5. two streams: A on s1, B on s2 launched right after (no dependency) — reference for concurrent queues
    hipStream_t s2; CK(hipStreamCreate(&s2));
    { unsigned int ag = a_grid; int w = 1; void* argsB[] = {&d, &ag, &w};
      for (int rep = 0; rep < 2; ++rep) {
        reset(d, st); CK(hipStreamSynchronize(st));
        CK(hipExtLaunchKernel((void*)kA, dim3(a_grid), dim3(256), argsA, 0, st, nullptr, nullptr, 0));
        CK(hipExtLaunchKernel((void*)kB, dim3(b_grid), dim3(256), argsB, 0, s2, nullptr, nullptr, 0));
        CK(hipStreamSynchronize(st)); CK(hipStreamSynchronize(s2));
        report("two streams, B waits", d);
      } }
    return 0;
}
