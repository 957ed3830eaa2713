// Probe 5: per-XCD work pools with stealing vs a static partition, pure HBM read.  `items` equal contiguous pieces of
// the buffer; static: workgroup i reads items i, i + grid, ...; pools: item j belongs to pool j % 8, a workgroup draws
// from the pool of the XCD it runs on (one atomic on that pool's line, claimed one item AHEAD of the one being read)
// and, when that pool is empty, from the next XCD's.  Fast XCDs end up reading more.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)
typedef float f4 __attribute__((ext_vector_type(4)));
struct Stat { unsigned long long first_start_c, last_end[8], items[8]; unsigned int pool[8][32]; };

__device__ __forceinline__ long claim(Stat* st, int xcc, long per_pool) {   // thread 0 only; -1 = nothing left anywhere
    for (int k = 0; k < 8; ++k) {
        const int pl = (xcc + k) & 7;
        if (__hip_atomic_load(&st->pool[pl][0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= per_pool) continue;
        const unsigned int j = atomicAdd(&st->pool[pl][0], 1u);
        if (j < per_pool) return (long)j * 8 + pl;
    }
    return -1;
}

__global__ __launch_bounds__(256) void read_k(const char* __restrict__ base, long n_items, long item_bytes, int mode, Stat* st, float* sink) {
    __shared__ long s_next;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int xcc = __builtin_amdgcn_s_getreg(20 | (3 << 11)) & 7;
    if (threadIdx.x == 0) atomicMax(&st->first_start_c, ~(unsigned long long)wall_clock64());
    f4 acc = {0, 0, 0, 0};
    long mine = 0;
    long c = mode ? -2 : blockIdx.x;
    if (mode) {
        if (threadIdx.x == 0) s_next = claim(st, xcc, n_items / 8);
        __syncthreads();
        c = s_next;
        __syncthreads();
    }
    while (c >= 0 && c < n_items) {
        if (mode && threadIdx.x == 0) s_next = claim(st, xcc, n_items / 8);      // one item ahead, under this item's loads
        const char* p = base + c * item_bytes + (long)w * (item_bytes / 4) + lane * 16;
        const long n_kib = item_bytes / 4 / 1024;          // KiB per wave
        for (long i0 = 0; i0 < n_kib; i0 += 16) {
            f4 v[16];
#pragma unroll
            for (int i = 0; i < 16; ++i) v[i] = __builtin_nontemporal_load((const f4*)(p + (i0 + i) * 1024));
#pragma unroll
            for (int i = 0; i < 16; ++i) acc += v[i];
        }
        ++mine;
        if (mode) { __syncthreads(); c = s_next; __syncthreads(); }
        else c += gridDim.x;
    }
    if (acc.x + acc.y + acc.z + acc.w == 123.456f) sink[0] = acc.x;
    if (threadIdx.x == 0) {
        atomicMax(&st->last_end[xcc], (unsigned long long)wall_clock64());
        atomicAdd(&st->items[xcc], (unsigned long long)mine);
    }
}

int main(int argc, char** argv) {
    const long mb = argc > 1 ? atol(argv[1]) : 416;
    const int grid = argc > 2 ? atoi(argv[2]) : 1024;
    const long item_kb = argc > 3 ? atol(argv[3]) : 128;
    const long bytes = mb << 20, item_bytes = item_kb << 10, n_items = bytes / item_bytes / 8 * 8;
    char* buf; CK(hipMalloc(&buf, bytes * 4)); CK(hipMemset(buf, 1, bytes * 4));
    Stat* st; CK(hipMalloc(&st, sizeof(Stat)));
    float* sink; CK(hipMalloc(&sink, 64));
    hipStream_t s; CK(hipStreamCreate(&s));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    printf("read %ld MB as %ld items of %ld KiB, grid %d x 256\n", mb, n_items, item_kb, grid);
    for (int rep = 0; rep < 6; ++rep) {
        for (int mode = 0; mode <= 1; ++mode) {
            CK(hipMemsetAsync(st, 0, sizeof(Stat), s));
            CK(hipEventRecord(e0, s));
            read_k<<<grid, 256, 0, s>>>(buf + (rep % 4) * bytes, n_items, item_bytes, mode, st, sink);
            CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            Stat h; CK(hipMemcpy(&h, st, sizeof h, hipMemcpyDeviceToHost));
            if (rep < 2) continue;
            unsigned long long t0 = ~h.first_start_c;
            printf("%s %7.1f us %5.2f TB/s | XCD end us:", mode ? "pools  " : "static ", ms * 1e3, n_items * item_bytes / (ms * 1e-3) / 1e12);
            for (int i = 0; i < 8; ++i) printf(" %5.1f", (h.last_end[i] - t0) * 0.01);
            printf(" | items per XCD:");
            for (int i = 0; i < 8; ++i) printf(" %4llu", h.items[i]);
            printf("\n");
        }
    }
    return 0;
}
