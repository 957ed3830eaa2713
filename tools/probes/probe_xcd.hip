// Probe 4: do the eight XCDs stream HBM at the same rate?  A pure read of `mb` MB by `grid` workgroups, (a) statically
// partitioned (workgroup i reads chunks i, i + grid, ... of 64 KiB — every XCD gets the same bytes), (b) dynamically:
// every workgroup draws the next chunk from ONE atomic ticket counter.  Per XCD: when its last workgroup ended, how many
// bytes it read.  If XCDs differ, (a) ends with the slowest XCD, (b) with the average.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)
typedef float f4 __attribute__((ext_vector_type(4)));
struct Stat { unsigned long long first_start_c, last_end[8], chunks[8]; unsigned int ticket, pad; };

template <int CHUNK_KB>
__global__ __launch_bounds__(256) void read_k(const char* __restrict__ base, long n_chunks, int dynamic, Stat* st, float* sink) {
    __shared__ long s_c;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    if (threadIdx.x == 0) atomicMax(&st->first_start_c, ~(unsigned long long)wall_clock64());
    f4 acc = {0, 0, 0, 0};
    long mine = 0;
    long c = blockIdx.x;
    for (;;) {
        if (dynamic) {
            if (threadIdx.x == 0) s_c = atomicAdd(&st->ticket, 1u);
            __syncthreads();
            c = s_c;
            __syncthreads();
        }
        if (c >= n_chunks) break;
        const char* p = base + c * (long)(CHUNK_KB * 1024) + w * (CHUNK_KB * 256) + lane * 16;   // each wave a quarter of the chunk
        constexpr int NL = CHUNK_KB * 256 / 1024;     // 1 KiB per wave instruction
        f4 v[NL < 16 ? NL : 16];
#pragma unroll
        for (int i0 = 0; i0 < NL; i0 += 16) {
#pragma unroll
            for (int i = 0; i < 16 && i0 + i < NL; ++i) v[i] = __builtin_nontemporal_load((const f4*)(p + (i0 + i) * 1024));
#pragma unroll
            for (int i = 0; i < 16 && i0 + i < NL; ++i) acc += v[i];
        }
        ++mine;
        if (!dynamic) c += gridDim.x;
    }
    if (acc.x + acc.y + acc.z + acc.w == 123.456f) sink[0] = acc.x;
    if (threadIdx.x == 0) {
        const int xcc = __builtin_amdgcn_s_getreg(20 | (3 << 11)) & 7;
        atomicMax(&st->last_end[xcc], (unsigned long long)wall_clock64());
        atomicAdd(&st->chunks[xcc], (unsigned long long)mine);
    }
}

int main(int argc, char** argv) {
    const long mb = argc > 1 ? atol(argv[1]) : 400;
    const int grid = argc > 2 ? atoi(argv[2]) : 1024;
    constexpr int CK_KB = 64;
    const long bytes = mb << 20, n_chunks = bytes / (CK_KB * 1024);
    char* buf; CK(hipMalloc(&buf, bytes * 4)); CK(hipMemset(buf, 1, bytes * 4));
    Stat* st; CK(hipMalloc(&st, sizeof(Stat)));
    float* sink; CK(hipMalloc(&sink, 64));
    hipStream_t s; CK(hipStreamCreate(&s));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    printf("read %ld MB in %d-KiB chunks, grid %d x 256\n", mb, CK_KB, grid);
    for (int rep = 0; rep < 6; ++rep) {
        for (int dyn = 0; dyn <= 1; ++dyn) {
            CK(hipMemsetAsync(st, 0, sizeof(Stat), s));
            CK(hipEventRecord(e0, s));
            read_k<CK_KB><<<grid, 256, 0, s>>>(buf + (rep % 4) * bytes, n_chunks, dyn, st, sink);
            CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            Stat h; CK(hipMemcpy(&h, st, sizeof h, hipMemcpyDeviceToHost));
            if (rep < 2) continue;
            unsigned long long t0 = ~h.first_start_c;
            printf("%s %7.1f us %5.2f TB/s | XCD end us:", dyn ? "tickets" : "static ", ms * 1e3, bytes / (ms * 1e-3) / 1e12);
            for (int i = 0; i < 8; ++i) printf(" %5.1f", (h.last_end[i] - t0) * 0.01);
            printf(" | MB per XCD:");
            for (int i = 0; i < 8; ++i) printf(" %4.0f", h.chunks[i] * (CK_KB / 1024.0));
            printf("\n");
        }
    }
    return 0;
}
