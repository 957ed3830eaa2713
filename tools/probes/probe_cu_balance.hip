// Probe (round 6): is a ragged decode-attention launch bound by the bytes of its HEAVIEST CU?
// 1024 workgroups of 4 waves (4 per CU), each streaming its own piece of a buffer with the decode kernel's access shape
// (two 8 KiB requests per wave in flight).  Workgroup w reads len[w] KiB.  Cases: all equal; bimodal (half 130, half 830
// units) placed (a) at random, (b) so that workgroups w, w+256, w+512, w+768 — one CU's four, IF the dispatcher deals
// workgroups round-robin over XCDs and CUs — hold two long and two short.  Records where every workgroup ran.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#include <map>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)
typedef float f4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256, 4) void read_k(const char* __restrict__ base, const int* __restrict__ off_kib, const int* __restrict__ len_kib,
                                                 unsigned* __restrict__ where, unsigned long long* __restrict__ t_end, float* sink) {
    __shared__ float pad[6 * 1024];      // 24 KiB of LDS like the decode kernel: 4 workgroups per CU by registers anyway
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    if (threadIdx.x == 0) {
        const unsigned hw = __builtin_amdgcn_s_getreg(4 | (0 << 6) | (31 << 11));      // HW_ID
        const unsigned xcc = __builtin_amdgcn_s_getreg(20 | (3 << 11)) & 15;
        where[blockIdx.x] = (xcc << 16) | (hw & 0xffff);
    }
    const char* p = base + (size_t)off_kib[blockIdx.x] * 1024 + (size_t)w * 8192 + lane * 16;
    const int n = len_kib[blockIdx.x] / 32;      // rounds of 32 KiB per workgroup (8 KiB per wave)
    f4 acc = {0, 0, 0, 0};
    f4 a[8], b[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) a[i] = __builtin_nontemporal_load((const f4*)(p + i * 1024));
    for (int r = 0; r < n; r += 2) {
        const char* q = p + (size_t)(r + 1) * 32768;
#pragma unroll
        for (int i = 0; i < 8; ++i) b[i] = __builtin_nontemporal_load((const f4*)(q + i * 1024));
#pragma unroll
        for (int i = 0; i < 8; ++i) acc += a[i];
        q += 32768;
#pragma unroll
        for (int i = 0; i < 8; ++i) a[i] = __builtin_nontemporal_load((const f4*)(q + i * 1024));
#pragma unroll
        for (int i = 0; i < 8; ++i) acc += b[i];
    }
    if (acc.x + acc.y + acc.z + acc.w == 123.456f) { sink[0] = acc.x; pad[threadIdx.x] = acc.y; }
    if (threadIdx.x == 0) t_end[blockIdx.x] = wall_clock64();
}

int main() {
    const int G = 1024;
    const size_t cap_kib = (size_t)G * 1024 + 4096;      // up to 1 MiB per workgroup
    char* buf; CK(hipMalloc(&buf, (cap_kib + 128) * 1024)); CK(hipMemset(buf, 1, (cap_kib + 128) * 1024));
    int *d_off, *d_len; unsigned* d_where; unsigned long long* d_end; float* sink;
    CK(hipMalloc(&d_off, G * 4)); CK(hipMalloc(&d_len, G * 4)); CK(hipMalloc(&d_where, G * 4)); CK(hipMalloc(&d_end, G * 8)); CK(hipMalloc(&sink, 64));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto run = [&](const char* name, const std::vector<int>& len) {
        std::vector<int> off(G); size_t o = 0, total = 0;
        for (int i = 0; i < G; ++i) { off[i] = (int)o; o += len[i] + 64; total += len[i]; }
        CK(hipMemcpy(d_off, off.data(), G * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(d_len, len.data(), G * 4, hipMemcpyHostToDevice));
        float best = 1e9, sum = 0; int nrep = 6;
        for (int rep = 0; rep < nrep + 2; ++rep) {
            CK(hipEventRecord(e0)); read_k<<<G, 256>>>(buf, d_off, d_len, d_where, d_end, sink); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (rep >= 2) { best = std::min(best, ms); sum += ms; }
        }
        std::vector<unsigned> where(G); CK(hipMemcpy(where.data(), d_where, G * 4, hipMemcpyDeviceToHost));
        std::map<unsigned, std::vector<int>> by_cu;
        for (int i = 0; i < G; ++i) by_cu[where[i] & 0xfffff00u].push_back(i);      // xcc | se | sh | cu  (wave / simd / pipe masked off)
        size_t heaviest = 0; int max_wgs = 0, min_wgs = 99;
        for (auto& kv : by_cu) { size_t s = 0; for (int i : kv.second) s += len[i]; heaviest = std::max(heaviest, s); max_wgs = std::max<int>(max_wgs, kv.second.size()); min_wgs = std::min<int>(min_wgs, kv.second.size()); }
        printf("%-34s mean %6.1f us (best %6.1f)  %5.2f TB/s | CUs seen %zu, workgroups per CU %d..%d, heaviest CU %zu KiB = %.2f x mean -> %.1f GB/s on that CU\n", name,
               sum / nrep * 1e3, best * 1e3, total * 1024.0 / (sum / nrep * 1e-3) / 1e12, by_cu.size(), min_wgs, max_wgs, heaviest, heaviest / (total / 256.0),
               heaviest * 1024.0 / (sum / nrep * 1e-3) / 1e9);
        return by_cu;
    };
    // 1 unit = 32 KiB = 64 keys of one head (K + V rows of 256 B): 13 units = 832 keys
    std::vector<int> eq(G, 13 * 32);
    auto placement = run("all 416 KiB (832 keys)", eq);
    {   // does workgroup w share its CU with w + 256, w + 512, w + 768?
        int ok = 0, n = 0;
        for (auto& kv : placement) { ++n; std::vector<int> v = kv.second; std::sort(v.begin(), v.end()); bool g = v.size() == 4; for (size_t i = 1; g && i < v.size(); ++i) g = (v[i] - v[0]) % 256 == 0; ok += g; }
        printf("   CUs whose four workgroups are {w, w+256, w+512, w+768}: %d of %d\n", ok, n);
        int shown = 0;
        for (auto& kv : placement) { if (shown++ >= 6) break; printf("   id %08x:", kv.first); for (int i : kv.second) printf(" %d", i); printf("\n"); }
    }
    srand(1);
    std::vector<int> bi(G);
    for (int i = 0; i < G; ++i) bi[i] = (i < G / 2 ? 2 : 13) * 32;
    std::vector<int> rnd = bi; std::random_shuffle(rnd.begin(), rnd.end());
    run("bimodal 64 / 416 KiB, shuffled", rnd);
    std::vector<int> bal(G);
    for (int i = 0; i < G; ++i) bal[i] = (((i / 256) & 1) ? 13 : 2) * 32;      // slots 0, 2 short; 1, 3 long
    run("bimodal, two long per assumed CU", bal);
    std::vector<int> byseq(G);      // like the decode grid: sequence = w / 32, lengths bimodal in shuffled sequence order
    { std::vector<int> seq(32); for (int i = 0; i < 32; ++i) seq[i] = (i < 16 ? 2 : 13) * 32; std::random_shuffle(seq.begin(), seq.end());
      for (int i = 0; i < G; ++i) byseq[i] = seq[i / 32]; }
    run("bimodal by sequence (w / 32)", byseq);
    std::vector<int> worst(G);
    for (int i = 0; i < G; ++i) worst[i] = ((i % 256) < 128 ? 13 : 2) * 32;      // assumed CUs 0..127 all long
    run("bimodal, four long per assumed CU", worst);
    std::vector<int> half(G, 7 * 32);
    run("all 224 KiB (even split of bimodal)", half);
    return 0;
}
