// fence_cost.cpp — what does a launch boundary cost on this GPU, and how much of it are the acquire / release fences?
// Decision it serves (round 5): the decode step's 164 dispatches carry barrier = 1, acquire = release =
// agent scope (profiles/r5_fence_scope.md); would a replay mechanism of our own (AQL packets written directly, fence
// scope NONE where the data passed between two launches is written through) shorten the step?
// The probe talks to HSA directly: one queue, N back-to-back dispatches of the same kernel, barrier bit set, with the
// header's fence scopes none / agent / system; time per dispatch = (doorbell -> completion signal of the last) / N.
//   build:  hipcc --offload-arch=gfx950 --genco --no-gpu-bundle-output -O3 fence_cost_kernels.hip -o fence_cost_kernels.hsaco
//           g++ -O2 -std=c++17 fence_cost.cpp -I/opt/rocm/include -L/opt/rocm/lib -lhsa-runtime64 -o fence_cost
//   run:    ./fence_cost fence_cost_kernels.hsaco
#include <hsa/hsa.h>
#include <hsa/hsa_ext_amd.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <vector>

#define CHECK(x) do { hsa_status_t s_ = (x); if (s_ != HSA_STATUS_SUCCESS) { const char* m = ""; hsa_status_string(s_, &m); \
  fprintf(stderr, "%s:%d %s -> %s\n", __FILE__, __LINE__, #x, m); exit(1); } } while (0)

static hsa_agent_t g_gpu; static bool g_have_gpu = false;
static hsa_region_t g_kernarg, g_local; static bool g_have_kernarg = false, g_have_local = false;

static hsa_status_t on_agent(hsa_agent_t a, void*) {
  hsa_device_type_t t; hsa_agent_get_info(a, HSA_AGENT_INFO_DEVICE, &t);
  if (t == HSA_DEVICE_TYPE_GPU && !g_have_gpu) { g_gpu = a; g_have_gpu = true; }
  return HSA_STATUS_SUCCESS;
}
static hsa_status_t on_region(hsa_region_t r, void*) {
  hsa_region_segment_t seg; hsa_region_get_info(r, HSA_REGION_INFO_SEGMENT, &seg);
  if (seg != HSA_REGION_SEGMENT_GLOBAL) return HSA_STATUS_SUCCESS;
  uint32_t flags; hsa_region_get_info(r, HSA_REGION_INFO_GLOBAL_FLAGS, &flags);
  if ((flags & HSA_REGION_GLOBAL_FLAG_KERNARG) && !g_have_kernarg) { g_kernarg = r; g_have_kernarg = true; }
  if ((flags & HSA_REGION_GLOBAL_FLAG_COARSE_GRAINED) && !g_have_local) { g_local = r; g_have_local = true; }
  return HSA_STATUS_SUCCESS;
}

struct Kernel { uint64_t object; uint32_t kernarg_size, group, priv; };
static Kernel get_kernel(hsa_executable_t exe, const char* name) {
  hsa_executable_symbol_t sym; std::string n = std::string(name) + ".kd";
  CHECK(hsa_executable_get_symbol_by_name(exe, n.c_str(), &g_gpu, &sym));
  Kernel k;
  CHECK(hsa_executable_symbol_get_info(sym, HSA_EXECUTABLE_SYMBOL_INFO_KERNEL_OBJECT, &k.object));
  CHECK(hsa_executable_symbol_get_info(sym, HSA_EXECUTABLE_SYMBOL_INFO_KERNEL_KERNARG_SEGMENT_SIZE, &k.kernarg_size));
  CHECK(hsa_executable_symbol_get_info(sym, HSA_EXECUTABLE_SYMBOL_INFO_KERNEL_GROUP_SEGMENT_SIZE, &k.group));
  CHECK(hsa_executable_symbol_get_info(sym, HSA_EXECUTABLE_SYMBOL_INFO_KERNEL_PRIVATE_SEGMENT_SIZE, &k.priv));
  return k;
}

int main(int argc, char** argv) {
  if (argc < 2) { fprintf(stderr, "usage: fence_cost <kernels.hsaco>\n"); return 2; }
  CHECK(hsa_init());
  CHECK(hsa_iterate_agents(on_agent, nullptr));
  if (!g_have_gpu) { fprintf(stderr, "no GPU agent\n"); return 1; }
  CHECK(hsa_agent_iterate_regions(g_gpu, on_region, nullptr));
  if (!g_have_kernarg || !g_have_local) { fprintf(stderr, "regions missing\n"); return 1; }
  std::ifstream f(argv[1], std::ios::binary); std::vector<char> blob((std::istreambuf_iterator<char>(f)), {});
  hsa_code_object_reader_t reader; CHECK(hsa_code_object_reader_create_from_memory(blob.data(), blob.size(), &reader));
  hsa_executable_t exe; CHECK(hsa_executable_create_alt(HSA_PROFILE_FULL, HSA_DEFAULT_FLOAT_ROUNDING_MODE_DEFAULT, nullptr, &exe));
  CHECK(hsa_executable_load_agent_code_object(exe, g_gpu, reader, nullptr, nullptr));
  CHECK(hsa_executable_freeze(exe, nullptr));
  Kernel kernels[2] = {get_kernel(exe, "k_empty"), get_kernel(exe, "k_stream")};
  for (auto& k : kernels) fprintf(stderr, "kernel object %llx kernarg %u group %u private %u\n", (unsigned long long)k.object, k.kernarg_size, k.group, k.priv);
  hsa_queue_t* q;
  CHECK(hsa_queue_create(g_gpu, 16384, HSA_QUEUE_TYPE_SINGLE,
                         [](hsa_status_t st, hsa_queue_t*, void*) { const char* m = ""; hsa_status_string(st, &m); fprintf(stderr, "queue error: %s\n", m); },
                         nullptr, UINT32_MAX, UINT32_MAX, &q));
  const size_t src_bytes = 1ull << 30;           // walked 32 MiB at a time: always cold
  void *src, *dst; CHECK(hsa_memory_allocate(g_local, src_bytes, &src)); CHECK(hsa_memory_allocate(g_local, 4 << 20, &dst));
  struct Args { void* src; void* dst; long n_chunks; long n_waves; char hidden[512 - 32]; };   // room for the hidden arguments the kernel never reads
  const int N = 2000;
  Args* args; CHECK(hsa_memory_allocate(g_kernarg, sizeof(Args) * N, (void**)&args));
  hsa_signal_t done; CHECK(hsa_signal_create(1, 0, nullptr, &done));
  const char* names[3] = {"none", "agent", "system"};
  const int scopes[3] = {HSA_FENCE_SCOPE_NONE, HSA_FENCE_SCOPE_AGENT, HSA_FENCE_SCOPE_SYSTEM};
  for (int kidx = 0; kidx < 2; ++kidx) {
    const long chunk_bytes = 32l << 20;          // k_stream: 32 MiB per dispatch (the o projection's weights)
    for (int rep = 0; rep < 2; ++rep)
    for (int v = 0; v < 3; ++v) {
      for (int i = 0; i < N; ++i) { memset(&args[i], 0, sizeof(Args)); args[i].src = (char*)src + (size_t)(i % 32) * chunk_bytes; args[i].dst = dst; args[i].n_chunks = kidx ? chunk_bytes / 8192 : 0; args[i].n_waves = 1024; }
      hsa_signal_store_relaxed(done, 1);
      uint64_t idx = hsa_queue_add_write_index_relaxed(q, N);
      for (int i = 0; i < N; ++i) {
        hsa_kernel_dispatch_packet_t* pkt = (hsa_kernel_dispatch_packet_t*)q->base_address + ((idx + i) & (q->size - 1));
        memset(((char*)pkt) + 4, 0, sizeof(*pkt) - 4);
        pkt->setup = 1 << HSA_KERNEL_DISPATCH_PACKET_SETUP_DIMENSIONS;
        pkt->workgroup_size_x = 256; pkt->workgroup_size_y = 1; pkt->workgroup_size_z = 1;
        pkt->grid_size_x = 256 * 256; pkt->grid_size_y = 1; pkt->grid_size_z = 1;
        pkt->kernel_object = kernels[kidx].object; pkt->kernarg_address = &args[i];
        pkt->private_segment_size = kernels[kidx].priv; pkt->group_segment_size = kernels[kidx].group;
        pkt->completion_signal = (i == N - 1) ? done : hsa_signal_t{0};
        // the last packet always releases at system scope so that the host sees the signal after everything
        const int sc = (i == N - 1) ? HSA_FENCE_SCOPE_SYSTEM : scopes[v];
        uint16_t header = (HSA_PACKET_TYPE_KERNEL_DISPATCH << HSA_PACKET_HEADER_TYPE) | (1 << HSA_PACKET_HEADER_BARRIER) |
                          (sc << HSA_PACKET_HEADER_SCACQUIRE_FENCE_SCOPE) | (sc << HSA_PACKET_HEADER_SCRELEASE_FENCE_SCOPE);
        __atomic_store_n((uint16_t*)pkt, header, __ATOMIC_RELEASE);
      }
      auto t0 = std::chrono::steady_clock::now();
      hsa_signal_store_screlease(q->doorbell_signal, idx + N - 1);
      int spins = 0;
      while (hsa_signal_wait_scacquire(done, HSA_SIGNAL_CONDITION_LT, 1, 2000000000ull, HSA_WAIT_STATE_BLOCKED) != 0) {
        if (++spins > 10) { fprintf(stderr, "no completion after %d waits (kernel %d, variant %d): giving up\n", spins, kidx, v); return 3; }
      }
      auto t1 = std::chrono::steady_clock::now();
      const double us = std::chrono::duration<double, std::micro>(t1 - t0).count() / N;
      if (rep == 1) printf("%-8s fences %-6s : %7.3f us per dispatch%s\n", kidx ? "k_stream" : "k_empty", names[v], us,
                           kidx ? "  (32 MiB read + 1 MiB written per dispatch, 256 workgroups)" : "  (256 empty workgroups)");
    }
  }
  return 0;
}
