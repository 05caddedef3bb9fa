// hsa_threads_probe.cpp — do ROCr's VMM calls scale across threads of one process? (HIP's do not: vmm_probe.)
// T threads, each maps + grants + unmaps its own share of n slots, all start together. Wall time per slot.
// build: hipcc --offload-arch=gfx950 -O2 -o hsa_threads_probe hsa_threads_probe.cpp -lhsa-runtime64 -pthread
#include <hip/hip_runtime.h>
#include <hsa/hsa.h>
#include <hsa/hsa_ext_amd.h>

#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <thread>
#include <vector>

static double now_us() {
  return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count();
}
static hsa_agent_t g_gpu;
static hsa_amd_memory_pool_t g_pool;
static bool g_hg = false, g_hp = false;

int main(int argc, char **argv) {
  const int n = argc > 1 ? atoi(argv[1]) : 4096;
  const size_t PAGE = 2u << 20;
  if (hipSetDevice(0) != hipSuccess || hipFree(nullptr) != hipSuccess) return 1;
  hsa_init();
  hsa_iterate_agents([](hsa_agent_t a, void *) -> hsa_status_t {
    hsa_device_type_t t; hsa_agent_get_info(a, HSA_AGENT_INFO_DEVICE, &t);
    if (t == HSA_DEVICE_TYPE_GPU && !g_hg) { g_gpu = a; g_hg = true; }
    return HSA_STATUS_SUCCESS; }, nullptr);
  hsa_amd_agent_iterate_memory_pools(g_gpu, [](hsa_amd_memory_pool_t p, void *) -> hsa_status_t {
    hsa_amd_segment_t seg; bool alloc = false; uint32_t fl = 0;
    hsa_amd_memory_pool_get_info(p, HSA_AMD_MEMORY_POOL_INFO_SEGMENT, &seg);
    hsa_amd_memory_pool_get_info(p, HSA_AMD_MEMORY_POOL_INFO_RUNTIME_ALLOC_ALLOWED, &alloc);
    hsa_amd_memory_pool_get_info(p, HSA_AMD_MEMORY_POOL_INFO_GLOBAL_FLAGS, &fl);
    if (seg == HSA_AMD_SEGMENT_GLOBAL && alloc && (fl & HSA_AMD_MEMORY_POOL_GLOBAL_FLAG_COARSE_GRAINED) && !g_hp) { g_pool = p; g_hp = true; }
    return HSA_STATUS_SUCCESS; }, nullptr);
  void *va0 = nullptr;
  if (hsa_amd_vmem_address_reserve_align(&va0, (size_t)n * PAGE, 0, PAGE, 0) != HSA_STATUS_SUCCESS) return 2;
  char *va = (char *)va0;
  std::vector<hsa_amd_vmem_alloc_handle_t> h(n);
  for (int i = 0; i < n; i++)
    if (hsa_amd_vmem_handle_create(g_pool, PAGE, MEMORY_TYPE_PINNED, 0, &h[i]) != HSA_STATUS_SUCCESS) return 3;
  hsa_amd_memory_access_desc_t acc{HSA_ACCESS_PERMISSION_RW, g_gpu};
  for (int T : {1, 2, 4, 8}) {
    for (int phase = 0; phase < 2; phase++) { // phase 0: everybody maps then everybody unmaps; phase 1: half the threads map while the other half unmaps
      std::atomic<int> go{0};
      std::vector<std::thread> th;
      std::vector<double> tmap(T), tun(T);
      const int per = n / T;
      for (int t = 0; t < T; t++)
        th.emplace_back([&, t] {
          while (!go.load()) {}
          double a = now_us();
          for (int i = t * per; i < (t + 1) * per; i++) {
            hsa_amd_vmem_map(va + (size_t)i * PAGE, PAGE, 0, h[i], 0);
            hsa_amd_vmem_set_access(va + (size_t)i * PAGE, PAGE, &acc, 1);
          }
          double b = now_us();
          for (int i = t * per; i < (t + 1) * per; i++) hsa_amd_vmem_unmap(va + (size_t)i * PAGE, PAGE);
          double c = now_us();
          tmap[t] = b - a;
          tun[t] = c - b;
        });
      double w0 = now_us();
      go = 1;
      for (auto &x : th) x.join();
      double w1 = now_us();
      if (phase == 0) {
        double m = 0, u = 0;
        for (int t = 0; t < T; t++) { m = m > tmap[t] ? m : tmap[t]; u = u > tun[t] ? u : tun[t]; }
        printf("threads=%d: %d slots map+access+unmap in %.0f us wall = %.2f us/slot (1 thread would need ~%.2f); slowest thread map %.2f unmap %.2f us/slot of its share\n",
               T, n, w1 - w0, (w1 - w0) / n, 8.2, m / per, u / per);
      }
    }
  }
  return 0;
}
