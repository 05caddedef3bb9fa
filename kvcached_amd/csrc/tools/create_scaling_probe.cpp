// create_scaling_probe.cpp — creating a physical handle costs O(live handles) (4 us at 1k, 25-70 us at 24-28k). Where?
// Creates 32k x 2 MiB handles through ROCr in chunks of 4096 and prints, per chunk, the wall time per create and —
// when run under LD_PRELOAD=ioctl_timer.so — the time spent inside ioctls (ioctl_timer_dump between chunks).
// build: hipcc --offload-arch=gfx950 -O2 -o create_scaling_probe create_scaling_probe.cpp -lhsa-runtime64 -ldl
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <hsa/hsa.h>
#include <hsa/hsa_ext_amd.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>

static double now_us() {
  return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count();
}
static void dump(const char *label) {
  auto f = (void (*)(const char *))dlsym(RTLD_DEFAULT, "ioctl_timer_dump");
  if (f) f(label);
}
static hsa_agent_t g_gpu;
static hsa_amd_memory_pool_t g_pool;
static bool g_hg = false, g_hp = false;
int main(int argc, char **argv) {
  const int total = argc > 1 ? atoi(argv[1]) : 32768;
  const bool map_too = argc > 2 && argv[2][0] == 'm'; // also map + grant every handle (live mappings instead of bare handles)
  const hsa_amd_memory_type_t mtype = (argc > 3 && argv[3][0] == 'n') ? MEMORY_TYPE_NONE : MEMORY_TYPE_PINNED;
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
  if (map_too && hsa_amd_vmem_address_reserve_align(&va0, (size_t)total * PAGE, 0, PAGE, 0) != HSA_STATUS_SUCCESS) return 2;
  hsa_amd_memory_access_desc_t acc{HSA_ACCESS_PERMISSION_RW, g_gpu};
  std::vector<hsa_amd_vmem_alloc_handle_t> h(total);
  dump("setup");
  for (int base = 0; base < total; base += 4096) {
    double tc = 0, tm = 0;
    for (int i = base; i < base + 4096 && i < total; i++) {
      double a = now_us();
      if (hsa_amd_vmem_handle_create(g_pool, PAGE, mtype, 0, &h[i]) != HSA_STATUS_SUCCESS) { printf("create failed at %d\n", i); return 3; }
      double b = now_us();
      tc += b - a;
      if (map_too) {
        hsa_amd_vmem_map((char *)va0 + (size_t)i * PAGE, PAGE, 0, h[i], 0);
        hsa_amd_vmem_set_access((char *)va0 + (size_t)i * PAGE, PAGE, &acc, 1);
        tm += now_us() - b;
      }
    }
    printf("live %6d..%6d: create %.2f us each%s", base, base + 4095, tc / 4096, map_too ? "" : "\n");
    if (map_too) printf(", map+access %.2f us each\n", tm / 4096);
    fflush(stdout);
    char label[64];
    snprintf(label, sizeof label, "creates with %d..%d live", base, base + 4095);
    dump(label);
  }
  double r0 = now_us();
  if (map_too) for (int i = 0; i < total; i++) hsa_amd_vmem_unmap((char *)va0 + (size_t)i * PAGE, PAGE);
  double r1 = now_us();
  for (int i = 0; i < total; i++) hsa_amd_vmem_handle_release(h[i]);
  printf("teardown: unmap %.2f us each, release %.2f us each (oldest first)\n", map_too ? (r1 - r0) / total : 0.0, (now_us() - r1) / total);
  dump("release");
  return 0;
}
