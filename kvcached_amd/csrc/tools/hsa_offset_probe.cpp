// hsa_offset_probe.cpp — HIP refuses hipMemMap with a non-zero offset into a handle. Does ROCr's
// hsa_amd_vmem_map(va, size, in_offset, handle) accept one? If it does, physical memory could be created in large
// chunks (creation is O(live handles) and 25-70 us at scale) and handed out in 2 MiB pieces.
// build: hipcc --offload-arch=gfx950 -O2 -o hsa_offset_probe hsa_offset_probe.cpp -lhsa-runtime64
#include <hip/hip_runtime.h>
#include <hsa/hsa.h>
#include <hsa/hsa_ext_amd.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>

static double now_us() {
  return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count();
}
static hsa_agent_t g_gpu;
static hsa_amd_memory_pool_t g_pool;
static bool g_hg = false, g_hp = false;
__global__ void fill32(unsigned *p, size_t n, unsigned v) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = v;
}
__global__ void count_ne(const unsigned *p, size_t n, unsigned want, unsigned long long *out) {
  unsigned long long c = 0;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) c += p[i] != want;
  if (c) atomicAdd(out, c);
}
static const char *S(hsa_status_t s) {
  const char *m = "?";
  hsa_status_string(s, &m);
  return m;
}
int main() {
  const size_t PAGE = 2u << 20, CHUNK = 64u << 20;
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
  hsa_amd_vmem_address_reserve_align(&va0, 64 * PAGE, 0, PAGE, 0);
  char *va = (char *)va0;
  hsa_amd_vmem_alloc_handle_t big{};
  hsa_status_t st = hsa_amd_vmem_handle_create(g_pool, CHUNK, MEMORY_TYPE_PINNED, 0, &big);
  printf("create 64 MiB handle: %s\n", S(st));
  hsa_amd_memory_access_desc_t acc{HSA_ACCESS_PERMISSION_RW, g_gpu};
  // piece k of the chunk at slot (31 - k): scattered, reversed
  int ok_maps = 0;
  double t0 = now_us();
  for (int k = 0; k < 32; k++) {
    st = hsa_amd_vmem_map(va + (size_t)(31 - k) * PAGE, PAGE, (size_t)k * PAGE, big, 0);
    if (st != HSA_STATUS_SUCCESS) { printf("map piece %d at offset %zu MiB: %s\n", k, (size_t)k * 2, S(st)); break; }
    st = hsa_amd_vmem_set_access(va + (size_t)(31 - k) * PAGE, PAGE, &acc, 1);
    if (st != HSA_STATUS_SUCCESS) { printf("set_access piece %d: %s\n", k, S(st)); break; }
    ok_maps++;
  }
  printf("mapped %d of 32 pieces with in_offset, %.2f us per piece (map+access)\n", ok_maps, (now_us() - t0) / (ok_maps ? ok_maps : 1));
  if (ok_maps == 32) {
    unsigned long long *cnt; hipMalloc(&cnt, 8);
    // each slot gets its own stamp; then verify through a second, linear mapping of the whole chunk that piece k holds slot (31-k)'s stamp
    for (int s = 0; s < 32; s++) fill32<<<64, 256>>>((unsigned *)(va + (size_t)s * PAGE), PAGE / 4, 0x100u + s);
    hipDeviceSynchronize();
    st = hsa_amd_vmem_map(va + 32 * PAGE, CHUNK, 0, big, 0);
    printf("second mapping of the whole chunk: %s\n", S(st));
    if (st == HSA_STATUS_SUCCESS) {
      hsa_amd_vmem_set_access(va + 32 * PAGE, CHUNK, &acc, 1);
      unsigned long long bad_total = 0;
      for (int k = 0; k < 32; k++) {
        hipMemset(cnt, 0, 8);
        count_ne<<<64, 256>>>((const unsigned *)(va + 32 * PAGE + (size_t)k * PAGE), PAGE / 4, 0x100u + (31 - k), cnt);
        unsigned long long bad = 0; hipMemcpy(&bad, cnt, 8, hipMemcpyDeviceToHost); bad_total += bad;
      }
      printf("pieces land where their offsets say: %llu wrong words\n", bad_total);
      hsa_amd_vmem_unmap(va + 32 * PAGE, CHUNK);
    }
    // unmap single pieces, remap one of them elsewhere
    double t1 = now_us();
    int ok_un = 0;
    for (int s = 0; s < 32; s += 2) { st = hsa_amd_vmem_unmap(va + (size_t)s * PAGE, PAGE); if (st == HSA_STATUS_SUCCESS) ok_un++; else { printf("unmap piece: %s\n", S(st)); break; } }
    printf("unmapped %d single pieces, %.2f us each\n", ok_un, (now_us() - t1) / (ok_un ? ok_un : 1));
    st = hsa_amd_vmem_map(va + 40 * PAGE, PAGE, 5 * PAGE, big, 0);
    printf("re-map piece 5 at another slot: %s\n", S(st));
    if (st == HSA_STATUS_SUCCESS) {
      hsa_amd_vmem_set_access(va + 40 * PAGE, PAGE, &acc, 1);
      hipMemset(cnt, 0, 8);
      count_ne<<<64, 256>>>((const unsigned *)(va + 40 * PAGE), PAGE / 4, 0x100u + (31 - 5), cnt);
      unsigned long long bad = 0; hipMemcpy(&bad, cnt, 8, hipMemcpyDeviceToHost);
      printf("  it still holds its data: %llu wrong words\n", bad);
      hsa_amd_vmem_unmap(va + 40 * PAGE, PAGE);
    }
    for (int s = 1; s < 32; s += 2) hsa_amd_vmem_unmap(va + (size_t)s * PAGE, PAGE);
  }
  st = hsa_amd_vmem_handle_release(big);
  printf("release: %s\n", S(st));
  return 0;
}
