// hsa_vmm_probe.cpp — hipMemUnmap spends ~10 of its 12-15 us spinning on an HSA signal (a marker round trip through
// the GPU queue, tools/unmap_trace.cpp); the driver work itself is 2.7 us of ioctls. What does the VMM path cost
// when it talks to ROCr directly (hsa_amd_vmem_*), and what still works in HIP with memory HIP has never seen?
//   1. per-call cost of handle_create / map / set_access / unmap / handle_release, 2 MiB pages;
//   2. a HIP kernel writes/reads such memory (raw pointers in kernel arguments);
//   3. hipMemcpy / hipMemset / hipPointerGetAttributes on such a pointer;
//   4. stale translations: is the explicit TLB invalidation still needed?
// build: hipcc --offload-arch=gfx950 -O2 -o hsa_vmm_probe hsa_vmm_probe.cpp -lhsa-runtime64
#include <hip/hip_runtime.h>
#include <hsa/hsa.h>
#include <hsa/hsa_ext_amd.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x)                                                                                                          \
  do {                                                                                                                 \
    hipError_t e_ = (x);                                                                                               \
    if (e_ != hipSuccess) {                                                                                            \
      fprintf(stderr, "%s:%d %s -> %s\n", __FILE__, __LINE__, #x, hipGetErrorString(e_));                              \
      exit(1);                                                                                                         \
    }                                                                                                                  \
  } while (0)
#define HK(x)                                                                                                          \
  do {                                                                                                                 \
    hsa_status_t s_ = (x);                                                                                             \
    if (s_ != HSA_STATUS_SUCCESS) {                                                                                    \
      const char *m = "?";                                                                                             \
      hsa_status_string(s_, &m);                                                                                       \
      fprintf(stderr, "%s:%d %s -> 0x%x %s\n", __FILE__, __LINE__, #x, (unsigned)s_, m);                               \
      exit(1);                                                                                                         \
    }                                                                                                                  \
  } while (0)

static double now_us() {
  return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

static hsa_agent_t g_gpu, g_cpu;
static bool g_have_cpu = false;
static hsa_amd_memory_pool_t g_pool;
static bool g_have_gpu = false, g_have_pool = false;

static hsa_status_t on_agent(hsa_agent_t a, void *) {
  hsa_device_type_t t;
  hsa_agent_get_info(a, HSA_AGENT_INFO_DEVICE, &t);
  if (t == HSA_DEVICE_TYPE_GPU && !g_have_gpu) {
    g_gpu = a;
    g_have_gpu = true;
  }
  if (t == HSA_DEVICE_TYPE_CPU && !g_have_cpu) {
    g_cpu = a;
    g_have_cpu = true;
  }
  return HSA_STATUS_SUCCESS;
}
static hsa_status_t on_pool(hsa_amd_memory_pool_t p, void *) {
  hsa_amd_segment_t seg;
  hsa_amd_memory_pool_get_info(p, HSA_AMD_MEMORY_POOL_INFO_SEGMENT, &seg);
  uint32_t flags = 0;
  hsa_amd_memory_pool_get_info(p, HSA_AMD_MEMORY_POOL_INFO_GLOBAL_FLAGS, &flags);
  bool alloc = false;
  hsa_amd_memory_pool_get_info(p, HSA_AMD_MEMORY_POOL_INFO_RUNTIME_ALLOC_ALLOWED, &alloc);
  if (seg == HSA_AMD_SEGMENT_GLOBAL && alloc && (flags & HSA_AMD_MEMORY_POOL_GLOBAL_FLAG_COARSE_GRAINED) && !g_have_pool) {
    g_pool = p;
    g_have_pool = true;
  }
  return HSA_STATUS_SUCCESS;
}

__global__ void fill32(unsigned *p, size_t n, unsigned v) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = v;
}
__global__ void count_ne(const unsigned *p, size_t n, unsigned want, unsigned long long *out) {
  unsigned long long c = 0;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) c += p[i] != want;
  if (c) atomicAdd(out, c);
}

int main(int argc, char **argv) {
  const int n = argc > 1 ? atoi(argv[1]) : 1024;
  const bool hip_reserve = argc > 2 && argv[2][0] == 'h' && argv[2][1] == 'i'; // VA range reserved through HIP (so HIP knows the range)
  const int op = argc > 3 ? atoi(argv[3]) : -1;
  const bool cpu_access = argc > 4; // also grant the CPU agent access: is the pointer then a usable host pointer for HIP's fallback?                                 // which HIP call to try on the HSA-mapped pointer (each may crash)
  const size_t PAGE = 2u << 20;
  CK(hipSetDevice(0));
  CK(hipFree(nullptr)); // HIP has initialised ROCr; hsa_init only adds a reference
  HK(hsa_init());
  HK(hsa_iterate_agents(on_agent, nullptr));
  HK(hsa_amd_agent_iterate_memory_pools(g_gpu, on_pool, nullptr));
  if (!g_have_gpu || !g_have_pool) {
    fprintf(stderr, "no GPU agent / coarse-grained pool\n");
    return 1;
  }
  size_t gran = 0;
  hsa_amd_memory_pool_get_info(g_pool, HSA_AMD_MEMORY_POOL_INFO_RUNTIME_ALLOC_GRANULE, &gran);
  printf("pool alloc granule %zu\n", gran);

  void *va0 = nullptr;
  double t0 = now_us();
  if (hip_reserve)
    CK(hipMemAddressReserve(&va0, (size_t)n * PAGE, PAGE, nullptr, 0));
  else
    HK(hsa_amd_vmem_address_reserve_align(&va0, (size_t)n * PAGE, 0, PAGE, 0));
  printf("reserve %d x 2 MiB: %.1f us -> %p\n", n, now_us() - t0, va0);
  char *va = (char *)va0;
  std::vector<hsa_amd_vmem_alloc_handle_t> h(n);
  hsa_amd_memory_access_desc_t acc{HSA_ACCESS_PERMISSION_RW, g_gpu};
  hsa_amd_memory_access_desc_t both[2] = {{HSA_ACCESS_PERMISSION_RW, g_gpu}, {HSA_ACCESS_PERMISSION_RW, g_cpu}};
  unsigned long long *cnt;
  CK(hipMalloc(&cnt, 8));
  hipStream_t s;
  CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));

  for (int round = 0; round < (op < 0 || op == 99 ? 3 : 0); round++) {
    double a = now_us();
    if (round == 0)
      for (int i = 0; i < n; i++) HK(hsa_amd_vmem_handle_create(g_pool, PAGE, MEMORY_TYPE_PINNED, 0, &h[i]));
    double b = now_us();
    for (int i = 0; i < n; i++) HK(hsa_amd_vmem_map(va + (size_t)i * PAGE, PAGE, 0, h[(i + round * 7) % n], 0));
    double c = now_us();
    for (int i = 0; i < n; i++) HK(hsa_amd_vmem_set_access(va + (size_t)i * PAGE, PAGE, cpu_access ? both : &acc, cpu_access ? 2 : 1));
    double d = now_us();
    // HIP kernels on memory HIP has never heard of
    const unsigned stamp = 0x1000u + round;
    fill32<<<2048, 256, 0, s>>>((unsigned *)va, (size_t)n * PAGE / 4, stamp);
    CK(hipMemsetAsync(cnt, 0, 8, s));
    count_ne<<<2048, 256, 0, s>>>((const unsigned *)va, (size_t)n * PAGE / 4, stamp, cnt);
    unsigned long long bad = ~0ull;
    CK(hipMemcpyAsync(&bad, cnt, 8, hipMemcpyDeviceToHost, s));
    CK(hipStreamSynchronize(s));
    double e = now_us();
    for (int i = 0; i < n; i++) HK(hsa_amd_vmem_unmap(va + (size_t)i * PAGE, PAGE));
    double f = now_us();
    printf("round %d: create %.2f  map %.2f  set_access %.2f  unmap %.2f us/page;  kernel check: %llu words differ\n", round,
           round == 0 ? (b - a) / n : 0.0, (c - b) / n, (d - c) / n, (f - e) / n, bad);
    fflush(stdout);
  }

  // what HIP's copy/memset/query paths do with such a pointer
  if (op >= 0 && op != 99)
    for (int i = 0; i < n; i++) HK(hsa_amd_vmem_handle_create(g_pool, PAGE, MEMORY_TYPE_PINNED, 0, &h[i]));
  HK(hsa_amd_vmem_map(va, PAGE, 0, h[0], 0));
  if (cpu_access) {
    double ta = now_us();
    hsa_status_t sa = hsa_amd_vmem_set_access(va, PAGE, both, 2);
    printf("set_access(GPU+CPU): 0x%x in %.1f us\n", (unsigned)sa, now_us() - ta);
    if (sa != HSA_STATUS_SUCCESS) HK(hsa_amd_vmem_set_access(va, PAGE, &acc, 1));
  } else {
    HK(hsa_amd_vmem_set_access(va, PAGE, &acc, 1));
  }
  fill32<<<256, 256, 0, s>>>((unsigned *)va, PAGE / 4, 0xabcd1234u);
  CK(hipStreamSynchronize(s));
  unsigned host[4] = {0, 0, 0, 0};
  unsigned src[4] = {7, 7, 7, 7};
  hipError_t st = hipSuccess;
  hipPointerAttribute_t at{};
  void *dbuf;
  CK(hipMalloc(&dbuf, 4096));
  printf("reserve=%s op=%d: ", hip_reserve ? "hip" : "hsa", op);
  fflush(stdout);
  switch (op) {
  case 0: st = hipMemcpy(host, va, sizeof host, hipMemcpyDeviceToHost); printf("hipMemcpy D2H: %s, read 0x%x\n", hipGetErrorString(st), host[0]); break;
  case 1: st = hipMemcpy(host, va, sizeof host, hipMemcpyDefault); printf("hipMemcpy Default (to host): %s, read 0x%x\n", hipGetErrorString(st), host[0]); break;
  case 2: st = hipMemcpy(va, src, sizeof src, hipMemcpyHostToDevice); printf("hipMemcpy H2D: %s\n", hipGetErrorString(st)); break;
  case 3: st = hipMemset(va, 0, 4096); printf("hipMemset: %s\n", hipGetErrorString(st)); break;
  case 4: st = hipPointerGetAttributes(&at, va); printf("hipPointerGetAttributes: %s (type %d device %d)\n", hipGetErrorString(st), (int)at.type, at.device); break;
  case 5: st = hipMemcpy(dbuf, va, 4096, hipMemcpyDeviceToDevice); printf("hipMemcpy D2D out of it: %s\n", hipGetErrorString(st)); break;
  case 6: st = hipMemcpyAsync(host, va, sizeof host, hipMemcpyDeviceToHost, s); CK(hipStreamSynchronize(s)); printf("hipMemcpyAsync D2H: %s, read 0x%x\n", hipGetErrorString(st), host[0]); break;
  case 7: { // coherence of HIP's host-side fallback: GPU rewrites, host re-reads the same words, many times
    int stale = 0;
    for (unsigned it = 1; it <= 200; ++it) {
      fill32<<<256, 256, 0, s>>>((unsigned *)va, PAGE / 4, 0x5000u + it);
      CK(hipStreamSynchronize(s));
      CK(hipMemcpy(host, va + (it % 7) * 4096, sizeof host, hipMemcpyDeviceToHost));
      if (host[0] != 0x5000u + it) ++stale;
      unsigned w[4] = {it, it, it, it}; // and the other way round: host writes, a kernel reads
      CK(hipMemcpy(va + 8192, w, sizeof w, hipMemcpyHostToDevice));
      CK(hipMemsetAsync(cnt, 0, 8, s));
      count_ne<<<1, 4, 0, s>>>((const unsigned *)(va + 8192), 4, it, cnt);
      unsigned long long bad = 0;
      CK(hipMemcpyAsync(&bad, cnt, 8, hipMemcpyDeviceToHost, s));
      CK(hipStreamSynchronize(s));
      if (bad) ++stale;
    }
    printf("coherence: %d stale observations in 200 rewrite/re-read rounds\n", stale);
    std::vector<char> big(64u << 20);
    for (int k = 0; k < 32; k++) { HK(hsa_amd_vmem_map(va + (size_t)(k + 1) * PAGE, PAGE, 0, h[k + 1], 0)); HK(hsa_amd_vmem_set_access(va + (size_t)(k + 1) * PAGE, PAGE, both, 2)); }
    double ta = now_us();
    CK(hipMemcpy(big.data(), va + PAGE, 32 * PAGE, hipMemcpyDeviceToHost));
    double tb = now_us();
    CK(hipMemcpy(va + PAGE, big.data(), 32 * PAGE, hipMemcpyHostToDevice));
    double tc = now_us();
    printf("64 MiB through HIP's host fallback: D2H %.1f MB/s, H2D %.1f MB/s\n", 64.0 * 1.048576 / ((tb - ta) * 1e-6), 64.0 * 1.048576 / ((tc - tb) * 1e-6));
    for (int k = 0; k < 32; k++) HK(hsa_amd_vmem_unmap(va + (size_t)(k + 1) * PAGE, PAGE));
    break;
  }
  case 8: { // "registered with HIP once, backed through ROCr": does HIP keep serving copies for a VA whose mapping
            // it created, after that mapping was replaced behind its back?
    HK(hsa_amd_vmem_unmap(va, PAGE)); // the op prologue mapped h[0] through ROCr: start from an empty slot
    void *hva = nullptr;
    CK(hipMemAddressReserve(&hva, 4 * PAGE, PAGE, nullptr, 0));
    char *q = (char *)hva;
    hipMemAllocationProp prop{};
    prop.type = hipMemAllocationTypePinned;
    prop.location.type = hipMemLocationTypeDevice;
    hipMemGenericAllocationHandle_t shell;
    CK(hipMemCreate(&shell, PAGE, &prop, 0));
    for (int k = 0; k < 4; k++) CK(hipMemMap(q + (size_t)k * PAGE, PAGE, 0, shell, 0)); // HIP now knows these 4 slots
    for (int k = 0; k < 4; k++) HK(hsa_amd_vmem_unmap(q + (size_t)k * PAGE, PAGE));      // ... and we take them away
    printf("registered 4 slots with HIP, unmapped them through ROCr\n");
    for (int round = 0; round < 3; round++) {
      for (int k = 0; k < 4; k++) {
        HK(hsa_amd_vmem_map(q + (size_t)k * PAGE, PAGE, 0, h[(round * 4 + k) % n], 0));
        HK(hsa_amd_vmem_set_access(q + (size_t)k * PAGE, PAGE, &acc, 1));
      }
      const unsigned stamp = 0x7700u + round;
      fill32<<<512, 256, 0, s>>>((unsigned *)q, 4 * PAGE / 4, stamp);
      CK(hipStreamSynchronize(s));
      std::vector<unsigned> back(4 * PAGE / 4, 0);
      double ta = now_us();
      st = hipMemcpy(back.data(), q, 4 * PAGE, hipMemcpyDeviceToHost);
      double tb = now_us();
      size_t wrong = 0;
      for (unsigned w : back) wrong += w != stamp;
      printf("round %d: hipMemcpy D2H of 8 MiB: %s, %.1f MB/s, %zu words wrong\n", round, hipGetErrorString(st),
             8.0 * 1.048576 / ((tb - ta) * 1e-6), wrong);
      for (auto &w : back) w = 0x9900u + round;
      st = hipMemcpyAsync(q, back.data(), 4 * PAGE, hipMemcpyHostToDevice, s);
      CK(hipStreamSynchronize(s));
      CK(hipMemsetAsync(cnt, 0, 8, s));
      count_ne<<<512, 256, 0, s>>>((const unsigned *)q, 4 * PAGE / 4, 0x9900u + round, cnt);
      unsigned long long bad = ~0ull;
      CK(hipMemcpyAsync(&bad, cnt, 8, hipMemcpyDeviceToHost, s));
      CK(hipStreamSynchronize(s));
      printf("         hipMemcpyAsync H2D: %s, kernel sees %llu wrong words\n", hipGetErrorString(st), bad);
      st = hipMemcpy(dbuf, q + PAGE, 4096, hipMemcpyDeviceToDevice);
      printf("         D2D: %s\n", hipGetErrorString(st));
      double tu = now_us();
      for (int k = 0; k < 4; k++) HK(hsa_amd_vmem_unmap(q + (size_t)k * PAGE, PAGE));
      printf("         ROCr unmap %.2f us/slot\n", (now_us() - tu) / 4);
    }
    // teardown the way HIP expects it: something must be mapped where HIP believes its mapping is
    for (int k = 0; k < 4; k++) HK(hsa_amd_vmem_map(q + (size_t)k * PAGE, PAGE, 0, h[k], 0));
    for (int k = 0; k < 4; k++) {
      st = hipMemUnmap(q + (size_t)k * PAGE, PAGE);
      if (st != hipSuccess) printf("teardown hipMemUnmap slot %d: %s\n", k, hipGetErrorString(st));
    }
    st = hipMemRelease(shell);
    printf("teardown: release %s, address free %s\n", hipGetErrorString(st), hipGetErrorString(hipMemAddressFree(hva, 4 * PAGE)));
    HK(hsa_amd_vmem_map(va, PAGE, 0, h[0], 0)); // restore what the epilogue expects
    break;
  }
  default: printf("no op\n");
  }
  fflush(stdout);
  (void)hipGetLastError();
  HK(hsa_amd_vmem_unmap(va, PAGE));

  // release
  double r0 = now_us();
  for (int i = 0; i < n; i++) HK(hsa_amd_vmem_handle_release(h[i]));
  printf("handle_release %.2f us/page\n", (now_us() - r0) / n);
  if (hip_reserve)
    CK(hipMemAddressFree(va0, (size_t)n * PAGE));
  else
    HK(hsa_amd_vmem_address_free(va0, (size_t)n * PAGE));
  printf("done\n");
  return 0;
}
