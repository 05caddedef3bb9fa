// remap_diag: does a VA slot that aliased a shared "zero page" translate to its NEW physical page
// immediately after hipMemUnmap + hipMemMap + hipMemSetAccess? Pure HIP, no torch.
// Build: hipcc --offload-arch=gfx950 -O2 -std=c++17 remap_diag.cpp -o remap_diag
#include <hip/hip_runtime.h>
#include <hsa/hsa.h>
#include <hsa/hsa_ext_amd.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(e)                                                                                      \
  do {                                                                                             \
    hipError_t s_ = (e);                                                                           \
    if (s_ != hipSuccess) {                                                                        \
      printf("FAIL %s -> %s (line %d)\n", #e, hipGetErrorString(s_), __LINE__);                    \
      exit(1);                                                                                     \
    }                                                                                              \
  } while (0)

static const size_t PAGE = 2u << 20;

__global__ void fill32(unsigned *p, size_t n, unsigned v) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x, st = (size_t)gridDim.x * blockDim.x;
  for (; i < n; i += st) p[i] = v;
}
__global__ void count_ne(const unsigned *p, size_t n, unsigned v, unsigned long long *out, unsigned *sample) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x, st = (size_t)gridDim.x * blockDim.x;
  unsigned long long c = 0;
  for (; i < n; i += st)
    if (p[i] != v) {
      ++c;
      *sample = p[i];
    }
  if (c) atomicAdd(out, c);
}

#include <chrono>
#include <thread>
// candidate ways of making the driver invalidate the GPU TLBs after VMM map/unmap calls
static void flush_trigger(int mode) {
  auto t0 = std::chrono::steady_clock::now();
  void *p = nullptr;
  switch (mode) {
  case 3: CK(hipMalloc(&p, 64u << 20)); CK(hipFree(p)); break;          // big device alloc + free
  case 4: CK(hipMalloc(&p, 4096)); CK(hipFree(p)); break;               // small (sub-allocated?)
  case 5: CK(hipHostMalloc(&p, 2u << 20, 0)); CK(hipHostFree(p)); break; // pinned host alloc + free
  case 6: std::this_thread::sleep_for(std::chrono::milliseconds(200)); break;
  case 7: {                                                             // a throw-away VMM page elsewhere
    hipMemAllocationProp prop{};
    prop.type = hipMemAllocationTypePinned;
    prop.location.type = hipMemLocationTypeDevice;
    hipMemGenericAllocationHandle_t h;
    void *va = nullptr;
    hipMemAccessDesc acc{};
    acc.location = prop.location;
    acc.flags = hipMemAccessFlagsProtReadWrite;
    CK(hipMemAddressReserve(&va, 2u << 20, 2u << 20, nullptr, 0));
    CK(hipMemCreate(&h, 2u << 20, &prop, 0));
    CK(hipMemMap(va, 2u << 20, 0, h, 0));
    CK(hipMemSetAccess(va, 2u << 20, &acc, 1));
    CK(hipMemUnmap(va, 2u << 20));
    CK(hipMemRelease(h));
    CK(hipMemAddressFree(va, 2u << 20));
    break;
  }
  case 8: CK(hipMalloc(&p, 2u << 20)); CK(hipFree(p)); break;           // 2 MiB device alloc + free
  case 9: CK(hipMalloc(&p, 64u << 20)); break;                          // alloc only (leaks; map ioctl only)
  case 10: {                                                            // register + unregister one host page
    static char *page = (char *)aligned_alloc(4096, 4096);
    CK(hipHostRegister(page, 4096, hipHostRegisterDefault));
    CK(hipHostUnregister(page));
    break;
  }
  case 11: {                                                            // register only (fresh page each time)
    static char *buf = (char *)aligned_alloc(4096, 4096 * 64);
    static int k = 0;
    buf[4096 * k] = 1;
    CK(hipHostRegister(buf + 4096 * k++, 4096, hipHostRegisterDefault));
    break;
  }
  case 12: {                                                            // unregister only (registered up front)
    static char *buf = nullptr;
    static int k = 0;
    if (!buf) {
      buf = (char *)aligned_alloc(4096, 4096 * 64);
      for (int i = 0; i < 64; i++) { buf[4096 * i] = 1; CK(hipHostRegister(buf + 4096 * i, 4096, hipHostRegisterDefault)); }
      t0 = std::chrono::steady_clock::now();
    }
    CK(hipHostUnregister(buf + 4096 * k++));
    break;
  }
  case 13: CK(hipMalloc(&p, 2u << 20)); break;                          // 2 MiB alloc only (leaks)
  case 14: {                                                            // free only (allocated up front)
    static std::vector<void *> blocks;
    if (blocks.empty()) {
      for (int i = 0; i < 64; i++) { void *q; CK(hipMalloc(&q, 2u << 20)); blocks.push_back(q); }
      t0 = std::chrono::steady_clock::now();
    }
    CK(hipFree(blocks.back()));
    blocks.pop_back();
    break;
  }
  case 15: CK(hipHostMalloc(&p, 4096, 0)); CK(hipHostFree(p)); break;   // small pinned host alloc + free
  case 16: CK(hipExtMallocWithFlags(&p, 2u << 20, hipDeviceMallocFinegrained)); CK(hipFree(p)); break;
  case 17: CK(hipExtMallocWithFlags(&p, 2u << 20, hipDeviceMallocUncached)); CK(hipFree(p)); break;
  case 18: {                                                            // managed range migrated back and forth
    static char *m = nullptr;
    static int flip = 0;
    if (!m) { CK(hipMallocManaged((void **)&m, 2u << 20, hipMemAttachGlobal)); m[0] = 1; t0 = std::chrono::steady_clock::now(); }
    CK(hipMemPrefetchAsync(m, 2u << 20, (flip++ & 1) ? hipCpuDeviceId : 0, 0));
    CK(hipStreamSynchronize(0));
    break;
  }
  case 21:   // ROCr: re-grant the GPU access to a small, persistent host-pool buffer (a KFD map ioctl every time?)
  case 22: { // ... or to a small device-pool buffer
    static hsa_agent_t gpu, cpu;
    static void *buf = nullptr;
    if (!buf) {
      hsa_init();
      struct F { hsa_agent_t g, c; bool hg = false, hc = false; } f;
      hsa_iterate_agents([](hsa_agent_t a, void *q) -> hsa_status_t {
        auto *f = (F *)q; hsa_device_type_t t; hsa_agent_get_info(a, HSA_AGENT_INFO_DEVICE, &t);
        if (t == HSA_DEVICE_TYPE_GPU && !f->hg) { f->g = a; f->hg = true; }
        if (t == HSA_DEVICE_TYPE_CPU && !f->hc) { f->c = a; f->hc = true; }
        return HSA_STATUS_SUCCESS; }, &f);
      gpu = f.g; cpu = f.c;
      struct P { hsa_amd_memory_pool_t p; bool found = false; bool want_fine; } pp;
      pp.want_fine = mode == 21;
      hsa_amd_agent_iterate_memory_pools(mode == 21 ? cpu : gpu, [](hsa_amd_memory_pool_t p, void *q) -> hsa_status_t {
        auto *o = (P *)q; hsa_amd_segment_t seg; bool alloc = false; uint32_t fl = 0;
        hsa_amd_memory_pool_get_info(p, HSA_AMD_MEMORY_POOL_INFO_SEGMENT, &seg);
        hsa_amd_memory_pool_get_info(p, HSA_AMD_MEMORY_POOL_INFO_RUNTIME_ALLOC_ALLOWED, &alloc);
        hsa_amd_memory_pool_get_info(p, HSA_AMD_MEMORY_POOL_INFO_GLOBAL_FLAGS, &fl);
        if (seg == HSA_AMD_SEGMENT_GLOBAL && alloc && !o->found) { o->p = p; o->found = true; }
        return HSA_STATUS_SUCCESS; }, &pp);
      if (hsa_amd_memory_pool_allocate(pp.p, 4096, 0, &buf) != HSA_STATUS_SUCCESS) { printf("pool allocate failed\n"); exit(1); }
      t0 = std::chrono::steady_clock::now();
    }
    if (hsa_amd_agents_allow_access(1, &gpu, nullptr, buf) != HSA_STATUS_SUCCESS) printf("allow_access failed\n");
    break;
  }
  case 19: CK(hipMalloc(&p, 1u << 20)); CK(hipFree(p)); break;          // 1 MiB
  case 20: CK(hipMalloc(&p, 256u << 10)); CK(hipFree(p)); break;        // 256 KiB
  default: return;
  }
  double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
  static int n = 0;
  if (n++ < 4) printf("  [flush_trigger mode %d took %.1f us]\n", mode, us);
}

struct Ctx {
  hipStream_t s1, s2;
  unsigned long long *cnt;
  unsigned *sample;
};

static unsigned long long check(Ctx &c, char *va, unsigned want, hipStream_t s, unsigned *sample_out) {
  CK(hipMemsetAsync(c.cnt, 0, 8, s));
  CK(hipMemsetAsync(c.sample, 0, 4, s));
  count_ne<<<256, 256, 0, s>>>((const unsigned *)va, PAGE / 4, want, c.cnt, c.sample);
  unsigned long long h = 0;
  CK(hipMemcpyAsync(&h, c.cnt, 8, hipMemcpyDeviceToHost, s));
  CK(hipMemcpyAsync(sample_out, c.sample, 4, hipMemcpyDeviceToHost, s));
  CK(hipStreamSynchronize(s));
  return h;
}

int main(int argc, char **argv) {
  int mode = argc > 1 ? atoi(argv[1]) : 0;
  bool verbose = argc > 2; // 0 plain; 1 = hipDeviceSynchronize after remap; 2 = no touching of aliases first
  CK(hipSetDevice(0));
  Ctx c;
  CK(hipStreamCreateWithFlags(&c.s1, hipStreamNonBlocking));
  CK(hipStreamCreateWithFlags(&c.s2, hipStreamNonBlocking));
  CK(hipMalloc(&c.cnt, 8));
  CK(hipMalloc(&c.sample, 4));
  hipMemAllocationProp prop{};
  prop.type = hipMemAllocationTypePinned;
  prop.location.type = hipMemLocationTypeDevice;
  prop.location.id = 0;
  hipMemAccessDesc acc{};
  acc.location = prop.location;
  acc.flags = hipMemAccessFlagsProtReadWrite;

  const int SLOTS = 64;
  char *va = nullptr;
  CK(hipMemAddressReserve((void **)&va, SLOTS * PAGE, PAGE, (void *)0x1f000000000ull, 0));
  hipMemGenericAllocationHandle_t zero;
  CK(hipMemCreate(&zero, PAGE, &prop, 0));
  for (int i = 0; i < SLOTS; i++) CK(hipMemMap(va + i * PAGE, PAGE, 0, zero, 0));
  CK(hipMemSetAccess(va, SLOTS * PAGE, &acc, 1));
  fill32<<<256, 256, 0, c.s1>>>((unsigned *)va, PAGE / 4, 0u); // zero page <- 0 through alias 0
  CK(hipStreamSynchronize(c.s1));
  unsigned smp = 0;
  if (mode != 2) {
    unsigned long long bad = 0;
    for (int i = 0; i < SLOTS; i++) bad += check(c, va + i * PAGE, 0u, c.s2, &smp); // touch every alias
    printf("mode %d: all aliases read 0: bad=%llu\n", mode, bad);
  }

  int anomalies = 0;
  for (int round = 0; round < 3; round++) {
    std::vector<int> slots = {0, 3, 7, 16, 19, 23, 40, 41, 42};
    std::vector<hipMemGenericAllocationHandle_t> hs(slots.size());
    // remap: alias -> private page, zero it on s1 (like the allocator)
    for (size_t k = 0; k < slots.size(); k++) {
      char *p = va + slots[k] * PAGE;
      CK(hipMemUnmap(p, PAGE));
      CK(hipMemCreate(&hs[k], PAGE, &prop, 0));
      CK(hipMemMap(p, PAGE, 0, hs[k], 0));
      CK(hipMemSetAccess(p, PAGE, &acc, 1));
      if (mode < 3) fill32<<<256, 256, 0, c.s1>>>((unsigned *)p, PAGE / 4, 0u);
    }
    CK(hipStreamSynchronize(c.s1));
    if (mode == 1) CK(hipDeviceSynchronize());
    flush_trigger(mode);
    // like the test: per slot, read (expect 0) then write a unique value, on another stream
    for (size_t k = 0; k < slots.size(); k++) {
      char *p = va + slots[k] * PAGE;
      unsigned long long bad = check(c, p, 0u, c.s2, &smp);
      if (bad) {
        if (verbose) printf("  round %d slot %d: first read expected 0, %llu dwords differ (sample 0x%x)\n", round, slots[k], bad, smp);
        anomalies++;
        unsigned long long again = check(c, p, 0u, c.s2, &smp);
        if (verbose) printf("     re-read: %llu differ\n", again);
      }
      fill32<<<256, 256, 0, c.s2>>>((unsigned *)p, PAGE / 4, 0x1000u + slots[k]);
      CK(hipStreamSynchronize(c.s2));
    }
    for (size_t k = 0; k < slots.size(); k++) {
      unsigned long long bad = check(c, va + slots[k] * PAGE, 0x1000u + slots[k], c.s2, &smp);
      if (bad) {
        if (verbose) printf("  round %d slot %d: private value lost, %llu differ (sample 0x%x)\n", round, slots[k], bad, smp);
        anomalies++;
      }
    }
    // the zero page must still be zero (seen through untouched aliases)
    for (int s : {1, 2, 8, 63}) {
      unsigned long long bad = check(c, va + s * PAGE, 0u, c.s2, &smp);
      if (bad) {
        if (verbose) printf("  round %d alias %d: zero page polluted, %llu differ (sample 0x%x)\n", round, s, bad, smp);
        anomalies++;
      }
    }
    // back to aliases
    for (size_t k = 0; k < slots.size(); k++) {
      char *p = va + slots[k] * PAGE;
      CK(hipMemUnmap(p, PAGE));
      CK(hipMemMap(p, PAGE, 0, zero, 0));
      CK(hipMemSetAccess(p, PAGE, &acc, 1));
      CK(hipMemRelease(hs[k]));
    }
    flush_trigger(mode);
    for (size_t k = 0; k < slots.size(); k++) {
      unsigned long long bad = check(c, va + slots[k] * PAGE, 0u, c.s2, &smp);
      if (bad) {
        if (verbose) printf("  round %d slot %d: after un-backing expected zero page, %llu differ (sample 0x%x)\n", round, slots[k], bad, smp);
        anomalies++;
      }
    }
  }
  printf("mode %d: anomalies=%d\n", mode, anomalies);
  return 0;
}
