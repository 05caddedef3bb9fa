// kv_allocator.cpp — see kv_allocator.hpp.
#include "kv_allocator.hpp"
#include "run_scan.hpp"

#include <fcntl.h>
#include <sys/mman.h>
#include <unistd.h>

#include <algorithm>
#include <cassert>

#include "kernels.hpp"

namespace kvc {

namespace {
std::mutex g_mu; // guards the registry below
// Never destroyed: a process that ends without kvc_shutdown() must not tear GPU state down from static destructors -
// the HIP runtime, ROCr and libdrm are going away in an unspecified order at that point (that crashed at exit); the
// kernel reclaims mappings and memory with the process. The Python module shuts down in an atexit hook, in good time.
auto &g_allocators = *new std::unordered_map<int64_t, std::unique_ptr<KvAllocator>>;
auto &g_contexts = *new std::unordered_map<int, std::unique_ptr<GpuContext>>;
DeviceSpec g_device;
bool g_contiguous = false;
bool g_initialized = false;
size_t g_page_size = kBasePage;
std::atomic<size_t> g_vaddr_offset{0}; // running offset behind kStartAddr (reference: ftensor.cpp:17)
std::atomic<size_t> g_override_free{0}, g_override_total{0};
std::atomic<size_t> g_pending_unmap_bytes{0}; // async unmap: released by the caller, not yet by the driver

int resolve_dev_index(const DeviceSpec &d) {
  if (d.index >= 0) return d.index;
  int cur = 0;
  HIP_CHECK(hipGetDevice(&cur));
  return cur;
}

GpuContext *context_for(int dev) {
  auto it = g_contexts.find(dev);
  if (it == g_contexts.end()) it = g_contexts.emplace(dev, std::make_unique<GpuContext>(dev)).first;
  return it->second.get();
}
} // namespace

// The hybrid backend rests on one property of the HIP runtime: once it has recorded a mapping for a VA it keeps
// addressing that VA in copies and memsets, whatever ROCr has mapped there since. Checked once per init on a
// scratch page (a few hundred microseconds): placeholder mapped through HIP and removed through ROCr, a real page
// mapped through ROCr, written with hipMemset and read back with hipMemcpy. Any surprise -> the plain HIP backend.
namespace {
bool hybrid_self_test(int dev) {
  if (KVC_TEST_HOOK("FAIL_HYBRID_SELFTEST")) return false; // tests: exercise the fallback
  const size_t ps = kBasePage;
  void *va = nullptr;
  hipMemGenericAllocationHandle_t shell = nullptr;
  hsa_amd_vmem_alloc_handle_t real{};
  bool have_real = false, hip_mapped = false, rocr_mapped = false, ok = false;
  try {
    const HsaDevice &hd = hsa_device(dev);
    auto prop = make_alloc_prop(dev, false);
    if (hipMemAddressReserve(&va, ps, ps, nullptr, 0) != hipSuccess) throw 1;
    if (hipMemCreate(&shell, ps, &prop, 0) != hipSuccess) throw 2;
    if (hipMemMap(va, ps, 0, shell, 0) != hipSuccess) throw 3;
    hip_mapped = true;
    if (hsa_amd_vmem_unmap(va, ps) != HSA_STATUS_SUCCESS) throw 4;
    if (hsa_amd_vmem_handle_create(hd.pool, ps, MEMORY_TYPE_PINNED, 0, &real) != HSA_STATUS_SUCCESS) throw 5;
    have_real = true;
    if (hsa_amd_vmem_map(va, ps, 0, real, 0) != HSA_STATUS_SUCCESS) throw 6;
    rocr_mapped = true;
    hsa_amd_memory_access_desc_t d{HSA_ACCESS_PERMISSION_RW, hd.agent};
    if (hsa_amd_vmem_set_access(va, ps, &d, 1) != HSA_STATUS_SUCCESS) throw 7;
    unsigned char host[256];
    memset(host, 0, sizeof host);
    if (hipMemset(va, 0x5a, sizeof host) != hipSuccess) throw 8;
    if (hipMemcpy(host, va, sizeof host, hipMemcpyDeviceToHost) != hipSuccess) throw 9;
    for (unsigned char c : host)
      if (c != 0x5a) throw 10;
    memset(host, 0xc3, sizeof host);
    if (hipMemcpy(static_cast<char *>(va) + 4096, host, sizeof host, hipMemcpyHostToDevice) != hipSuccess) throw 11;
    unsigned char back[256];
    if (hipMemcpy(back, static_cast<char *>(va) + 4096, sizeof back, hipMemcpyDeviceToHost) != hipSuccess) throw 12;
    if (memcmp(back, host, sizeof back) != 0) throw 13;
    ok = true;
  } catch (int step) {
    KVC_LOG(LOG_WARNING, "hybrid VMM self test stopped at step %d", step);
  } catch (const std::exception &e) {
    KVC_LOG(LOG_WARNING, "hybrid VMM self test: %s", e.what());
  }
  (void)hipGetLastError();
  // teardown in the order HIP expects: something mapped at the VA, HIP unmaps it
  if (hip_mapped) {
    if (!rocr_mapped && have_real) rocr_mapped = hsa_amd_vmem_map(va, ps, 0, real, 0) == HSA_STATUS_SUCCESS;
    if (hipMemUnmap(va, ps) != hipSuccess) {
      (void)hipGetLastError();
      if (rocr_mapped) (void)hsa_amd_vmem_unmap(va, ps);
    }
  }
  { // the page was written through this VA: its translation must not outlive it (the page goes back below, and the
    // VA may be handed out again for a region whose first map no longer invalidates by itself)
    void *p = nullptr;
    if (!KVC_TEST_HOOK("SKIP_TEARDOWN_FLUSH") && hipMalloc(&p, kBasePage) == hipSuccess) (void)hipFree(p);
    (void)hipGetLastError();
  }
  if (have_real) (void)hsa_amd_vmem_handle_release(real);
  if (shell) (void)hipMemRelease(shell);
  if (va) (void)hipMemAddressFree(va, ps);
  (void)hipGetLastError();
  return ok;
}

// The drm backend edits the process's GPU page tables through libdrm_amdgpu, next to ROCr (DrmVm, hip_vmm.hpp).
// Before any page is served that way, prove on one slot that (1) the amdgpu_device libdrm handed us drives the SAME
// address space as ROCr - a GEM_VA map over a VA that ROCr has mapped must be refused, and accepted once ROCr has
// unmapped it; this step touches no memory, so a wrong VM cannot fault - and (2) data follows the handle: two pages
// swapped under one VA keep their own contents, seen through HIP's memset/copy on a HIP-registered slot.
bool drm_self_test(int dev) {
  if (KVC_TEST_HOOK("FAIL_DRM_SELFTEST")) return false; // tests: exercise the fallback
  std::string why;
  DrmVm &vm = DrmVm::instance();
  if (!vm.open(dev, &why)) {
    KVC_LOG(LOG_WARNING, "direct DRM mapping unavailable: %s", why.c_str());
    return false;
  }
  const size_t ps = kBasePage;
  void *va = nullptr;
  hipMemGenericAllocationHandle_t shell = nullptr;
  hsa_amd_vmem_alloc_handle_t a{}, b{};
  bool have_a = false, have_b = false, hip_mapped = false, rocr_mapped = false, ok = false;
  void *drm_mapped = nullptr; // the bo currently mapped at va through DRM
  auto shootdown = [&]() {
    void *p = nullptr;
    if (hipMalloc(&p, kBasePage) == hipSuccess) (void)hipFree(p);
  };
  try {
    const HsaDevice &hd = hsa_device(dev);
    auto prop = make_alloc_prop(dev, false);
    if (hipMemAddressReserve(&va, ps, ps, nullptr, 0) != hipSuccess) throw 1;
    if (hipMemCreate(&shell, ps, &prop, 0) != hipSuccess) throw 2;
    if (hipMemMap(va, ps, 0, shell, 0) != hipSuccess) throw 3;
    hip_mapped = true;
    if (hsa_amd_vmem_unmap(va, ps) != HSA_STATUS_SUCCESS) throw 4;
    if (hsa_amd_vmem_handle_create(hd.pool, ps, MEMORY_TYPE_PINNED, 0, &a) != HSA_STATUS_SUCCESS) throw 5;
    have_a = true;
    if (hsa_amd_vmem_handle_create(hd.pool, ps, MEMORY_TYPE_PINNED, 0, &b) != HSA_STATUS_SUCCESS) throw 6;
    have_b = true;
    if (!vm.adopt(a.handle) || !vm.adopt(b.handle)) throw 7;
    void *bo_a = vm.find(a.handle), *bo_b = vm.find(b.handle);
    // (1) same address space?
    if (hsa_amd_vmem_map(va, ps, 0, a, 0) != HSA_STATUS_SUCCESS) throw 8;
    rocr_mapped = true;
    hsa_amd_memory_access_desc_t d{HSA_ACCESS_PERMISSION_RW, hd.agent};
    if (hsa_amd_vmem_set_access(va, ps, &d, 1) != HSA_STATUS_SUCCESS) throw 9;
    if (vm.map(bo_b, va, ps) == 0) { // accepted: libdrm gave us some other VM
      (void)vm.unmap(bo_b, va, ps);
      throw 10;
    }
    if (hsa_amd_vmem_unmap(va, ps) != HSA_STATUS_SUCCESS) throw 11;
    rocr_mapped = false;
    if (vm.map(bo_b, va, ps) != 0) throw 12;
    drm_mapped = bo_b;
    // (2) data follows the handle
    shootdown();
    unsigned char host[256];
    if (hipMemset(va, 0x5a, sizeof host) != hipSuccess || hipDeviceSynchronize() != hipSuccess) throw 13;
    if (vm.unmap(bo_b, va, ps) != 0) throw 14;
    drm_mapped = nullptr;
    if (vm.map(bo_a, va, ps) != 0) throw 15;
    drm_mapped = bo_a;
    shootdown();
    if (hipMemset(va, 0xc3, sizeof host) != hipSuccess || hipDeviceSynchronize() != hipSuccess) throw 16;
    memset(host, 0, sizeof host);
    if (hipMemcpy(host, va, sizeof host, hipMemcpyDeviceToHost) != hipSuccess) throw 17;
    for (unsigned char c : host)
      if (c != 0xc3) throw 18;
    if (vm.unmap(bo_a, va, ps) != 0) throw 19;
    drm_mapped = nullptr;
    if (vm.map(bo_b, va, ps) != 0) throw 20;
    drm_mapped = bo_b;
    shootdown();
    if (hipMemcpy(host, va, sizeof host, hipMemcpyDeviceToHost) != hipSuccess) throw 21;
    for (unsigned char c : host)
      if (c != 0x5a) throw 22;
    ok = true;
    // (3) optional on top: physical memory straight from KFD (flat creation cost). Its failure only means that
    // handles keep coming from ROCr.
    if (vm.unmap(bo_b, va, ps) != 0) throw 23;
    drm_mapped = nullptr;
    bool kfd_ok = false;
    if (env_bool("KVCACHED_DRM_KFD_CREATE", true) && !KVC_TEST_HOOK("FAIL_KFD_SELFTEST")) {
      std::string why_kfd;
      phys_handle_t k = 0;
      int step = 0;
      try {
        if (!vm.open_kfd(&why_kfd)) throw 1;
        k = vm.create(ps);
        void *bo_k = vm.find(k);
        if (!bo_k || vm.map(bo_k, va, ps) != 0) throw 2;
        drm_mapped = bo_k;
        shootdown();
        if (hipMemset(va, 0x77, sizeof host) != hipSuccess || hipDeviceSynchronize() != hipSuccess) throw 3;
        if (hipMemcpy(host, va, sizeof host, hipMemcpyDeviceToHost) != hipSuccess) throw 4;
        for (unsigned char c : host)
          if (c != 0x77) throw 5;
        if (vm.unmap(bo_k, va, ps) != 0) throw 6;
        drm_mapped = nullptr;
        shootdown();
        kfd_ok = true;
      } catch (int st) {
        step = st;
      } catch (const std::exception &e) {
        why_kfd = e.what();
        step = -1;
      }
      (void)hipGetLastError();
      if (drm_mapped) {
        (void)vm.unmap(drm_mapped, va, ps);
        drm_mapped = nullptr;
      }
      if (k) (void)vm.forget(k);
      if (!kfd_ok)
        KVC_LOG(LOG_WARNING, "physical pages straight from KFD unavailable (step %d%s%s): creating them through ROCr", step,
                why_kfd.empty() ? "" : ": ", why_kfd.c_str());
    }
    if (!kfd_ok) vm.disable_kfd();
  } catch (int step) {
    KVC_LOG(LOG_WARNING, "direct DRM mapping self test stopped at step %d", step);
  } catch (const std::exception &e) {
    KVC_LOG(LOG_WARNING, "direct DRM mapping self test: %s", e.what());
  }
  (void)hipGetLastError();
  if (drm_mapped) (void)vm.unmap(drm_mapped, va, ps);
  if (hip_mapped) { // HIP must find something to unmap
    if (!rocr_mapped && have_a) rocr_mapped = hsa_amd_vmem_map(va, ps, 0, a, 0) == HSA_STATUS_SUCCESS;
    if (hipMemUnmap(va, ps) != hipSuccess) {
      (void)hipGetLastError();
      if (rocr_mapped) (void)hsa_amd_vmem_unmap(va, ps);
    }
  }
  shootdown();
  if (have_a) {
    vm.forget(a.handle);
    (void)hsa_amd_vmem_handle_release(a);
  }
  if (have_b) {
    vm.forget(b.handle);
    (void)hsa_amd_vmem_handle_release(b);
  }
  if (shell) (void)hipMemRelease(shell);
  if (va) (void)hipMemAddressFree(va, ps);
  (void)hipGetLastError();
  return ok;
}

// Does the TLB invalidation in effect (GpuContext::tlb_shootdown: the KFD ioctl pair, or its hipMalloc fallback)
// really invalidate, with the backend in effect? Two pages swapped under one VA, looked at through the GPU's own
// address translation (a one-lane kernel): after A has been used through the VA and B mapped in its place, a write
// through the VA must land in B - if it lands in A (found when A is mapped back), the translation survived the
// invalidation, and pages recycled between requests or engines would leak into each other. No fallback makes that
// acceptable: init fails (tests/test_gpu_vmm.py::test_init_refuses_to_start_when_tlb_invalidation_is_ineffective
// breaks the flush with a hook and expects exactly that).
void tlb_self_test(GpuContext *ctx) {
  const size_t ps = kBasePage;
  const int dev = ctx->dev();
  void *va = nullptr;
  unsigned *out = nullptr;
  phys_handle_t a = 0, b = 0, mapped = 0;
  std::string failure;
  auto look = [&](unsigned value, bool write) -> unsigned {
    unsigned host = 0;
    HIP_CHECK(launch_peek_poke(va, out, value, write, ctx->stream()));
    HIP_CHECK(hipStreamSynchronize(ctx->stream()));
    HIP_CHECK(hipMemcpy(&host, out, sizeof host, hipMemcpyDeviceToHost));
    return host;
  };
  auto put = [&](phys_handle_t h) {
    if (vmm_map(va, ps, h)) vmm_set_access(va, ps, dev);
    mapped = h;
  };
  auto take = [&]() {
    vmm_unmap(va, ps, mapped);
    mapped = 0;
  };
  try {
    va = vmm_reserve(ps, ps, nullptr);
    HIP_CHECK(hipMalloc(reinterpret_cast<void **>(&out), 64));
    a = vmm_create(dev, ps, false);
    b = vmm_create(dev, ps, false);
    put(a);
    if (look(0x5a5a5a5au, true) != 0x5a5a5a5au) throw GpuError("a freshly mapped page does not read back what was written");
    take();
    put(b);               // the VA's translation to A may still be cached ...
    ctx->tlb_shootdown(); // ... until now (the path every unmap batch takes)
    if (look(0xc3c3c3c3u, true) != 0xc3c3c3c3u) throw GpuError("the second page does not read back what was written");
    take();
    put(a);
    ctx->tlb_shootdown();
    const unsigned seen = look(0, false);
    if (seen != 0x5a5a5a5au) {
      char buf[200];
      snprintf(buf, sizeof buf, "a write through a re-mapped VA landed in the page that had been unmapped (0x%08x where 0x5a5a5a5a "
               "was expected): the GPU kept a stale translation across the invalidation", seen);
      failure = buf;
    }
  } catch (const std::exception &e) {
    failure = e.what();
  }
  (void)hipGetLastError();
  if (mapped) (void)vmm_try_unmap(va, ps, mapped);
  try {
    ctx->tlb_shootdown(); // the scratch pages were touched through this VA: nothing of it may survive them
  } catch (...) {
  }
  if (a) (void)vmm_try_release(a);
  if (b) (void)vmm_try_release(b);
  if (out) (void)hipFree(out);
  if (va) (void)vmm_try_address_free(va, ps);
  (void)hipGetLastError();
  if (!failure.empty())
    throw GpuError("kvcached_amd: TLB invalidation is ineffective on this system (" + failure +
                   "). Pages could leak between requests and co-located engines: refusing to start.");
}
} // namespace

// ------------------------------------------------------------------ registry
void KvAllocator::init(const std::string &dev_str, size_t page_size, bool contiguous_layout) {
  std::unordered_map<int64_t, std::unique_ptr<KvAllocator>> old; // destroyed after g_mu is released
  std::unique_lock<std::mutex> g(g_mu);
  const DeviceSpec want_dev = parse_device(dev_str);
  if (!g_allocators.empty()) {
    // One device per process (as in the reference: one FTensorAllocator multiton, allocator.cpp:18-22). Re-initialising
    // for the SAME device re-creates the allocators like the reference does (allocator.cpp:75-78); asking for another
    // device while KV tensors of the first one exist would pull the driver state (DrmVm, pooled pages) from under them.
    const bool other_device = want_dev.is_gpu != g_device.is_gpu ||
                              (want_dev.is_gpu && want_dev.index >= 0 && want_dev.index != g_device.index);
    if (other_device) {
      bool live = false;
      for (auto &kv : g_allocators) live = live || kv.second->kv_tensors_created();
      if (live)
        throw InvalidError("init_kvcached(\"" + dev_str + "\") while KV tensors of " +
                           (g_device.is_gpu ? "cuda:" + std::to_string(g_device.index) : std::string("cpu")) +
                           " exist: one device per process - call shutdown_kvcached() first");
    }
    KVC_LOG(LOG_ERROR, "KvAllocator has been initialized. Re-initializing...");
    old.swap(g_allocators);
    g_initialized = false; // until this init has succeeded
    g.unlock();
    old.clear();
    g.lock();
    if (other_device) { // nothing of the old device may linger in pools that DrmVm::open() is about to invalidate
      std::unordered_map<int, std::unique_ptr<GpuContext>> ctxs;
      ctxs.swap(g_contexts);
      g.unlock();
      for (auto &kv : ctxs) kv.second->drain_pools();
      ctxs.clear();
      g.lock();
    }
  }
  if (page_size > 0) {
    if (page_size % kBasePage != 0) // reference aborts here (allocator.cpp:84-90); we report it
      throw InvalidError("Invalid page size: " + std::to_string(page_size) +
                         ", must be a multiple of 2MB (2097152 bytes)");
    g_page_size = page_size;
  }
  // environment knobs, read once per init
  // Default ON since round 2, as in the reference (csrc/ftensor.cpp:160-176: unbacked VA reads as zeros instead of
  // faulting). Round 1 had it off: with per-slot aliases through ROCr every creation cost O(mappings in the process) and
  // the cycle ran at 160 GB/s; with the drm backend's zero extent (backfill_all) it is 1.5 TB/s and start-up is 2 304
  // ioctls for a 288 GiB reservation. KVCACHED_ZERO_BACKFILL=false ("lazy": unbacked VA stays unmapped, a stray access
  // FAULTS) is the opt-in for engines that never touch a block they do not own: 4.3 TB/s, no invalidation on the
  // allocation path.
  options().zero_backfill = env_bool("KVCACHED_ZERO_BACKFILL", true) ? 1 : 0;
  options().zero_fill = env_bool("KVCACHED_ZERO_FILL", true) ? 1 : 0;
  options().pool_bytes = env_i64("KVCACHED_PHYS_POOL_MB", 16384) << 20;
  options().tlb_shootdown = env_bool("KVCACHED_TLB_SHOOTDOWN", true) ? 1 : 0;
  options().defer_unmap_shootdown = env_bool("KVCACHED_DEFER_UNMAP_SHOOTDOWN", false) ? 1 : 0;
  options().pool_idle_ms = std::max<int64_t>(0, env_i64("KVCACHED_POOL_IDLE_MS", 1000));
  options().async_unmap = env_bool("KVCACHED_ASYNC_UNMAP", false) ? 1 : 0;
  options().async_shootdown = env_bool("KVCACHED_ASYNC_SHOOTDOWN", true) ? 1 : 0;
  options().hip_reg_group_mb = std::max<int64_t>(0, env_i64("KVCACHED_HIP_REG_GROUP_MB", 64));
  options().clear_run_slots = std::max<int64_t>(0, env_i64("KVCACHED_DRM_CLEAR_RUN", 16));
  options().phys_chunk_pages = std::min<int64_t>(kMaxExtentPages, std::max<int64_t>(1, env_i64("KVCACHED_PHYS_CHUNK_PAGES", 64)));
  options().extent_waste_pct = std::min<int64_t>(100, std::max<int64_t>(0, env_i64("KVCACHED_EXTENT_WASTE_PCT", 5)));
  options().phys_reserve_bytes = std::max<int64_t>(0, env_i64("KVCACHED_PHYS_RESERVE_MB", 2048)) << 20;
  options().scrub_on_release = env_bool("KVCACHED_SCRUB_ON_RELEASE", true) ? 1 : 0;
  options().map_waits_for_all_flushes = env_bool("KVCACHED_MAP_WAITS_FOR_ALL_FLUSHES", false) ? 1 : 0;
  options().deferred_unmap_flush_us = std::max<int64_t>(0, env_i64("KVCACHED_UNMAP_INVALIDATION_US", 0));
  {
    const char *ms = std::getenv("KVCACHED_MAP_SHOOTDOWN");
    options().map_shootdown_always = (ms && std::string(ms) == "always") ? 1 : 0;
  }
  {
    const char *be = std::getenv("KVCACHED_VMM_BACKEND");
    const std::string b = be ? be : "drm";
    if (b != "hip" && b != "hybrid" && b != "drm")
      throw InvalidError("KVCACHED_VMM_BACKEND must be 'drm', 'hybrid' or 'hip'");
    const int want = b == "hybrid" ? kVmmHybrid : (b == "drm" ? kVmmDrm : kVmmHip);
    // Pooled handles belong to the backend that made them; init's self tests may yet change the backend, and the
    // extent size may differ too: start every init with empty pools.
    for (auto &kv : g_contexts) kv.second->drain_pools();
    vmm_backend() = want;
  }
  options().access_run_slots = std::max<int64_t>(1, env_i64("KVCACHED_ACCESS_RUN_SLOTS", 1));
  options().zero_alias_fanout = std::max<int64_t>(1, env_i64("KVCACHED_ZERO_ALIAS_FANOUT", 256));
  options().fill_chunk_slots = std::max<int64_t>(1, env_i64("KVCACHED_FILL_CHUNK_SLOTS", 1024));
  g_device = want_dev;
  g_contiguous = contiguous_layout;
  if (g_device.is_gpu) {
    HIP_CHECK(hipInit(0));
    g_device.index = resolve_dev_index(g_device);
    GpuContext *ctx = context_for(g_device.index); // validates VMM support + granularity (reference: allocator.cpp:324-343)
    if (vmm_hip_registered() && !hybrid_self_test(g_device.index)) {
      KVC_LOG(LOG_WARNING, "hybrid VMM backend failed its self test on this HIP runtime: using the plain HIP backend "
                           "(KVCACHED_VMM_BACKEND=hip), map/unmap will be ~2x slower");
      vmm_backend() = kVmmHip;
    }
    if (vmm_backend() == kVmmDrm && !drm_self_test(g_device.index)) {
      KVC_LOG(LOG_WARNING, "direct DRM mapping failed its self test: using the hybrid backend (KVCACHED_VMM_BACKEND=hybrid)");
      DrmVm::instance().close();
      vmm_backend() = kVmmHybrid;
    }
    // Whatever backend is left standing: pages are only private if unmapping them really invalidates the GPU's TLBs.
    // No fallback for that - init fails.
    if (options().tlb_shootdown.load()) tlb_self_test(ctx);
  }
  g_allocators[0] = std::make_unique<KvAllocator>(g_device, contiguous_layout,
                                                  g_device.is_gpu ? context_for(g_device.index) : nullptr);
  g_initialized = true;
}

void KvAllocator::shutdown() {
  // Allocators unmap/release through their GpuContext and must not be destroyed under g_mu
  // (region teardown is slow, and nothing below may re-enter the registry); contexts go last.
  std::unordered_map<int64_t, std::unique_ptr<KvAllocator>> victims;
  {
    std::lock_guard<std::mutex> g(g_mu);
    victims.swap(g_allocators);
    g_initialized = false;
  }
  victims.clear();
  std::unordered_map<int, std::unique_ptr<GpuContext>> ctxs;
  {
    std::lock_guard<std::mutex> g(g_mu);
    if (g_allocators.empty()) ctxs.swap(g_contexts);
  }
  for (auto &kv : ctxs) kv.second->drain_pools();
  ctxs.clear();
}

KvAllocator *KvAllocator::global(int64_t group_id) {
  std::lock_guard<std::mutex> g(g_mu);
  auto it = g_allocators.find(group_id);
  if (it != g_allocators.end()) return it->second.get();
  if (g_allocators.empty()) throw InvalidError("KvAllocator::init() must be called first (init_kvcached)");
  // lazily created for an unseen group, with the device/layout of init() (allocator.cpp:101-114)
  auto &slot = g_allocators[group_id];
  slot = std::make_unique<KvAllocator>(g_device, g_contiguous, g_device.is_gpu ? context_for(g_device.index) : nullptr);
  return slot.get();
}

bool KvAllocator::initialized() {
  std::lock_guard<std::mutex> g(g_mu);
  return g_initialized;
}
DeviceSpec KvAllocator::device() { return g_device; }
size_t KvAllocator::page_size() { return g_page_size; }
GpuContext *KvAllocator::gpu() {
  std::lock_guard<std::mutex> g(g_mu);
  if (!g_initialized || !g_device.is_gpu) return nullptr;
  return context_for(g_device.index);
}

void mem_get_info(size_t *free_b, size_t *total_b) {
  size_t of = g_override_free.load(), ot = g_override_total.load();
  if (ot != 0) {
    *free_b = of;
    *total_b = ot;
    return;
  }
  GpuContext *ctx = KvAllocator::gpu();
  if (!ctx) throw NoGpuError("hipMemGetInfo needs a GPU device (init_kvcached with \"cuda:N\"), or a mem-info override");
  ctx->bind();
  HIP_CHECK(hipMemGetInfo(free_b, total_b));
  // What sits idle in our own handle pool is as good as free for this allocator (the reference would have
  // released it, csrc/page.cpp:17): without this, alloc() right after a large free() could be refused.
  *free_b = std::min(*total_b, *free_b + ctx->idle_pool_bytes() + g_pending_unmap_bytes.load());
}
void set_mem_info_override(size_t free_b, size_t total_b) {
  g_override_free = free_b;
  g_override_total = total_b;
}
void device_synchronize() {
  GpuContext *ctx = KvAllocator::gpu();
  if (!ctx) return;
  ctx->bind();
  HIP_CHECK(hipDeviceSynchronize());
}

// ------------------------------------------------------------------ KvAllocator
KvAllocator::KvAllocator(DeviceSpec dev, bool contiguous_layout, GpuContext *ctx)
    : dev_(dev), contiguous_(contiguous_layout), ctx_(ctx) {
  exportable_ = env_bool("KVCACHED_EXPORTABLE_HANDLES", false);
}

KvAllocator::~KvAllocator() {
  if (reclaimer_.joinable()) {
    {
      std::lock_guard<std::mutex> g(mu_);
      reclaimer_stop_ = true;
    }
    pending_cv_.notify_all();
    reclaimer_.join();
  }
  std::lock_guard<std::mutex> g(mu_);
  size_t still_queued = 0;
  std::vector<Phys> lanes; // page ids still backed by lanes: their pages go home once every row is unmapped
  std::vector<phys_handle_t> peers_buffers; // ... and buffers of a peer that back page ids here: released once each
  if (!rows_.empty())
    for (size_t p = 0; p < ids_per_row_; ++p) {
      const size_t idx = rows_[0].first + p;
      if (rows_[0].r->mapped[idx] == 4) lanes.push_back(Phys{rows_[0].r->handle[idx], rows_[0].r->seq[idx]});
      if (rows_[0].r->mapped[idx] == 5) peers_buffers.push_back(rows_[0].r->handle[idx]);
    }
  for (auto &r : layers_) {
    for (auto m : r->mapped) still_queued += (m == 3) ? r->page_size : 0;
    destroy_region(*r); // also unmaps what was still queued (state 3 counts as mapped there)
  }
  g_pending_unmap_bytes -= std::min(still_queued, g_pending_unmap_bytes.load());
  layers_.clear();
  if (lane_pool_ && ctx_) {
    try {
      ctx_->ensure_flushed(); // (destroy_region has invalidated; a region that had nothing mapped owes nothing)
    } catch (...) {
      (void)hipGetLastError();
    }
    if (!lanes.empty()) lane_pool_->release_batch(lanes.data(), lanes.size());
    lane_pool_->drain(0); // an allocator that goes away gives its memory back (the reference releases in ~FTensor, ftensor.cpp:78-98)
  }
  std::sort(peers_buffers.begin(), peers_buffers.end());
  peers_buffers.erase(std::unique(peers_buffers.begin(), peers_buffers.end()), peers_buffers.end());
  for (auto h : peers_buffers) (void)vmm_try_release(h);
  peer_refs_.clear();
}

// ------------------------------------------------------------------ async unmap
// With KVC_OPT_ASYNC_UNMAP the caller's free() path only marks slots and queues them; this thread carries out the
// driver calls (hipMemUnmap is 15 us per slot, 1 ms for one page id of the Llama-3-8B geometry - time the
// scheduler thread of an engine would otherwise spend blocked). Bookkeeping stays synchronous, so page ids and
// block tables are unaffected. A slot that is backed again before its turn is simply kept (no driver call at
// all, just the zero fill). The driver serialises page-table updates per process, so this buys latency, not
// throughput: the reclaimer works in small chunks and yields to foreground map calls.
void KvAllocator::lock_foreground(std::unique_lock<std::mutex> &lk) {
  foreground_waiting_.fetch_add(1);
  lk.lock();
  foreground_waiting_.fetch_sub(1);
}

void KvAllocator::reclaimer_loop() {
  constexpr size_t kChunk = 128; // slots per critical section: <= 2 ms of driver calls between two chances to yield
  if (ctx_) {
    try {
      ctx_->bind();
    } catch (...) {
    }
  }
  std::unique_lock<std::mutex> lk(mu_);
  for (;;) {
    pending_cv_.wait(lk, [&] { return reclaimer_stop_ || !pending_.empty(); });
    if (reclaimer_stop_) return;
    std::vector<Slot> chunk;
    while (!pending_.empty() && chunk.size() < kChunk) {
      Slot s = pending_.front();
      pending_.pop_front();
      if (s.region->mapped[s.index] == 3) { // still released (not re-backed, not already handled via a duplicate entry)
        s.region->mapped[s.index] = 1;
        chunk.push_back(s);
      }
    }
    if (chunk.empty()) {
      if (pending_.empty()) drained_cv_.notify_all();
      continue;
    }
    reclaimer_busy_ = true;
    Unmapped u;
    try {
      if (dev_.is_gpu) {
        unmap_collect(chunk, u);
      } else {
        for (auto &c : chunk) c.region->mapped[c.index] = 0;
        stats().pages_unmapped += (int64_t)chunk.size();
      }
    } catch (const std::exception &e) {
      KVC_LOG(LOG_ERROR, "async unmap failed: %s", e.what());
    }
    g_pending_unmap_bytes -= std::min(chunk.size() * chunk[0].region->page_size, g_pending_unmap_bytes.load());
    lk.unlock();
    try {
      if (dev_.is_gpu) unmap_finish(u, false); // TLB invalidation + handles back to the pool: no allocator state involved
    } catch (const std::exception &e) {
      KVC_LOG(LOG_ERROR, "async unmap (finish) failed: %s", e.what());
    }
    while (foreground_waiting_.load() > 0) std::this_thread::yield(); // map calls go first
    lk.lock();
    reclaimer_busy_ = false;
    if (pending_.empty()) drained_cv_.notify_all();
  }
}

void KvAllocator::flush_unmaps() {
  std::unique_lock<std::mutex> lk(mu_);
  if (!reclaimer_.joinable()) return;
  drained_cv_.wait(lk, [&] { return pending_.empty() && !reclaimer_busy_; });
}

void KvAllocator::flush_all_unmaps() {
  std::vector<KvAllocator *> all;
  {
    std::lock_guard<std::mutex> g(g_mu);
    for (auto &kv : g_allocators) all.push_back(kv.second.get());
  }
  for (auto *a : all) a->flush_unmaps();
  if (GpuContext *ctx = gpu()) { // ... and no invalidation is left owed or in flight, no page still being zeroed
    ctx->bind();
    ctx->ensure_flushed();
    ctx->flush_limbo();
    ctx->wait_all_scrubs();
  }
}

void KvAllocator::quiesce_all(bool on) {
  static std::mutex q_mu;                                   // one holder at a time
  static std::vector<std::unique_lock<std::mutex>> held;    // (owned by the holder's thread between begin and end)
  static std::unique_lock<std::mutex> holder;
  if (on) {
    std::unique_lock<std::mutex> me(q_mu);
    std::vector<KvAllocator *> all;
    {
      std::lock_guard<std::mutex> g(g_mu);
      for (auto &kv : g_allocators) all.push_back(kv.second.get());
    }
    std::sort(all.begin(), all.end()); // a fixed order: two holders cannot exist, but map calls of several groups do
    for (auto *a : all) {
      std::unique_lock<std::mutex> lk(a->mu_, std::defer_lock);
      a->lock_foreground(lk); // (ahead of the reclaimer)
      held.push_back(std::move(lk));
    }
    holder = std::move(me);
  } else {
    if (!holder.owns_lock()) throw InvalidError("kvc_quiesce_end without kvc_quiesce_begin");
    held.clear();
    holder.unlock();
    holder = std::unique_lock<std::mutex>();
  }
}

size_t KvAllocator::pending_unmap_bytes() { return g_pending_unmap_bytes.load(); }

std::unique_ptr<KvRegion> KvAllocator::make_region(const std::string &name, size_t size, size_t page_size) {
  if (size % kBasePage != 0) throw InvalidError("alloc size not aligned.");
  auto r = std::make_unique<KvRegion>();
  r->name = name;
  r->size = size;
  r->page_size = page_size;
  r->on_gpu = dev_.is_gpu;
  if (size % page_size != 0) throw InvalidError("region size is not a multiple of its page size");
  size_t off = g_vaddr_offset.fetch_add(size);
  void *hint = reinterpret_cast<void *>(kStartAddr + off);
  if (dev_.is_gpu) {
    r->base = static_cast<char *>(vmm_reserve(size, kBasePage, hint));
  } else {
    void *p = mmap(hint, size, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS | MAP_NORESERVE, -1, 0);
    if (p == MAP_FAILED) throw InvalidError("mmap failed.");
    r->base = static_cast<char *>(p);
  }
  r->handle.assign(r->num_slots(), phys_handle_t{});
  r->seq.assign(r->num_slots(), 0);
  r->stale_epoch.assign(r->num_slots(), 0);
  r->mapped.assign(r->num_slots(), 0);
  r->mark.assign((r->num_slots() + 63) / 64, 0);
  r->registered.assign(r->num_slots(), 0);
  r->reg_group = std::max<size_t>(1, (size_t)std::max<int64_t>(0, options().hip_reg_group_mb.load()) * (1u << 20) / r->page_size);
  return r;
}

// reference: FTensor::init_with_zero_ (ftensor.cpp:160-176) — every unbacked slot aliases a physical
// page of zeros so that stray reads do not fault. The reference uses ONE such page for the whole
// reservation; on ROCm hipMemUnmap of an alias costs O(aliases of that handle) (~6 ns each: 37 us
// per slot at 6k aliases, 220-260 us at 32k, ~1 ms at the 147k slots of a 288 GiB reservation), so
// the zero page is sharded: one handle per `fanout` slots (default 256 => 0.4 % of the VA size).
// The rest state of unbacked slots on the drm backend where reads of them must not fault: PRT (DrmVm::map_prt), one
// mapping per group of 64 slots. Default in compat mode. NOT in lazy mode: a PRT translation IS cached by the TLBs once something
// has looked at the address (tools/prt_tlb_probe.cpp: every read stale and 0.8 % of the writes lost after backing such
// slots without an invalidation), so backing a PRT slot owes an invalidation before the page is used - and lazy mode's
// contract (nothing touches unbacked VA) is what makes its map path free of one: plain unmapped VA, where a violation
// faults instead of poisoning the next mapping. KVCACHED_PRT=true|false overrides either default.
bool KvAllocator::prt_all(KvRegion &r, bool by_default) {
  if (!dev_.is_gpu || vmm_backend() != kVmmDrm || !DrmVm::instance().can_clear() || !env_bool("KVCACHED_PRT", by_default)) return false;
  hipDeviceProp_t prop{};
  if (hipGetDeviceProperties(&prop, ctx_->dev()) != hipSuccess || strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
    (void)hipGetLastError();
    return false; // what a PRT page does to a load is a property of the GPU: checked on gfx950 (tools/prt_probe.cpp), nowhere else
  }
  // HIP must have been introduced to every slot BEFORE the range is occupied (its placeholder mapping cannot overlap ours)
  for (size_t i = 0; i < r.num_slots(); ++i)
    if (vmm_hip_registered()) register_slot(r, i);
  // One mapping per group of 64 slots (2 304 ioctls for the 147 k slots of a 288 GiB reservation), not one for the region:
  // a page-table entry carries a fragment size that may be as large as its MAPPING, and when a mapping is split what is
  // left of it has to be written again (DrmVm::refresh_prt_remainders) - that stays a matter of <= 64 entries.
  const size_t group = KvRegion::kPrtGroupSlots * r.page_size;
  for (size_t off = 0; off < r.size; off += group) {
    const int rc = DrmVm::instance().map_prt(r.base + off, std::min(group, r.size - off));
    if (rc != 0) {
      KVC_LOG(LOG_WARNING, "PRT mapping refused (%s): unbacked slots fall back to %s", strerror(rc < 0 ? -rc : rc),
              options().zero_backfill.load() ? "zero aliases" : "unmapped VA");
      if (off) {
        StaleAfter mark;
        (void)DrmVm::instance().clear(r.base, off);
      }
      return false;
    }
  }
  r.prt = true;
  return true;
}

int KvAllocator::rest_replace(KvRegion &r, size_t first, size_t n) {
  StaleAfter mark; // whatever was mapped there was a live translation
  char *va = r.base + first * r.page_size;
  if (r.prt) return DrmVm::instance().map_prt(va, n * r.page_size, /*replace=*/true);
  return DrmVm::instance().replace(DrmVm::instance().find(r.zx_handle), va, n * r.page_size, (first % r.zx_pages) * r.page_size);
}

int KvAllocator::rest_map(KvRegion &r, size_t first, size_t n) {
  char *va = r.base + first * r.page_size;
  if (r.prt) return DrmVm::instance().map_prt(va, n * r.page_size);
  return DrmVm::instance().map(DrmVm::instance().find(r.zx_handle), va, n * r.page_size, (first % r.zx_pages) * r.page_size);
}

void KvAllocator::backfill_all(KvRegion &r) {
  if (prt_all(r, true)) { // "reads as zeros" from the page tables themselves: no zero page to keep and to alias
    r.backfilled = true;
    return;
  }
  // drm backend: one buffer of zeros, a whole group of slots aliased per ioctl, no per-slot object anywhere (§4.2):
  // 2 304 ioctls for the 147 k slots of a 288 GiB reservation instead of 147 k map + set_access pairs.
  if (env_bool("KVCACHED_ZERO_EXTENT", true)) {
    size_t zp = 0;
    if (const phys_handle_t zh = ctx_->zero_extent(r.page_size, &zp)) {
      void *zbo = DrmVm::instance().find(zh);
      size_t done = 0;
      try {
        for (size_t i = 0; i < r.num_slots(); ++i)
          if (vmm_hip_registered()) register_slot(r, i); // compat promises zeros to ANY access, hipMemcpy included
        for (; done < r.num_slots(); done += zp) {
          const size_t k = std::min(zp, r.num_slots() - done);
          if (DrmVm::instance().map(zbo, r.base + done * r.page_size, k * r.page_size, 0) != 0)
            throw GpuError("aliasing the zero extent failed");
        }
      } catch (...) {
        if (done) {
          StaleAfter mark;
          (void)DrmVm::instance().clear(r.base, std::min(done, r.num_slots()) * r.page_size);
        }
        throw;
      }
      r.zx = true;
      r.zx_handle = zh;
      r.zx_pages = zp;
      r.backfilled = true;
      return; // invalid -> valid: nothing to invalidate; the extent was filled when it was made
    }
  }
  r.fanout = (size_t)std::max<int64_t>(1, options().zero_alias_fanout.load());
  const size_t n_zero = (r.num_slots() + r.fanout - 1) / r.fanout;
  r.zero.assign(n_zero, phys_handle_t{});
  size_t made = 0;
  try {
    for (; made < n_zero; ++made) r.zero[made] = vmm_create(ctx_->dev(), r.page_size, false, /*direct=*/false);
    const size_t run = (size_t)std::max<int64_t>(1, options().access_run_slots.load());
    for (size_t i = 0; i < r.num_slots(); ++i) {
      if (vmm_hip_registered()) register_slot(r, i); // compat promises zeros to ANY access, hipMemcpy included
      vmm_map(r.base + i * r.page_size, r.page_size, r.zero_of(i));
    }
    for (size_t i = 0; i < r.num_slots(); i += run) {
      size_t k = std::min(run, r.num_slots() - i);
      vmm_set_access(r.base + i * r.page_size, k * r.page_size, ctx_->dev());
    }
  } catch (...) {
    (void)vmm_try_unmap(r.base, r.size);
    try {
      ctx_->tlb_shootdown(); // the aliases were live translations
    } catch (...) {
    }
    for (size_t j = 0; j < made; ++j) (void)vmm_try_release(r.zero[j]);
    r.zero.clear();
    (void)hipGetLastError();
    throw;
  }
  r.backfilled = true;
  // the driver hands out zeroed memory today, but that is not a documented guarantee: fill each
  // zero page once, through its first alias
  ctx_->tlb_shootdown();
  std::vector<void *> firsts;
  for (size_t z = 0; z < n_zero; ++z) firsts.push_back(r.base + z * r.fanout * r.page_size);
  ctx_->zero_fill(firsts.data(), firsts.size(), r.page_size, nullptr);
  ctx_->sync(nullptr);
}

// Hybrid backend. HIP resolves a device pointer through its own table of allocations and mappings; a VA that only
// ROCr knows is "pageable host memory" to hipMemcpy. So each slot is introduced to HIP once: hipMemMap of a placeholder
// handle at the slot's VA (HIP records VA -> memory object), then hsa_amd_vmem_unmap of that very mapping (ROCr's
// state: nothing mapped; HIP never looks again). Every later backing of the slot goes through ROCr only. HIP's copy
// engines address memory by VA and the hardware walks the page tables ROCr maintains, so all hipMemcpy flavours keep
// working, at full speed (tools/hsa_vmm_probe.cpp op 8: D2H 32-50 GB/s, 0 wrong words over re-backing rounds).
// One placeholder per `fanout` slots (as for the zero pages) keeps the number of mappings HIP hangs off one handle
// small. Cost: hipMemMap 3 us + unmap 3 us, once per slot per region lifetime.
void KvAllocator::register_slot(KvRegion &r, size_t slot) {
  if (r.registered[slot]) return;
  constexpr size_t kShellFanout = 4096; // mappings per placeholder handle
  auto prop = make_alloc_prop(ctx_->dev(), false);
  // One hipMemMap introduces a whole group of slots (64 MiB of VA): a slot is registered the first time ANY slot of
  // its group is backed, so nothing can be mapped inside the group at this point. The registration costs the same
  // ~3 us per call whatever its size, i.e. 0.1 us per slot instead of 3.3.
  const bool group = r.in_full_group(slot);
  const size_t first = group ? slot / r.reg_group * r.reg_group : slot, count = group ? r.reg_group : 1;
  auto &shells = group ? r.shell_group : r.shell;
  const size_t tail_base = r.reg_group > 1 ? r.num_slots() / r.reg_group * r.reg_group : 0; // slots behind the last full group
  const size_t shard = (group ? slot / r.reg_group : slot - tail_base) / kShellFanout;
  if (shells.size() <= shard) shells.resize(shard + 1, nullptr);
  if (!shells[shard]) HIP_CHECK(hipMemCreate(&shells[shard], count * r.page_size, &prop, 0));
  char *va = r.base + first * r.page_size;
  HIP_CHECK(hipMemMap(va, count * r.page_size, 0, shells[shard], 0));
  HSA_CHECK(hsa_amd_vmem_unmap(va, count * r.page_size));
  for (size_t i = first; i < first + count; ++i) r.registered[i] = 1;
}

// Teardown of the above: HIP believes its placeholder mappings are still there and must be allowed to unmap them
// (hipMemAddressFree and hipMemRelease expect that). Everything of ours inside a registered unit - pages, zero aliases -
// is unmapped first; then a stand-in of the unit's size is mapped through ROCr for HIP's unmap to remove.
void KvAllocator::unregister_slots(KvRegion &r) {
  if (r.rest_direct()) { // pages and the rest state (PRT, zero aliases) alike are DRM mappings of ours: one ranged CLEAR drops them all
    StaleAfter mark;
    if (DrmVm::instance().clear(r.base, r.size) != 0) KVC_LOG(LOG_ERROR, "dropping the mappings of %s failed", r.name.c_str());
  }
  hsa_amd_vmem_alloc_handle_t standin[2] = {{}, {}}; // [0] one slot, [1] one group
  bool have_standin[2] = {false, false};
  size_t failures = 0;
  for (size_t first = 0; first < r.num_slots();) {
    const bool group = r.in_full_group(first);
    const size_t count = group ? r.reg_group : 1;
    if (!r.registered[first]) {
      first += count;
      continue;
    }
    char *va = r.base + first * r.page_size;
    for (size_t i = first; i < first + count; ++i) {
      char *sva = r.base + i * r.page_size;
      if (r.rest_direct()) continue; // (cleared above)
      if (r.mapped[i]) {
        if (!vmm_try_unmap(sva, r.page_size, r.handle[i])) KVC_LOG(LOG_ERROR, "unmap during cleanup failed (slot %zu)", i);
      } else if (r.backfilled) {
        (void)vmm_try_unmap(sva, r.page_size); // a zero alias
      }
    }
    const int k = group ? 1 : 0;
    if (!have_standin[k])
      have_standin[k] = hsa_amd_vmem_handle_create(hsa_device(ctx_->dev()).pool, count * r.page_size, MEMORY_TYPE_PINNED, 0,
                                                   &standin[k]) == HSA_STATUS_SUCCESS;
    if (!have_standin[k] || hsa_amd_vmem_map(va, count * r.page_size, 0, standin[k], 0) != HSA_STATUS_SUCCESS) {
      failures += count;
    } else if (hipMemUnmap(va, count * r.page_size) != hipSuccess) {
      (void)hipGetLastError();
      (void)hsa_amd_vmem_unmap(va, count * r.page_size);
      failures += count;
    }
    for (size_t i = first; i < first + count; ++i) r.registered[i] = 0;
    first += count;
  }
  for (int k = 0; k < 2; ++k)
    if (have_standin[k]) (void)hsa_amd_vmem_handle_release(standin[k]);
  for (auto *shells : {&r.shell, &r.shell_group}) {
    for (auto h : *shells)
      if (h && hipMemRelease(h) != hipSuccess) (void)hipGetLastError();
    shells->clear();
  }
  if (failures) KVC_LOG(LOG_ERROR, "%zu slots of %s could not be unregistered from HIP", failures, r.name.c_str());
}

void KvAllocator::destroy_region(KvRegion &r) {
  if (!r.base) return;
  if (!r.on_gpu) {
    munmap(r.base, r.size);
    r.base = nullptr;
    return;
  }
  GpuContext *ctx = ctx_;
  if (ctx) (void)hipSetDevice(ctx->dev());
  // Tolerate stale mappings during teardown: log, do not throw (ftensor.cpp:78-98).
  bool whole = false;
  if (vmm_hip_registered()) {
    // HIP unmaps every slot it was told about (that removes our pages' and aliases' mappings too); what was never
    // registered cannot be mapped either
    unregister_slots(r);
    whole = true;
  } else if (r.backfilled) {
    whole = vmm_try_unmap(r.base, r.size);
    if (!whole) KVC_LOG(LOG_ERROR, "unmapping the whole region %s in one call failed", r.name.c_str());
  }
  std::vector<Phys> dead, pieces; // dead: a peer's pages (imports); pieces: our own, which go home through their pool
  ExtentPool *pp = ctx ? ctx->extents(r.page_size, exportable_) : nullptr;
  for (size_t i = 0; i < r.num_slots(); ++i) {
    if (!r.mapped[i]) continue;
    if (!whole) {
      if (!vmm_try_unmap(r.base + i * r.page_size, r.page_size, r.handle[i])) KVC_LOG(LOG_ERROR, "unmap during cleanup failed (slot %zu)", i);
    }
    if (r.mapped[i] != 4 && r.mapped[i] != 5) // (4 / 5: a page of a lane / of a peer's lane - released once, from ~KvAllocator, not once per row)
      (r.mapped[i] != 2 ? pieces : dead).push_back(Phys{r.handle[i], r.seq[i]});
    r.mapped[i] = 0;
  }
  std::sort(dead.begin(), dead.end(), [](const Phys &a, const Phys &b) { return a.seq < b.seq; }); // oldest first
  if (r.backfilled && !whole) // aliases that could not be dropped in one call
    for (size_t i = 0; i < r.num_slots(); ++i) (void)vmm_try_unmap(r.base + i * r.page_size, r.page_size);
  // Pages and zero pages leave the process and the VA range may be handed out again: no translation of either may
  // survive (any unmap above has set tlb_stale(); a region that never had anything mapped owes nothing).
  if (ctx && tlb_stale().load() && !KVC_TEST_HOOK("SKIP_TEARDOWN_FLUSH")) { // hook: prove the test has teeth
    try {
      ctx->tlb_shootdown();
    } catch (...) {
      (void)hipGetLastError();
    }
  }
  for (auto &p : dead) {
    if (!vmm_try_release(p.h)) KVC_LOG(LOG_ERROR, "releasing a physical handle during cleanup failed");
    stats().vmm.released++;
  }
  if (!pieces.empty()) {
    pp->release_batch(pieces.data(), pieces.size());
    pp->drain(0); // a region that goes away gives its memory back (the reference releases in ~FTensor, ftensor.cpp:78-98)
  }
  for (auto z : r.zero) (void)vmm_try_release(z);
  r.zero.clear();
  if (!vmm_try_address_free(r.base, r.size)) KVC_LOG(LOG_ERROR, "freeing the VA range of %s failed", r.name.c_str());
  (void)hipGetLastError();
  r.base = nullptr;
}

std::vector<KvAllocator::TensorDesc> KvAllocator::create_kv_tensors(size_t size, size_t dtype_size,
                                                                    const std::string &dev_str, int64_t num_layers,
                                                                    int64_t num_kv_buffers, bool unified_pool) {
  if (dtype_size != 1 && dtype_size != 2 && dtype_size != 4 && dtype_size != 8)
    throw std::runtime_error("Unsupported dtype size: " + std::to_string(dtype_size));
  if (num_layers <= 0 || num_kv_buffers <= 0) throw InvalidError("num_layers and num_kv_buffers must be positive");
  (void)dev_str; // the reference only asserts it equals the init device (allocator.cpp:298-301)
  std::lock_guard<std::mutex> g(mu_);
  if (num_layers_ != 0 && num_layers_ != num_layers)
    throw InvalidError("create_kv_tensors called again with a different num_layers");
  const size_t ps = g_page_size;
  size_t aligned = size;
  if (size % ps != 0) {
    aligned = ((size + ps - 1) / ps) * ps;
    KVC_LOG(LOG_WARNING, "Size %zu is not aligned to page size %zu, aligning to %zu", size, ps, aligned);
  }
  GpuContext *ctx = ctx_;
  if (ctx) ctx->bind();
  const bool backfill = dev_.is_gpu && options().zero_backfill.load();
  const size_t region_page = contiguous_ ? ps * (size_t)num_layers * (size_t)num_kv_buffers : ps;

  std::vector<TensorDesc> out;
  if (contiguous_) {
    // one region for all layers; slot = compound page (page x layers x kv buffers), allocator.cpp:139-147
    for (auto &r : layers_) destroy_region(*r);
    layers_.clear();
    auto r = make_region("kv_contiguous", aligned * (size_t)num_layers, region_page);
    if (backfill)
      backfill_all(*r);
    else
      (void)prt_all(*r, false); // lazy mode: unmapped VA unless KVCACHED_PRT=true asks for PRT
    out.push_back({r->base, r->size});
    layers_.push_back(std::move(r));
  } else {
    if (!layers_.empty() && (layers_.size() != (size_t)num_layers || layers_[0]->size != aligned))
      throw InvalidError("create_kv_tensors called again with a different size");
    if (layers_.empty()) {
      if (num_kv_buffers == 2 && !unified_pool && aligned % (2 * ps) != 0)
        throw InvalidError("Invalid tensor size: " + std::to_string(aligned) + ", must be a multiple of 2 * page size " +
                           std::to_string(2 * ps));
      for (int64_t i = 0; i < num_layers; ++i) {
        auto r = make_region("kv_" + std::to_string(i), aligned, ps);
        if (backfill)
          backfill_all(*r);
        else
          (void)prt_all(*r, false);
        layers_.push_back(std::move(r));
      }
    }
    for (auto &r : layers_) out.push_back({r->base, r->size});
  }
  const bool first_time = num_layers_ == 0;
  num_layers_ = num_layers;
  num_kv_buffers_ = num_kv_buffers;
  unified_pool_ = unified_pool;
  tensor_bytes_per_layer_ = aligned;
  if (first_time || contiguous_) setup_lanes();
  return out;
}

bool KvAllocator::kv_tensors_created() {
  std::lock_guard<std::mutex> g(mu_);
  return num_layers_ > 0;
}

bool KvAllocator::uses_prt() {
  std::lock_guard<std::mutex> g(mu_);
  return !layers_.empty() && layers_[0]->prt;
}

size_t KvAllocator::lanes_per_extent() {
  std::lock_guard<std::mutex> g(mu_);
  return lanes_ && lane_pool_ ? lane_pool_->max_extent_pages() : 0;
}

std::vector<void *> KvAllocator::region_bases() {
  std::lock_guard<std::mutex> g(mu_);
  std::vector<void *> b;
  for (auto &r : layers_) {
    b.push_back(r->base);
    if (!contiguous_ && !unified_pool_ && num_kv_buffers_ == 2) b.push_back(r->base + r->size / 2);
  }
  return b;
}

// Offsets -> (region, slot) in the reference's order: layer-major, then offset, K before V
// (allocator.cpp:168-207). Contiguous: one compound slot per offset. Unified pool: one slot per layer.
std::vector<KvAllocator::Slot> KvAllocator::slots_for(const offset_t *offsets, size_t n) {
  std::vector<Slot> s;
  auto add = [&](KvRegion *r, offset_t off) {
    if (off < 0 || (size_t)off % r->page_size != 0 || (size_t)off >= r->size)
      throw InvalidError("offset " + std::to_string(off) + " is not a valid page offset of " + r->name);
    s.push_back({r, (size_t)off / r->page_size});
  };
  if (contiguous_) {
    for (size_t i = 0; i < n; ++i) add(layers_[0].get(), offsets[i]);
  } else if (unified_pool_ || num_kv_buffers_ == 1) {
    // One slot per layer per offset. NB for a single KV buffer (MLA) the reference's per-layer branch
    // still maps `offset + size/2` as well (allocator.cpp:189-206), i.e. two pages where its own
    // accounting (page_allocator.cpp:688-695) counts one, and out of range for the upper half of the
    // page ids; we back exactly what is accounted (DESIGN.md "reference quirks").
    s.reserve(layers_.size() * n);
    for (auto &r : layers_)
      for (size_t i = 0; i < n; ++i) add(r.get(), offsets[i]);
  } else {
    s.reserve(layers_.size() * n * 2);
    for (auto &r : layers_) {
      const offset_t v_base = (offset_t)(r->size / 2); // get_v_base_offset, allocator.cpp:46-52
      for (size_t i = 0; i < n; ++i) {
        add(r.get(), offsets[i]);
        add(r.get(), offsets[i] + v_base);
      }
    }
  }
  return s;
}

bool KvAllocator::map_to_kv_tensors(const offset_t *offsets, size_t n) {
  const int64_t t0 = now_ns();
  GpuContext::Foreground fg(ctx_);
  std::unique_lock<std::mutex> g(mu_, std::defer_lock);
  lock_foreground(g);
  if (num_layers_ == 0) {
    KVC_LOG(LOG_ERROR, "try to map to KV tensors when KV tensors are not created");
    return false;
  }
  SegTimer sg;
  if (lanes_ && try_map_lanes(offsets, n)) { // page ids of a multi-row geometry: backed as units (lanes)
    stats().map_calls++;
    stats().map_ns += now_ns() - t0;
    return true;
  }
  const auto slots = slots_for(offsets, n);
  sg.mark(0);
  map_slots(slots, nullptr);
  stats().map_calls++;
  stats().map_ns += now_ns() - t0;
  return true;
}

bool KvAllocator::unmap_from_kv_tensors(const offset_t *offsets, size_t n) {
  const int64_t t0 = now_ns();
  GpuContext::Foreground fg(ctx_);
  std::unique_lock<std::mutex> g(mu_, std::defer_lock);
  lock_foreground(g);
  if (num_layers_ == 0) {
    KVC_LOG(LOG_ERROR, "try to unmap from KV tensors when KV tensors are not created");
    return false;
  }
  SegTimer sg;
  std::vector<offset_t> others;
  if (lanes_) { // page ids backed by lanes go back as units; whatever else the call names takes the generic path
    (void)unmap_lanes(offsets, n, &others);
    if (others.empty()) {
      stats().unmap_calls++;
      stats().unmap_ns += now_ns() - t0;
      return true;
    }
    offsets = others.data();
    n = others.size();
    sg.t = now_ns();
  }
  auto slots = slots_for(offsets, n);
  sg.mark(10);
  bool async = options().async_unmap.load() != 0; // also on the cpu device: same queue and thread, no driver calls
  for (auto &r : layers_) async = async && !r->backfilled; // compat mode promises zeros behind an unmap: stay synchronous
  if (async) {
    std::vector<Slot> now; // imported pages are a peer's memory: dropped at once
    size_t queued = 0;
    for (auto &s : slots) {
      uint8_t &m = s.region->mapped[s.index];
      if (m == 1) {
        m = 3;
        pending_.push_back(s);
        ++queued;
      } else if (m == 2) {
        now.push_back(s);
      } else { // reference: log + skip (ftensor.cpp:124-127)
        KVC_LOG(LOG_ERROR, "Page %zu is not mapped.", s.index);
      }
    }
    if (queued) {
      g_pending_unmap_bytes += queued * slots[0].region->page_size;
      stats().unmaps_queued += (int64_t)queued;
      if (!reclaimer_.joinable()) reclaimer_ = std::thread(&KvAllocator::reclaimer_loop, this);
      pending_cv_.notify_one();
    }
    if (!now.empty()) unmap_slots(now);
  } else {
    unmap_slots(slots);
  }
  stats().unmap_calls++;
  stats().unmap_ns += now_ns() - t0;
  return true;
}

// Async unmap, map side: when the pool has nothing idle, the page of a slot that is released but not yet
// unmapped is the cheapest handle there is (one hipMemUnmap, which was owed anyway) - and the handle population
// stays what it would be with synchronous unmaps instead of growing by hipMemCreate (O(live handles)). Needs mu_.
// The stale translation of the victim's VA is covered by the caller's TLB invalidation (before its fill).
bool KvAllocator::steal_pending(size_t ps, Phys *out) {
  while (!pending_.empty()) {
    Slot s = pending_.front();
    pending_.pop_front();
    KvRegion &r = *s.region;
    if (r.mapped[s.index] != 3 || r.page_size != ps) continue;
    const int64_t t0 = now_ns();
    if (r.rest_direct() && vmm_direct_bo(r.handle[s.index])) {
      if (rest_replace(r, s.index, 1) != 0) throw GpuError("putting a released slot back into its rest state failed");
    } else {
      vmm_unmap(r.base + s.index * ps, ps, r.handle[s.index]);
      if (r.rest_direct() && rest_map(r, s.index, 1) != 0) KVC_LOG(LOG_ERROR, "putting slot %zu back into its rest state failed", s.index);
    }
    if (vmm_extent_pages(r.handle[s.index]) > 1) // one page out of a larger mapping: see unmap_collect
      if (void *bo = vmm_direct_bo(r.handle[s.index])) (void)DrmVm::instance().refresh_mappings_of(bo, ps);
    stats().t_unmap += now_ns() - t0;
    r.mapped[s.index] = 0;
    r.stale_epoch[s.index] = ctx_->next_flush_epoch();
    *out = Phys{r.handle[s.index], r.seq[s.index]};
    g_pending_unmap_bytes -= std::min(ps, g_pending_unmap_bytes.load());
    stats().pages_unmapped++;
    return true;
  }
  return false;
}

using SlotRun = SlotRunOf<KvRegion>;
using RunScan = RunScanOf<KvRegion>;

// The hot loop. Per slot: [register with HIP, once per 64 MiB group] -> [unmap the zero alias] -> pooled handle -> map
// (drm backend: one GEM_VA ioctl; else map + one set_access per contiguous run); zero_fill_pages launches that run on
// the GPU while the host keeps issuing driver calls for the next slots (3/4 of the batch, then the rest); a TLB
// invalidation before the first fill only if one is owed; one stream sync at the end.
void KvAllocator::map_slots(const std::vector<Slot> &slots, const std::vector<phys_handle_t> *imported,
                            std::vector<uint8_t> *imported_consumed) {
  if (slots.empty()) return;
  if (!dev_.is_gpu) { // reference CPUPage::map is a no-op (page.cpp:34-37); keep the double-map diagnostics
    for (auto &s : slots) {
      if (s.region->mapped[s.index] == 3) { // async unmap: released, not yet reclaimed -> kept
        s.region->mapped[s.index] = 1;
        g_pending_unmap_bytes -= std::min(s.region->page_size, g_pending_unmap_bytes.load());
        stats().unmaps_cancelled++;
        stats().pages_mapped++;
        continue;
      }
      if (s.region->mapped[s.index]) {
        KVC_LOG(LOG_ERROR, "Page %zu is already mapped.", s.index);
        continue;
      }
      s.region->mapped[s.index] = 1;
      stats().pages_mapped++;
    }
    return;
  }
  GpuContext *ctx = ctx_;
  ctx->bind();
  const size_t ps = slots[0].region->page_size;
  ExtentPool *pool = ctx->extents(ps, exportable_);
  if (!exportable_ && !imported && !ctx->primary_pool()) ctx->set_primary_pool(pool); // the engine's own pool: the reserve is its
  const bool cold = pool->creations() == 0;
  const bool fill = options().zero_fill.load() && !imported;
  const size_t kMaxRunBytes = ps * (size_t)std::max<int64_t>(1, options().access_run_slots.load());

  std::vector<Slot> done;
  done.reserve(slots.size());
  std::vector<void *> pending, run_pages;
  char *run_start = nullptr;
  size_t run_len = 0;
  bool launched = false;
  size_t next_import = 0;

  // Pages become usable in chunks: driver calls for <=`chunk` slots, ONE TLB shootdown (its cost grows only
  // mildly with the number of new mappings: 0.31 ms @256, 0.40 ms @1024), then the fill kernel
  // for exactly those slots runs on the GPU while the host issues the driver calls of the next chunk.
  // A TLB invalidation is owed before the new pages are touched only where a translation of these VAs (or of these
  // pages) may still sit in a TLB: an unmap whose invalidation was deferred, a slot whose zero alias or PRT entry is
  // being replaced, a page taken from a slot that was released but not yet unmapped. The translation of an UNMAPPED
  // address is never cached on GFX9+ - KFD itself flushes after unmap only on this GPU family, and re-backing slots
  // between busy neighbours with no invalidation after the map reads back right (tools/drm_vmm_probe.cpp mode 3) - so
  // a plain map of an unmapped slot whose last unmap was invalidated needs nothing (KVCACHED_MAP_SHOOTDOWN=always
  // restores it). A PRT entry is another matter: cached once looked at (tools/prt_tlb_probe.cpp).
  const bool always_flush = options().map_shootdown_always.load() != 0;
  bool dirty_tlb = ctx->tlb_owed(); // an invalidation is owed before anything of this batch is touched
  // ... but only the invalidations that cover THESE slots' last unmaps have to be waited for: a stale translation of some
  // other address cannot shadow a mapping made here, and it is gone a moment later anyway (every unmap batch has its
  // invalidation under way on the context's thread). KVCACHED_MAP_WAITS_FOR_ALL_FLUSHES=true restores the blanket wait.
  uint64_t need_epoch = 0;
  for (auto &s : slots) need_epoch = std::max(need_epoch, s.region->stale_epoch[s.index]);
  const bool blanket = options().map_waits_for_all_flushes.load() != 0;
  void *zx_dirty = nullptr; // compat mode, zero extent: its group mappings were split by this batch's REPLACEs (§4.8's hazard)
  bool prt_dirty = false;   // PRT mappings were split by this batch: what is left of them carries fragments that span the new pages
  auto flush_for_batch = [&]() {
    const int64_t t_rw = now_ns();
    if (prt_dirty) { // rewritten before the invalidation that covers them (DrmVm::refresh_prt_remainders)
      if (!KVC_TEST_HOOK("SKIP_PRT_REMAINDER_REFRESH") && !DrmVm::instance().refresh_prt_remainders()) // (hook: prove the test has teeth)
        throw GpuError("rewriting the remainders of split PRT mappings failed");
      tlb_stale().store(true);
      need_epoch = ctx->next_flush_epoch();
      prt_dirty = false;
    }
    if (zx_dirty) { // the remainders of the zero extent's mappings are rewritten before the invalidation that covers them
      if (!DrmVm::instance().refresh_mappings_of(zx_dirty, ps)) KVC_LOG(LOG_ERROR, "rewriting the remaining mappings of the zero extent failed");
      tlb_stale().store(true);
      need_epoch = ctx->next_flush_epoch();
      zx_dirty = nullptr;
    }
    stats().seg[20] += now_ns() - t_rw;
    if (blanket || always_flush)
      ctx->ensure_flushed();
    else
      ctx->ensure_flushed_through(need_epoch);
  };
  // Fill launches: the kernel for the slots mapped so far runs while the host issues the driver calls for the rest, so
  // only the LAST launch is exposed. Large launches are more efficient (ramp and tail are ~10 us whatever the size:
  // 6.8 TB/s at 2 GiB, 6.4 at 512 MiB), the exposed one should be short: a batch of n >= 512 slots is filled as
  // 3/4 + 1/4 (1024 pages: 768 pages hidden behind the last 256 maps, 256 pages = 0.09 ms exposed instead of 0.33 ms
  // for one launch; p50 map batch 2.85 -> 2.6 ms). KVCACHED_FILL_CHUNK_SLOTS caps a launch (and while a TLB
  // invalidation is owed per chunk - compat mode, deferred unmaps - larger chunks mean fewer of them).
  const size_t chunk = (size_t)std::max<int64_t>(1, options().fill_chunk_slots.load());
  const size_t n_total = slots.size();
  // (with run-sized extents the driver calls of a whole batch take less time than a launch: nothing to hide a first
  // instalment behind, and one large launch is the more efficient one)
  size_t next_cut = (n_total >= 512 && !pool->multi_page()) ? std::min(chunk, n_total - std::max<size_t>(128, n_total / 4)) : chunk;
  uint64_t max_ticket = 0; // pages that were zeroed on their way back: nothing to launch, only that scrub to wait for
  auto launch_pending = [&](bool all) {
    size_t i = 0;
    while (pending.size() - i >= next_cut || (all && i < pending.size())) {
      const size_t end = i + std::min(next_cut, pending.size() - i);
      if (always_flush && dirty_tlb) tlb_stale().store(true);
      dirty_tlb = false;
      flush_for_batch(); // only if one of these slots (or, compat mode, an alias replaced in this batch) is still owed one
      for (; i < end; i += std::min<size_t>(kMaxPtrsPerLaunch, end - i))
        ctx->zero_fill(pending.data() + i, std::min<size_t>(kMaxPtrsPerLaunch, end - i), ps, nullptr);
      launched = true;
      next_cut = chunk; // after the first launch: whatever is left, in launches of at most `chunk`
    }
    pending.erase(pending.begin(), pending.begin() + i);
  };
  auto flush_run = [&]() {
    if (!run_len) return;
    const int64_t ta = now_ns();
    vmm_set_access(run_start, run_len, ctx->dev());
    stats().t_access += now_ns() - ta;
    if (always_flush) dirty_tlb = true;
    if (fill) {
      pending.insert(pending.end(), run_pages.begin(), run_pages.end());
      launch_pending(false);
    }
    run_pages.clear();
    run_len = 0;
  };

  std::vector<Slot> kept; // async unmap: released but not yet unmapped -> simply kept, only zero-filled again
  // Run-sized extents (KVCACHED_PHYS_CHUNK_PAGES > 1, drm backend): the unbacked slots of the batch are collected
  // first and then backed run by run - adjacent slots get adjacent pages of one buffer and ONE map ioctl.
  const bool chunked = pool->multi_page() && !imported;
  RunScan fresh;
  const bool hip_reg = vmm_hip_registered();
  SegTimer sg;
  try {
    for (auto &s : slots) {
      KvRegion &r = *s.region;
      if (chunked && r.mapped[s.index] == 0 && (!r.backfilled || r.rest_direct())) { // the common case, nothing per slot but a bit
        if (hip_reg && !r.registered[s.index]) {
          const int64_t t0 = now_ns();
          register_slot(r, s.index); // once per slot
          stats().t_unmap_alias += now_ns() - t0;
        }
        // the same slot listed twice in one call: the reference logs "already mapped" for the second and goes on (ftensor.cpp:104-107)
        if (!fresh.add(&r, s.index)) KVC_LOG(LOG_ERROR, "Page %zu is already mapped.", s.index);
        continue; // backed below, run by run: adjacent slots share one ioctl
      }
      if (r.mapped[s.index] == 3 && !imported) {
        r.mapped[s.index] = 1; // its queue entry is dropped by the reclaimer (state no longer 3)
        kept.push_back(s);
        g_pending_unmap_bytes -= std::min(ps, g_pending_unmap_bytes.load());
        stats().unmaps_cancelled++;
        if (fill) pending.push_back(r.base + s.index * ps);
        continue;
      }
      if (r.mapped[s.index]) { // reference: log + skip, the batch still succeeds (ftensor.cpp:104-107)
        KVC_LOG(LOG_ERROR, "Page %zu is already mapped.", s.index);
        if (imported) ++next_import;
        continue;
      }
      char *va = r.base + s.index * ps;
      int64_t t0 = now_ns();
      if (hip_reg && !r.registered[s.index]) register_slot(r, s.index); // once per slot
      if (r.rest_direct() && !chunked) { // (chunked: replaced run by run below)
        if (DrmVm::instance().clear(va, ps) != 0) throw GpuError("dropping the rest mapping of a slot failed");
        // a zero alias was a live translation, and so - for the TLBs - is a PRT entry that something has looked at
        // (tools/prt_tlb_probe.cpp): an invalidation is owed before the page is used
        tlb_stale().store(true);
        if (r.zx) zx_dirty = DrmVm::instance().find(r.zx_handle);
        if (r.prt) prt_dirty = true;
        dirty_tlb = true;
        need_epoch = ctx->next_flush_epoch();
      } else if (r.backfilled && !r.rest_direct()) {
        vmm_unmap(va, ps);
        dirty_tlb = true; // the alias's translation is live
        need_epoch = ctx->next_flush_epoch();
      }
      int64_t t1 = now_ns();
      bool recycled = false;
      Phys ph;
      size_t import_index = 0;
      if (imported) {
        import_index = next_import;
        ph = Phys{(*imported)[next_import++], 0};
      } else if (pool->acquire_run(1, &ph, &recycled, false) == 1) {
        recycled = true;
      } else if (steal_pending(ps, &ph)) { // async unmap: take the page of a released slot instead of creating one
        recycled = true;
        dirty_tlb = true;
        need_epoch = ctx->next_flush_epoch(); // (the page's old address was live a moment ago: conservative)
      } else {
        if (ctx->limbo_bytes()) ctx->flush_limbo(); // pages of an earlier unmap wait for their invalidation: have it now rather than create
        (void)pool->acquire_run(1, &ph, &recycled, true);
      }
      phys_handle_t h = ph.h;
      int64_t t2 = now_ns();
      bool needs_access = true;
      try {
        needs_access = vmm_map(va, ps, h);
      } catch (...) {
        if (!imported) pool->release(ph);
        if (r.rest_direct())
          (void)rest_map(r, s.index, 1); // (its rest mapping was cleared above)
        else if (r.backfilled && vmm_try_map(va, ps, r.zero_of(s.index)))
          (void)vmm_try_set_access(va, ps, ctx->dev());
        throw;
      }
      int64_t t3 = now_ns();
      stats().t_unmap_alias += t1 - t0;
      stats().t_acquire += t2 - t1;
      stats().t_map += t3 - t2;
      r.handle[s.index] = h;
      r.seq[s.index] = ph.seq;
      r.mapped[s.index] = imported ? 2 : 1;
      if (imported && imported_consumed) (*imported_consumed)[import_index] = 1; // the slot owns it now (rollback releases it)
      done.push_back(s);
      if (!needs_access) { // mapped readable+writable in one ioctl (drm backend): straight to the fill queue
        if (always_flush) dirty_tlb = true;
        max_ticket = std::max(max_ticket, ph.wait_ticket); // (any older fill through the alias must be over before the caller writes)
        if (fill && ph.scrub_ticket) {
          max_ticket = std::max(max_ticket, ph.scrub_ticket);
          stats().pages_prescrubbed++;
        } else if (fill) {
          pending.push_back(va);
          launch_pending(false);
        }
        continue;
      }
      if (!(run_len && va == run_start + run_len && run_len < kMaxRunBytes)) {
        flush_run();
        run_start = va;
      }
      run_len += ps;
      run_pages.push_back(va);
    }
    flush_run();
    sg.mark(1);
    if (fresh.size()) {
      std::vector<Phys> got(kMaxExtentPages);
      const std::vector<SlotRun> runs = fresh.collect();
      stats().seg[23] += (int64_t)runs.size();
      sg.mark(2);
      for (const SlotRun &run : runs) {
        KvRegion &r = *run.r;
        for (size_t at = run.first, end = run.first + run.count; at < end;) { // as many neighbours as one extent can serve at a time
          const int64_t t1 = now_ns();
          bool recycled = false;
          size_t prescrubbed = 0;
          sg.mark(5);
          size_t n = pool->acquire_run(end - at, got.data(), &recycled, false);
          if (n == 0 && steal_pending(ps, &got[0])) {
            n = 1; // (its unmap has set tlb_stale: flushed before the fill)
            need_epoch = ctx->next_flush_epoch();
          }
          if (n == 0 && ctx->limbo_bytes()) { // pages of an earlier unmap wait for their invalidation: have it now rather than create
            ctx->flush_limbo();
            n = pool->acquire_run(end - at, got.data(), &recycled, false);
          }
          if (n == 0) n = pool->acquire_run(end - at, got.data(), &recycled, true);
          const int64_t t2 = now_ns();
          sg.mark(3);
          char *va = r.base + at * ps;
          try {
            if (r.rest_direct()) { // the slots carry their rest mapping: pages take its place in the same ioctl
              vmm_replace_pieces(va, ps, n, got[0].h, /*live=*/true);
              // What was there may sit in a TLB: a zero alias is a live translation (and the split remainders of its
              // mapping are rewritten first), and a PRT entry is cached like one as soon as anything has read or written
              // the address (tools/prt_tlb_probe.cpp: after a chip-wide read of PRT slots, backing them without an
              // invalidation left every read stale and lost 0.8 % of the writes). Invalidate before use.
              if (r.zx) zx_dirty = DrmVm::instance().find(r.zx_handle);
              if (r.prt) prt_dirty = true;
              tlb_stale().store(true);
              dirty_tlb = true;
              need_epoch = ctx->next_flush_epoch();
            } else {
              vmm_map_pieces(va, ps, n, got[0].h);
            }
          } catch (...) {
            pool->release_batch(got.data(), n);
            throw;
          }
          stats().t_acquire += t2 - t1;
          stats().t_map += now_ns() - t2;
          stats().seg[21]++;
          sg.mark(4);
          for (size_t k = 0; k < n; ++k) {
            const size_t index = at + k;
            r.handle[index] = got[k].h;
            r.seq[index] = got[k].seq;
            r.mapped[index] = 1;
            done.push_back(Slot{&r, index});
            max_ticket = std::max(max_ticket, got[k].wait_ticket);
            if (fill && got[k].scrub_ticket) {
              max_ticket = std::max(max_ticket, got[k].scrub_ticket);
              ++prescrubbed;
            } else if (fill)
              pending.push_back(r.base + index * ps);
          }
          if (prescrubbed) stats().pages_prescrubbed += (int64_t)prescrubbed;
          if (always_flush) dirty_tlb = true;
          if (fill) launch_pending(false);
          at += n;
        }
      }
    }
    sg.mark(5);
    if (fill) launch_pending(true);
    if (always_flush && dirty_tlb) tlb_stale().store(true);
    flush_for_batch(); // nothing may reach the new mappings through a stale translation of their own addresses
    sg.mark(6);
    const int64_t ts = now_ns();
    if (launched) ctx->sync(nullptr);
    sg.mark(7);
    ctx->wait_scrub(max_ticket);
    stats().t_sync += now_ns() - ts;
    sg.mark(8);
    if (max_ticket) stats().seg[9] += (int64_t)(ctx->scrubs_issued() - max_ticket); // (diagnostics: how many scrubs behind the newest the one waited for was)
  } catch (...) {
    // leave the regions as they were before this call; PageAllocator rolls the page ids back
    if (launched) (void)hipStreamSynchronize(ctx->stream());
    for (auto &k : kept) { // still in the reclaimer's queue: just hand them back to it
      k.region->mapped[k.index] = 3;
      g_pending_unmap_bytes += ps;
    }
    for (auto it = done.rbegin(); it != done.rend(); ++it) {
      KvRegion &r = *it->region;
      char *va = r.base + it->index * ps;
      if (r.rest_direct()) { // back to its rest state, the page dropped in the same ioctl
        (void)rest_replace(r, it->index, 1);
      } else {
        (void)vmm_try_unmap(va, ps, r.handle[it->index]);
      }
      if (vmm_extent_pages(r.handle[it->index]) > 1)
        if (void *bo = vmm_direct_bo(r.handle[it->index])) (void)DrmVm::instance().refresh_mappings_of(bo, ps);
      if (r.mapped[it->index] == 1)
        pool->release(Phys{r.handle[it->index], r.seq[it->index]});
      else
        (void)vmm_try_release(r.handle[it->index]);
      r.mapped[it->index] = 0;
      if (r.backfilled && !r.rest_direct() && vmm_try_map(va, ps, r.zero_of(it->index))) (void)vmm_try_set_access(va, ps, ctx->dev());
    }
    (void)hipGetLastError();
    try {
      ctx->tlb_shootdown();
    } catch (...) {
    }
    throw;
  }
  stats().pages_mapped += (int64_t)(done.size() + kept.size());
  (void)cold;
  if (pool->creations() > 0 && !imported && !exportable_) cold_start_reserve(pool); // (the reserve belongs to the engine's own pool, not to the exportable twin)
}

// Cold start: the first map calls of an allocator bring the reserve along (KVCACHED_PHYS_RESERVE_MB): pages
// that come back are zeroed behind the unmap, and a free()+alloc() cycle that can draw on an idle batch never waits for that
// fill. In instalments of 512 MiB per map call until it has stood once (the whole 2 GiB inside the first call was
// 7.6 ms in the smoke run and ~80 ms on VRAM the kernel has not wiped) - a bare C-ABI caller has no housekeeping thread to make
// it; where one exists (an engine: PageAllocator's watcher, 512 MiB per 100 ms tick) it takes over after the first instalment
// and keeps it up from then on.
void KvAllocator::cold_start_reserve(ExtentPool *pool) {
  if (reserve_built_) return;
  const size_t unit = pool->page_bytes();
  const size_t want = ctx_->reserve_target_bytes() / unit;
  if (!want || pool->idle_bytes() / unit >= want) {
    reserve_built_ = true;
    return;
  }
  (void)pool->refill_reserve(want, std::max<size_t>(1, (512u << 20) / unit));
  if (ctx_->has_housekeeper()) reserve_built_ = true; // (the thread makes the rest)
}

void KvAllocator::unmap_slots(const std::vector<Slot> &slots) {
  if (slots.empty()) return;
  if (!dev_.is_gpu) {
    for (auto &s : slots) {
      if (!s.region->mapped[s.index]) {
        KVC_LOG(LOG_ERROR, "Page %zu is not mapped.", s.index);
        continue;
      }
      s.region->mapped[s.index] = 0;
      stats().pages_unmapped++;
    }
    return;
  }
  Unmapped u;
  unmap_collect(slots, u);
  unmap_finish(u, true);
}

// Driver unmaps (and, in compat mode, the re-aliasing) of `slots`; the handles are only collected. Needs mu_.
void KvAllocator::unmap_collect(const std::vector<Slot> &slots, Unmapped &u) {
  GpuContext *ctx = ctx_;
  ctx->bind();
  const size_t ps = slots[0].region->page_size;
  u.page_size = ps;
  char *run_start = nullptr;
  size_t run_len = 0;
  const size_t max_run = ps * (size_t)std::max<int64_t>(1, options().access_run_slots.load());
  auto flush_run = [&]() {
    if (!run_len) return;
    vmm_set_access(run_start, run_len, ctx->dev());
    run_len = 0;
  };
  u.own.reserve(slots.size());
  // Pages of a multi-page extent may have been mapped together with their neighbours in one ioctl; taking some of them
  // out splits that mapping, and what is left of it must be rewritten (DrmVm::refresh_mappings_of) before the TLBs are
  // invalidated. An extent that loses ALL its mapped pages in this batch leaves nothing behind to rewrite.
  ExtentPool *xpool = ctx->extents(ps, exportable_);
  KeyGroups<uint32_t> touched(slots.size()); // extent -> pieces of it that this batch takes back
  std::vector<std::pair<KvRegion *, size_t>> gone;
  gone.reserve(slots.size());
  // drm backend, lazy regions: slots of this batch that are neighbours in VA go in runs - one CLEAR ioctl per run of up
  // to `clear_run_slots` instead of one UNMAP per slot (whatever the order the caller listed them in). Everything
  // inside such a run is a direct mapping of a page this very call gives up, so "drop all mappings in the range" is exact.
  std::vector<uint8_t> cleared(slots.size(), 0);
  SegTimer sg;
  const bool own_direct = xpool->multi_page(); // our own pages are direct DRM buffers (extent pools exist only on that path)
  { // PRT / zero extent: a run of adjacent slots goes back to its rest state with ONE ioctl that replaces whatever is
    // mapped there (zero extent: runs end at the extent's group boundaries - slot i shows page i % Z)
    RunScan scan;
    for (uint32_t i = 0; i < slots.size(); ++i) {
      KvRegion &r = *slots[i].region;
      const uint8_t m = r.mapped[slots[i].index];
      // (a page ROCr mapped is ROCr's to unmap; our own pages of an extent pool are KFD buffers imported into DRM, all of them:
      // no need to ask DrmVm about each one. A slot listed twice: the second mention is logged as "not mapped" below)
      if (r.rest_direct() && ((m == 1 && own_direct) || ((m == 1 || m == 2) && vmm_direct_bo(r.handle[slots[i].index]))))
        if (scan.add(&r, slots[i].index)) cleared[i] = 2; // (2: and its extent is counted in `touched` with its run)
    }
    const std::vector<SlotRun> runs = scan.collect([](const KvRegion &r) { return r.rest_group(); });
    sg.mark(11);
    for (const SlotRun &run : runs) {
      const int64_t t0 = now_ns();
      const int rc = rest_replace(*run.r, run.first, run.count);
      if (rc != 0) throw GpuError(std::string("DRM_AMDGPU_GEM_VA replace (back to the rest state) failed: ") + strerror(rc < 0 ? -rc : rc));
      stats().t_unmap += now_ns() - t0;
      stats().seg[22]++;
      // the extents these pages belong to, in address order: neighbours share theirs, so the list stays short
      if (own_direct)
        for (size_t k = run.first; k < run.first + run.count; ++k) {
          const phys_handle_t h = run.r->handle[k];
          if (run.r->mapped[k] == 1 && is_piece(h)) ++touched.at(chunk_of(h));
        }
    }
  }
  sg.mark(12);
  size_t max_clear = (size_t)options().clear_run_slots.load();
  if (max_clear >= 2 && xpool->multi_page()) max_clear = std::max<size_t>(max_clear, xpool->max_extent_pages()); // a whole extent in one go
  if (max_clear >= 2 && vmm_backend() == kVmmDrm && DrmVm::instance().can_clear()) {
    RunScan scan; // (a slot listed twice: its second mention is logged as "not mapped" below)
    for (uint32_t i = 0; i < slots.size(); ++i) {
      KvRegion &r = *slots[i].region;
      const uint8_t m = r.mapped[slots[i].index];
      if (!r.backfilled && !r.rest_direct() && ((m == 1 && own_direct) || ((m == 1 || m == 2) && vmm_direct_bo(r.handle[slots[i].index]))))
        if (scan.add(&r, slots[i].index)) cleared[i] = 1;
    }
    for (const SlotRun &run : scan.collect([](const KvRegion &) { return (size_t)-1; }, max_clear)) {
      const int64_t t0 = now_ns();
      vmm_unmap_run(run.r->base + run.first * ps, run.count * ps);
      stats().t_unmap += now_ns() - t0;
    }
  }
  for (size_t si = 0; si < slots.size(); ++si) {
    const Slot &s = slots[si];
    KvRegion &r = *s.region;
    if (r.mapped[s.index] != 1 && r.mapped[s.index] != 2) { // reference: log + skip (ftensor.cpp:124-127)
      KVC_LOG(LOG_ERROR, "Page %zu is not mapped.", s.index);
      continue;
    }
    char *va = r.base + s.index * ps;
    if (!cleared[si]) {
      const int64_t t0 = now_ns();
      vmm_unmap(va, ps, r.handle[s.index]);
      stats().t_unmap += now_ns() - t0;
    }
    const bool own = r.mapped[s.index] == 1;
    if (own)
      u.own.push_back(Phys{r.handle[s.index], r.seq[s.index]});
    else
      u.imported.push_back(r.handle[s.index]);
    r.mapped[s.index] = 0;
    gone.emplace_back(&r, s.index);
    ++u.n;
    if (own && own_direct && cleared[si] != 2 && is_piece(r.handle[s.index])) ++touched.at(chunk_of(r.handle[s.index])); // (a tagged handle of our own pool: a page of a multi-page extent; those of the runs above are counted already)
    if (r.rest_direct()) {
      // Its rest state is back already (REPLACE above). In compat mode the invalidation happens inside this call:
      // "unbacked VA reads as zeros" holds from the moment unmap returns (an asynchronous invalidation would let a read
      // that follows at once still see the old page - observable, so not done); a lazy region's runs behind the call.
      if (r.backfilled) u.any_backfilled = true;
      if (!cleared[si] && rest_map(r, s.index, 1) != 0) // unmapped through ROCr just now (a page imported that way)
        KVC_LOG(LOG_ERROR, "putting slot %zu back into its rest state failed", s.index);
    } else if (r.backfilled) { // put the shared zero page back (ftensor.cpp:135-136), access ranged per run
      u.any_backfilled = true;
      const int64_t tr = now_ns();
      vmm_map(va, ps, r.zero_of(s.index));
      stats().t_realias += now_ns() - tr;
      if (!(run_len && va == run_start + run_len && run_len < max_run)) {
        flush_run();
        run_start = va;
      }
      run_len += ps;
    }
  }
  flush_run();
  sg.mark(13);
  if (!touched.items().empty() && !KVC_TEST_HOOK("SKIP_REMAINDER_REFRESH")) { // hook: prove the test has teeth
    for (auto &kv : touched.items()) {
      // pieces still handed out beyond the ones this batch takes back = pages of the extent that stay mapped
      if (xpool->pieces_out(kv.first) > kv.second)
        if (void *bo = DrmVm::instance().find(kv.first))
          if (!DrmVm::instance().refresh_mappings_of(bo, ps)) KVC_LOG(LOG_ERROR, "rewriting the remaining mappings of an extent failed");
    }
    tlb_stale().store(true);
  }
  sg.mark(14);
  { // every driver call that removed or rewrote a translation has returned: the next invalidation to START covers them all
    const uint64_t epoch = ctx->next_flush_epoch();
    for (auto &g : gone) g.first->stale_epoch[g.second] = epoch;
    u.epoch = epoch;
  }
  sg.mark(15);
}

// TLB invalidation, then the handles go back to the pool / the driver. Touches no allocator state (the async
// reclaimer calls it without mu_).
void KvAllocator::unmap_finish(Unmapped &u, bool may_defer_shootdown) {
  if (!u.n) return;
  GpuContext *ctx = ctx_;
  ExtentPool *pool = ctx->extents(u.page_size, exportable_);
  // Stale TLB entries still translate the unmapped VAs to the old physical pages. Who can be hurt by them?
  //   * a reader of the VA itself: only in compat mode is that legal (unbacked VA reads as zeros), so there the
  //     invalidation happens now;
  //   * the next owner of the physical page: if it stays in OUR pool, its next use is a map_slots() batch, which
  //     invalidates before anything touches the page - the invalidation CAN wait for that (or for the moment the
  //     pool gives handles back to the driver: ExtentDriver::before_release). Optional and off by default
  //     (KVC_OPT_DEFER_UNMAP_SHOOTDOWN): measured, the cost is conserved, not saved - the invalidation mostly waits
  //     for the page-table updates the unmaps queued, so it only moves from free() into the next alloc()
  //     (per page: unmap 354 -> 21 us, next map 205 -> 379 us; batches unchanged).
  //   * memory leaving this process (imported pages, pool evictions): invalidate first.
  //   * who pays: with an allocator watcher thread around (an engine, not a bare C-ABI caller) the owed invalidation
  //     is left to its next 100 ms tick (GpuContext::housekeeping) - the 0.3-0.4 ms KFD round trip leaves the
  //     caller's free() and, unless an alloc follows within that tick, reaches nobody's critical path
  //     (KVCACHED_ASYNC_SHOOTDOWN, on by default; one page id: free() 0.6 -> 0.2 ms).
  SegTimer sg;
  const int64_t trail_us = options().deferred_unmap_flush_us.load();
  if (trail_us > 0 && u.any_backfilled && u.imported.empty() && may_defer_shootdown && !u.own.empty()) {
    // compat, relaxed (KVCACHED_UNMAP_INVALIDATION_US): the invalidation trails the call by at most trail_us - performed by
    // the context's thread, or absorbed by the next map batch's own. The pages wait for it un-scrubbed and un-offered.
    auto own = std::make_shared<std::vector<Phys>>(std::move(u.own));
    const size_t ps = u.page_size;
    ctx->park(u.epoch, own->size() * ps, [ctx, pool, own, ps]() {
      uint64_t ticket = 0;
      if (options().zero_fill.load() && options().scrub_on_release.load() && pool->multi_page()) {
        std::vector<uint64_t> addrs(own->size());
        if (pool->scrub_addresses(own->data(), own->size(), addrs.data())) ticket = ctx->scrub(addrs.data(), addrs.size(), ps);
      }
      pool->release_batch(own->data(), own->size(), ticket);
    });
    ctx->request_async_flush(trail_us);
    sg.mark(16);
    stats().pages_unmapped += u.n;
    return;
  }
  const bool defer = options().defer_unmap_shootdown.load() || options().async_shootdown.load();
  if (u.any_backfilled || !u.imported.empty() || !may_defer_shootdown || !defer) {
    // (starting it on the context's thread and overlapping the scrub launch and the pool with it was tried: the thread
    // hand-off costs what the overlap gains)
    ctx->ensure_flushed();
  } else if (options().defer_unmap_shootdown.load())
    ctx->defer_tlb_shootdown(); // explicitly left to the next map batch / release to the driver
  else
    ctx->request_async_flush(); // this context's own thread does it now, off the caller's path
  for (auto h : u.imported) {
    if (!vmm_try_release(h)) KVC_LOG(LOG_ERROR, "releasing an imported handle failed");
  }
  sg.mark(16);
  const int64_t tr0 = now_ns();
  // Zero the pages on their way back: the fill is queued (through the alias mappings of their buffers) BEFORE they are on
  // offer again, and whoever gets them next only waits for that ticket (GpuContext::scrub).
  uint64_t ticket = 0;
  if (options().zero_fill.load() && options().scrub_on_release.load() && pool->multi_page() && !u.own.empty()) {
    std::vector<uint64_t> addrs(u.own.size());
    if (pool->scrub_addresses(u.own.data(), u.own.size(), addrs.data())) ticket = ctx->scrub(addrs.data(), addrs.size(), u.page_size);
  }
  sg.mark(17);
  pool->release_batch(u.own.data(), u.own.size(), ticket);
  sg.mark(18);
  stats().t_release += now_ns() - tr0;
  stats().pages_unmapped += u.n;
}

// ------------------------------------------------------------------ lanes
// The geometry engines use on ROCm (the reference forces it there: kvcached/utils.py:150-171) is one region per layer with a
// K half and a V half, so ONE page id is `rows` = layers x 2 single slots in as many places of the address space
// (csrc/allocator.cpp:189-206: 64 for Llama-3-8B) and k consecutive page ids are `rows` runs of k slots. What the kernel
// charges for is the page-table RANGE, not the byte (tools/engine_ioctl_probe.cpp: 2.5 us per MAP, 4.1 per REPLACE over
// PRT, +0.23 us per further page of the range, 2.7 us per remainder of a PRT mapping that a REPLACE split), so a page id
// costs `rows` ioctls whatever user space does - and everything else should cost nothing. A page id is allocated and freed as a
// unit (PageAllocator), so it is BACKED as a unit: a LANE, the `rows` pages behind one page id, taken from and given back
// to the pool in one piece (1 pool operation per page id instead of 64; what is recycled has exactly the shape that is
// asked for, so the footprint is that of the page ids in use). A buffer holds k lanes ROW-MAJOR - page (row, lane) at
// (row x k + lane) - so that k consecutive page ids backed by one call are still ONE ioctl per row; lanes of such a buffer
// go back one by one (a free lane of a partly used buffer serves the next single page id at full speed: 64 ioctls
// either way) and the buffer leaves for the driver when the last one is home. ExtentPool does the bookkeeping with the
// lane as its unit; this file knows where a lane's pages are.
void KvAllocator::setup_lanes() {
  lanes_ = false;
  rows_.clear();
  lane_pool_ = nullptr;
  if (!dev_.is_gpu || contiguous_ || exportable_ || layers_.empty() || !ctx_) return;
  if (!env_bool("KVCACHED_LANE_EXTENTS", true) || options().phys_chunk_pages.load() <= 1) return;
  if (options().async_unmap.load()) return; // queued unmaps work slot by slot (the reclaimer's chunks): the per-slot path keeps them
  const bool kv = !unified_pool_ && num_kv_buffers_ == 2;
  for (auto &r : layers_) {
    if (r->backfilled && !r->rest_direct()) return; // (zero aliases through ROCr: the fallback's fallback keeps the per-slot path)
    rows_.push_back(Row{r.get(), 0});
    if (kv) rows_.push_back(Row{r.get(), r->num_slots() / 2}); // get_v_base_offset, allocator.cpp:46-52
  }
  ids_per_row_ = kv ? layers_[0]->num_slots() / 2 : layers_[0]->num_slots();
  if (rows_.size() >= 2) lane_pool_ = ctx_->lane_extents(rows_.size(), layers_[0]->page_size);
  if (!lane_pool_) {
    rows_.clear();
    return;
  }
  lanes_ = true;
}

namespace {
// slots [first, first + n) of a region, cut where its rest state must be (zero extent: its pages repeat; PRT: groups)
template <class F> void for_rest_pieces(const KvRegion &r, size_t first, size_t n, F &&f) {
  const size_t g = r.rest_group();
  while (n) {
    const size_t take = g == (size_t)-1 ? n : std::min(n, g - first % g);
    f(first, take);
    first += take;
    n -= take;
  }
}
} // namespace

bool KvAllocator::try_map_lanes(const offset_t *offsets, size_t n) {
  if (!n) return true;
  const size_t R = rows_.size(), ps = rows_[0].r->page_size;
  SegTimer sg;
  // Every slot the call names must be unbacked. Anything else - a page id named twice, a slot that is mapped already, an
  // imported page, a queued unmap, an offset that is none - is the generic path's business (it logs and goes on, or
  // throws, as the reference does: ftensor.cpp:104-107).
  uint64_t need_epoch = 0;
  RunScan scan;
  for (size_t i = 0; i < n; ++i) {
    const offset_t off = offsets[i];
    if (off < 0 || (size_t)off % ps != 0 || (size_t)off / ps >= ids_per_row_) return false;
    const size_t p = (size_t)off / ps;
    for (const Row &row : rows_) {
      if (row.r->mapped[row.first + p] != 0) return false;
      need_epoch = std::max(need_epoch, row.r->stale_epoch[row.first + p]);
    }
    if (!scan.add(rows_[0].r, rows_[0].first + p)) return false;
  }
  GpuContext *ctx = ctx_;
  ctx->bind();
  ExtentPool *pool = lane_pool_;
  if (!ctx->primary_pool()) ctx->set_primary_pool(pool);
  const bool cold = pool->creations() == 0;
  const bool fill = options().zero_fill.load() != 0;
  if (vmm_hip_registered()) // once per slot (a rest_direct region did it when it was made)
    for (size_t i = 0; i < n; ++i)
      for (const Row &row : rows_)
        if (!row.r->registered[row.first + (size_t)offsets[i] / ps]) {
          const int64_t t0 = now_ns();
          register_slot(*row.r, row.first + (size_t)offsets[i] / ps);
          stats().t_unmap_alias += now_ns() - t0;
        }
  sg.mark(1);
  const std::vector<SlotRun> runs = scan.collect();
  stats().seg[23] += (int64_t)runs.size();
  sg.mark(2);

  struct Chunk {
    size_t p, k, rows_done;
    bool settled; // the slots name the lanes
    Phys lane[kMaxExtentPages];
  };
  std::vector<Chunk> done;
  std::vector<void *> pending;
  uint64_t max_ticket = 0;
  size_t prescrubbed = 0, ids = 0;
  bool prt_dirty = false, replaced = false, launched = false;
  void *zx_dirty = nullptr;
  DrmVm &vm = DrmVm::instance();
  try {
    for (const SlotRun &run : runs) {
      for (size_t at = run.first - rows_[0].first, end = at + run.count; at < end;) {
        done.emplace_back();
        Chunk &c = done.back();
        c.p = at;
        c.k = c.rows_done = 0;
        c.settled = false;
        bool recycled = false;
        const int64_t t1 = now_ns();
        c.k = pool->acquire_run(end - at, c.lane, &recycled, false);
        if (!c.k && ctx->limbo_bytes()) { // pages of an earlier unmap wait for their invalidation: have it now rather than create
          ctx->flush_limbo();
          c.k = pool->acquire_run(end - at, c.lane, &recycled, false);
        }
        if (!c.k) c.k = pool->acquire_run(end - at, c.lane, &recycled, true);
        const int64_t t2 = now_ns();
        sg.mark(3);
        unsigned j0 = 0, ke = 1;
        void *bo = vmm_direct_bo(c.lane[0].h, &j0, &ke);
        if (!bo) throw GpuError("a lane's buffer is not a direct DRM buffer");
        for (size_t r = 0; r < R; ++r) { // one ioctl per row: pages (r, j0 .. j0 + k) are neighbours in the buffer
          KvRegion &reg = *rows_[r].r;
          char *va = reg.base + (rows_[r].first + at) * ps;
          const uint64_t boff = ((uint64_t)r * ke + j0) * ps;
          int rc;
          if (reg.rest_direct()) { // pages take the place of the rest mapping in the same ioctl
            tlb_stale().store(true);
            rc = vm.replace(bo, va, c.k * ps, boff);
            tlb_stale().store(true); // what was there may sit in a TLB (a zero alias; a PRT entry that was looked at: §4.2)
            replaced = true;
            if (reg.zx) zx_dirty = vm.find(reg.zx_handle);
            if (reg.prt) prt_dirty = true;
          } else {
            rc = vm.map(bo, va, c.k * ps, boff);
          }
          if (rc != 0) throw GpuError(std::string("DRM_AMDGPU_GEM_VA (backing a page id) failed: ") + strerror(rc < 0 ? -rc : rc));
          c.rows_done = r + 1;
        }
        stats().seg[21] += (int64_t)R;
        stats().t_acquire += t2 - t1;
        stats().t_map += now_ns() - t2;
        sg.mark(4);
        for (size_t j = 0; j < c.k; ++j) {
          max_ticket = std::max(max_ticket, c.lane[j].wait_ticket);
          const bool clean = fill && c.lane[j].scrub_ticket != 0;
          if (clean) {
            max_ticket = std::max(max_ticket, c.lane[j].scrub_ticket);
            prescrubbed += R;
          }
          for (size_t r = 0; r < R; ++r) {
            KvRegion &reg = *rows_[r].r;
            const size_t idx = rows_[r].first + at + j;
            reg.handle[idx] = c.lane[j].h;
            reg.seq[idx] = c.lane[j].seq;
            reg.mapped[idx] = 4;
            if (fill && !clean) pending.push_back(reg.base + idx * ps);
          }
        }
        c.settled = true;
        ids += c.k;
        at += c.k;
        sg.mark(5);
      }
    }
    // nothing may reach the new pages through a stale translation of their own addresses: the remainders of what the
    // REPLACEs split are rewritten (their page-table entries still carry the fragment of the whole), then the invalidation
    const int64_t t_rw = now_ns();
    if (prt_dirty && !KVC_TEST_HOOK("SKIP_PRT_REMAINDER_REFRESH") && !vm.refresh_prt_remainders())
      throw GpuError("rewriting the remainders of split PRT mappings failed");
    if (zx_dirty && !vm.refresh_mappings_of(zx_dirty, ps)) KVC_LOG(LOG_ERROR, "rewriting the remaining mappings of the zero extent failed");
    stats().seg[20] += now_ns() - t_rw;
    if (replaced) {
      tlb_stale().store(true);
      need_epoch = ctx->next_flush_epoch();
    }
    if (options().map_waits_for_all_flushes.load() || options().map_shootdown_always.load()) {
      if (options().map_shootdown_always.load()) tlb_stale().store(true);
      ctx->ensure_flushed();
    } else {
      ctx->ensure_flushed_through(need_epoch);
    }
    sg.mark(6);
    const int64_t ts = now_ns();
    for (size_t i = 0; i < pending.size(); i += kMaxPtrsPerLaunch) { // fresh memory, or pages that came back unscrubbed
      ctx->zero_fill(pending.data() + i, std::min<size_t>(kMaxPtrsPerLaunch, pending.size() - i), ps, nullptr);
      launched = true;
    }
    if (launched) ctx->sync(nullptr);
    sg.mark(7);
    ctx->wait_scrub(max_ticket);
    stats().t_sync += now_ns() - ts;
    sg.mark(8);
  } catch (...) {
    if (launched) (void)hipStreamSynchronize(ctx->stream());
    for (auto it = done.rbegin(); it != done.rend(); ++it) {
      for (size_t r = 0; r < it->rows_done; ++r) {
        KvRegion &reg = *rows_[r].r;
        if (reg.rest_direct())
          for_rest_pieces(reg, rows_[r].first + it->p, it->k, [&](size_t first, size_t cnt) { (void)rest_replace(reg, first, cnt); });
        else
          (void)vmm_try_unmap(reg.base + (rows_[r].first + it->p) * ps, it->k * ps, it->lane[0].h);
      }
      if (it->settled)
        for (size_t j = 0; j < it->k; ++j)
          for (const Row &row : rows_) row.r->mapped[row.first + it->p + j] = 0;
      if (it->k) pool->release_batch(it->lane, it->k);
    }
    (void)hipGetLastError();
    try {
      ctx->tlb_shootdown();
    } catch (...) {
    }
    throw;
  }
  stats().pages_mapped += (int64_t)(ids * R);
  if (prescrubbed) stats().pages_prescrubbed += (int64_t)prescrubbed;
  (void)cold;
  if (pool->creations() > 0) cold_start_reserve(pool);
  return true;
}

size_t KvAllocator::unmap_lanes(const offset_t *offsets, size_t n, std::vector<offset_t> *others) {
  const size_t R = rows_.size(), ps = rows_[0].r->page_size;
  SegTimer sg;
  RunScan scan;
  for (size_t i = 0; i < n; ++i) {
    const offset_t off = offsets[i];
    const bool valid = off >= 0 && (size_t)off % ps == 0 && (size_t)off / ps < ids_per_row_;
    const uint8_t state = valid ? rows_[0].r->mapped[rows_[0].first + (size_t)off / ps] : 0;
    if (state == 4 || state == 5) {
      if (!scan.add(rows_[0].r, rows_[0].first + (size_t)off / ps)) KVC_LOG(LOG_ERROR, "Page %zu is not mapped.", (size_t)off / ps); // named twice
    } else {
      others->push_back(off);
    }
  }
  if (!scan.size()) return 0;
  GpuContext *ctx = ctx_;
  ctx->bind();
  ExtentPool *pool = lane_pool_;
  const std::vector<SlotRun> runs = scan.collect();
  sg.mark(11);
  std::vector<Phys> lanes;
  std::vector<phys_handle_t> peers_buffers; // page ids that were backed by a peer's lane: its buffer is released once, after the invalidation
  KeyGroups<uint32_t> touched(n); // buffer -> lanes of it that this call takes back
  bool any_backfilled = false;
  for (const SlotRun &run : runs) {
    const size_t p0 = run.first - rows_[0].first;
    for (size_t r = 0; r < R; ++r) {
      KvRegion &reg = *rows_[r].r;
      const int64_t t0 = now_ns();
      if (reg.rest_direct()) { // back to the rest state, whatever is mapped there, in one ioctl per piece
        for_rest_pieces(reg, rows_[r].first + p0, run.count, [&](size_t first, size_t cnt) {
          const int rc = rest_replace(reg, first, cnt);
          if (rc != 0) throw GpuError(std::string("DRM_AMDGPU_GEM_VA replace (back to the rest state) failed: ") + strerror(rc < 0 ? -rc : rc));
          stats().seg[22]++;
        });
        any_backfilled = any_backfilled || reg.backfilled;
      } else {
        vmm_unmap_run(reg.base + (rows_[r].first + p0) * ps, run.count * ps);
        stats().seg[22]++;
      }
      stats().t_unmap += now_ns() - t0;
    }
    for (size_t p = p0; p < p0 + run.count; ++p) {
      const size_t idx0 = rows_[0].first + p;
      const phys_handle_t h = rows_[0].r->handle[idx0];
      if (rows_[0].r->mapped[idx0] == 5) {
        peers_buffers.push_back(h);
        continue;
      }
      lanes.push_back(Phys{h, rows_[0].r->seq[idx0]});
      ++touched.at(chunk_of(h));
    }
  }
  sg.mark(12);
  for (const SlotRun &run : runs)
    for (const Row &row : rows_) {
      const size_t first = row.first + (run.first - rows_[0].first);
      std::fill(row.r->mapped.begin() + first, row.r->mapped.begin() + first + run.count, 0);
    }
  sg.mark(13);
  // Lanes that were backed together share one mapping per row; taking some of them out splits it, and what is left of it
  // must be written again before the invalidation (DrmVm::refresh_mappings_of, §4.8). A buffer that loses ALL its mapped
  // lanes here leaves nothing behind.
  if (!KVC_TEST_HOOK("SKIP_REMAINDER_REFRESH")) {
    bool any = false;
    for (auto &kv : touched.items())
      if (pool->pieces_out(kv.first) > kv.second)
        if (void *bo = DrmVm::instance().find(kv.first)) {
          if (!DrmVm::instance().refresh_mappings_of(bo, ps)) KVC_LOG(LOG_ERROR, "rewriting the remaining mappings of a buffer failed");
          any = true;
        }
    if (any) tlb_stale().store(true);
  }
  sg.mark(14);
  const uint64_t epoch = ctx->next_flush_epoch(); // every driver call that removed or rewrote a translation has returned
  for (const SlotRun &run : runs)
    for (const Row &row : rows_) {
      const size_t first = row.first + (run.first - rows_[0].first);
      std::fill(row.r->stale_epoch.begin() + first, row.r->stale_epoch.begin() + first + run.count, epoch);
    }
  sg.mark(15);
  // the invalidation: inside the call where unbacked VA promises zeros (compat), behind it otherwise (see unmap_finish)
  const size_t n_ids = lanes.size() + peers_buffers.size();
  if (!peers_buffers.empty()) { // a peer's memory: no translation of it may survive the call
    ctx->ensure_flushed();
    for (auto h : peers_buffers) { // (one entry per page id: the buffer goes when the last page id it backs here has gone)
      auto it = peer_refs_.find(h);
      if (it != peer_refs_.end() && --it->second > 0) continue;
      if (it != peer_refs_.end()) peer_refs_.erase(it);
      if (!vmm_try_release(h)) KVC_LOG(LOG_ERROR, "releasing an imported buffer failed");
    }
    if (lanes.empty()) {
      stats().pages_unmapped += (int64_t)(n_ids * R);
      return n_ids;
    }
  }
  const int64_t trail_us = options().deferred_unmap_flush_us.load();
  if (trail_us > 0 && any_backfilled && peers_buffers.empty()) { // compat, relaxed: see unmap_finish
    auto own = std::make_shared<std::vector<Phys>>(std::move(lanes));
    ctx->park(epoch, own->size() * R * ps, [ctx, pool, own, R, ps]() {
      uint64_t ticket = 0;
      if (options().zero_fill.load() && options().scrub_on_release.load()) {
        std::vector<uint64_t> addrs;
        addrs.reserve(own->size() * R);
        for (const Phys &l : *own)
          if (const uint64_t tag = pool->tag_of(l.h)) {
            const uint64_t ke = pages_of(l.h), j = piece_of(l.h);
            for (size_t r = 0; r < R; ++r) addrs.push_back(tag + ((uint64_t)r * ke + j) * ps);
          }
        if (!addrs.empty()) ticket = ctx->scrub(addrs.data(), addrs.size(), ps);
      }
      pool->release_batch(own->data(), own->size(), ticket);
    });
    ctx->request_async_flush(trail_us);
    sg.mark(16);
    stats().pages_unmapped += (int64_t)(own->size() * R);
    return own->size();
  }
  const bool defer = options().defer_unmap_shootdown.load() || options().async_shootdown.load();
  if (any_backfilled || !defer)
    ctx->ensure_flushed();
  else if (options().defer_unmap_shootdown.load())
    ctx->defer_tlb_shootdown();
  else
    ctx->request_async_flush();
  sg.mark(16);
  const int64_t tr0 = now_ns();
  // zeroed on their way back, through the alias mapping of their buffer: page (row, lane) sits at (row x k + lane)
  uint64_t ticket = 0;
  if (options().zero_fill.load() && options().scrub_on_release.load()) {
    std::vector<uint64_t> addrs;
    addrs.reserve(lanes.size() * R);
    for (const Phys &l : lanes)
      if (const uint64_t tag = pool->tag_of(l.h)) {
        const uint64_t ke = pages_of(l.h), j = piece_of(l.h);
        for (size_t r = 0; r < R; ++r) addrs.push_back(tag + ((uint64_t)r * ke + j) * ps);
      }
    if (!addrs.empty()) ticket = ctx->scrub(addrs.data(), addrs.size(), ps);
  }
  sg.mark(17);
  pool->release_batch(lanes.data(), lanes.size(), ticket);
  sg.mark(18);
  stats().t_release += now_ns() - tr0;
  stats().pages_unmapped += (int64_t)(n_ids * R);
  return n_ids;
}

// ------------------------------------------------------------------ TP shared pool
// hipMemImportFromShareableHandle's `osHandle` convention differs between HIP runtimes: the one
// bundled with PyTorch 2.10+rocm7.0 dereferences it as `int *` (passing the fd by value
// segfaults inside amd::roc::Device::ImportShareableHSAHandle), ROCm 7.2's takes the fd by value
// like CUDA. The pointer form is tried first: a by-value runtime reads the pointer's bits as a
// (huge, invalid) fd number and fails cleanly, after which the by-value form is used.
namespace {
std::atomic<int> g_import_convention{0}; // 0 unknown, 1 pointer to fd, 2 fd by value
std::atomic<int64_t> g_imports_direct{0}, g_imports_runtime{0}; // pages imported straight into KFD + DRM / through ROCr or HIP
}
int64_t import_count(bool direct) { return direct ? g_imports_direct.load() : g_imports_runtime.load(); }
static phys_handle_t import_posix_fd(int fd) {
  if (fd < 0 || fcntl(fd, F_GETFD) == -1) throw InvalidError("import of an invalid file descriptor");
  if (vmm_backend() == kVmmDrm && DrmVm::instance().kfd_ready() && env_bool("KVCACHED_DRM_KFD_IMPORT", true)) {
    // straight into KFD + DRM: mapped with one ioctl like our own pages. A buffer that lives on ANOTHER GPU (rank 0's
    // pages seen from a peer over xGMI) may be refused by this shortcut - ROCr's import, which sets up peer access for
    // the local agent, is the fallback (hsa_amd_vmem_import_shareable_handle + map + set_access below).
    try {
      if (KVC_TEST_HOOK("FAIL_KFD_IMPORT")) throw GpuError("AMDKFD_IOC_IMPORT_DMABUF failed [injected]");
      const phys_handle_t h = DrmVm::instance().import_fd(fd);
      ++g_imports_direct;
      return h;
    } catch (const GpuError &e) {
      static std::atomic<bool> warned{false};
      if (!warned.exchange(true))
        KVC_LOG(LOG_WARNING, "direct import of a shared page failed (%s): importing through ROCr from now on", e.what());
    }
  }
  ++g_imports_runtime;
  if (vmm_uses_rocr()) { // ROCr takes the dmabuf fd by value
    hsa_amd_vmem_alloc_handle_t hh{};
    HSA_CHECK(hsa_amd_vmem_import_shareable_handle(fd, &hh));
    return hh.handle;
  }
  hipMemGenericAllocationHandle_t h{};
  int conv = g_import_convention.load();
  if (conv != 2) {
    alignas(8) static thread_local int slot;
    slot = fd;
    hipError_t st = hipMemImportFromShareableHandle(&h, static_cast<void *>(&slot), hipMemHandleTypePosixFileDescriptor);
    if (st == hipSuccess) {
      g_import_convention = 1;
      return reinterpret_cast<phys_handle_t>(h);
    }
    (void)hipGetLastError();
    int rt = 0;
    (void)hipRuntimeGetVersion(&rt);
    // runtimes before 7.1 are known to dereference: never hand them a small integer as a pointer
    if (conv == 1 || rt < 70100000) hip_check(st, "hipMemImportFromShareableHandle(&h, &fd, posix_fd)", __FILE__, __LINE__);
  }
  HIP_CHECK(hipMemImportFromShareableHandle(&h, reinterpret_cast<void *>(static_cast<uintptr_t>(fd)),
                                            hipMemHandleTypePosixFileDescriptor));
  g_import_convention = 2;
  return reinterpret_cast<phys_handle_t>(h);
}

int KvAllocator::export_mapped_slots(const offset_t *offsets, size_t n, int *out_fds, int64_t cap) {
  std::lock_guard<std::mutex> g(mu_);
  if (!dev_.is_gpu) throw NoGpuError("export_mapped_slots needs a GPU device");
  if (!exportable_) throw InvalidError("handles are not exportable: set KVCACHED_EXPORTABLE_HANDLES=1 before init");
  auto slots = slots_for(offsets, n);
  if ((int64_t)slots.size() > cap) return (int)slots.size();
  ctx_->bind();
  for (auto &s : slots) // nothing is exported unless everything can be
    if (s.region->mapped[s.index] != 1) throw InvalidError("export of a slot that is not backed by a local page");
  int k = 0;
  try {
    for (auto &s : slots) {
      int fd = -1;
      if (vmm_uses_rocr()) {
        if (vmm_backend() == kVmmDrm) fd = DrmVm::instance().export_fd(s.region->handle[s.index]); // -1: a ROCr handle
        if (fd < 0) HSA_CHECK(hsa_amd_vmem_export_shareable_handle(&fd, as_hsa(s.region->handle[s.index]), 0));
      } else {
        HIP_CHECK(hipMemExportToShareableHandle(&fd, as_hip(s.region->handle[s.index]), hipMemHandleTypePosixFileDescriptor, 0));
      }
      out_fds[k++] = fd;
    }
  } catch (...) { // the caller never learns how many were written: close them here
    for (int i = 0; i < k; ++i) ::close(out_fds[i]);
    throw;
  }
  return k;
}

int KvAllocator::export_page_ids(const offset_t *offsets, size_t n, int *out_fds, int64_t *out_meta, int64_t cap) {
  std::lock_guard<std::mutex> g(mu_);
  if (!dev_.is_gpu) throw NoGpuError("export_page_ids needs a GPU device");
  if (!lanes_) throw InvalidError("page ids are not backed as units here (kvc_get_option(129) == 0): export slot by slot");
  if ((int64_t)n > cap) return (int)n;
  const size_t ps = rows_[0].r->page_size;
  for (size_t i = 0; i < n; ++i) { // nothing is exported unless everything can be
    const offset_t off = offsets[i];
    if (off < 0 || (size_t)off % ps != 0 || (size_t)off / ps >= ids_per_row_ || rows_[0].r->mapped[rows_[0].first + (size_t)off / ps] != 4)
      throw InvalidError("export of a page id that is not backed by a lane of this process");
  }
  ctx_->bind();
  KeyGroups<int> bufs(n); // buffer -> its place among the exported fds (order of first appearance)
  int k = 0;
  try {
    for (size_t i = 0; i < n; ++i) {
      const phys_handle_t h = rows_[0].r->handle[rows_[0].first + (size_t)offsets[i] / ps];
      const size_t before = bufs.items().size();
      int &at = bufs.at(chunk_of(h));
      if (bufs.items().size() != before) { // first page id of this buffer: export it (AMDKFD_IOC_EXPORT_DMABUF)
        const int fd = DrmVm::instance().export_fd(chunk_of(h));
        if (fd < 0) throw GpuError("a lane's buffer cannot be exported");
        out_fds[k] = fd;
        at = k++;
      }
      out_meta[3 * i] = at;
      out_meta[3 * i + 1] = (int64_t)pages_of(h);
      out_meta[3 * i + 2] = (int64_t)piece_of(h);
    }
  } catch (...) {
    for (int i = 0; i < k; ++i) ::close(out_fds[i]);
    throw;
  }
  return k;
}

bool KvAllocator::map_imported_page_ids(const offset_t *offsets, size_t n, const int *fds, size_t n_fds, const int64_t *meta) {
  std::lock_guard<std::mutex> g(mu_);
  if (!dev_.is_gpu) throw NoGpuError("map_imported_page_ids needs a GPU device");
  if (num_layers_ == 0) return false;
  if (rows_.empty()) throw InvalidError("page ids cannot be imported as units here (single-row geometry or not the drm backend): import slot by slot");
  if (!n) return true;
  const size_t R = rows_.size(), ps = rows_[0].r->page_size;
  struct Id {
    size_t p;
    uint64_t buf, k, j;
  };
  std::vector<Id> ids(n);
  for (size_t i = 0; i < n; ++i) {
    const offset_t off = offsets[i];
    if (off < 0 || (size_t)off % ps != 0 || (size_t)off / ps >= ids_per_row_) throw InvalidError("offset " + std::to_string(off) + " is not a page id of this geometry");
    const int64_t b = meta[3 * i], k = meta[3 * i + 1], j = meta[3 * i + 2];
    if (b < 0 || (size_t)b >= n_fds || k < 1 || k > (int64_t)kMaxExtentPages || j < 0 || j >= k)
      throw InvalidError("import of a page id: (fd " + std::to_string(b) + ", lane " + std::to_string(j) + " of " + std::to_string(k) + ") names no lane of the buffers that came along");
    ids[i] = Id{(size_t)off / ps, (uint64_t)b, (uint64_t)k, (uint64_t)j};
    for (const Row &row : rows_)
      if (row.r->mapped[row.first + ids[i].p] != 0) throw InvalidError("import into a page id that is backed already");
  }
  std::sort(ids.begin(), ids.end(), [](const Id &a, const Id &b) { return a.p < b.p; });
  for (size_t i = 1; i < n; ++i)
    if (ids[i].p == ids[i - 1].p) throw InvalidError("import of a page id that is named twice");
  GpuContext *ctx = ctx_;
  ctx->bind();
  DrmVm &vm = DrmVm::instance();
  if (!vm.kfd_ready()) throw InvalidError("importing page ids as units needs the drm backend with pages straight from KFD");
  std::vector<phys_handle_t> bufs(n_fds, 0);
  struct Done {
    size_t p, cnt, rows_done;
    phys_handle_t h;
    bool settled;
  };
  std::vector<Done> done;
  bool prt_dirty = false, replaced = false;
  void *zx_dirty = nullptr;
  uint64_t need_epoch = 0;
  try {
    for (size_t f = 0; f < n_fds; ++f) { // AMDKFD_IOC_IMPORT_DMABUF + DRM import, once per buffer
      bufs[f] = vm.import_fd(fds[f]);
      ++g_imports_direct;
    }
    for (size_t i = 0; i < n;) {
      size_t cnt = 1; // page ids that are neighbours here AND in one buffer: one ioctl per row
      while (i + cnt < n && ids[i + cnt].p == ids[i].p + cnt && ids[i + cnt].buf == ids[i].buf && ids[i + cnt].k == ids[i].k &&
             ids[i + cnt].j == ids[i].j + cnt)
        ++cnt;
      const Id &first = ids[i];
      done.push_back(Done{first.p, cnt, 0, bufs[first.buf], false});
      Done &d = done.back();
      void *bo = vm.find(d.h);
      if (!bo) throw GpuError("an imported buffer is not a direct DRM buffer");
      for (size_t r = 0; r < R; ++r) {
        KvRegion &reg = *rows_[r].r;
        const size_t idx = rows_[r].first + first.p;
        for (size_t c = 0; c < cnt; ++c) {
          if (vmm_hip_registered() && !reg.registered[idx + c]) register_slot(reg, idx + c);
          need_epoch = std::max(need_epoch, reg.stale_epoch[idx + c]);
        }
        char *va = reg.base + idx * ps;
        const uint64_t boff = (r * first.k + first.j) * ps;
        int rc;
        if (reg.rest_direct()) {
          tlb_stale().store(true);
          rc = vm.replace(bo, va, cnt * ps, boff);
          tlb_stale().store(true);
          replaced = true;
          if (reg.zx) zx_dirty = vm.find(reg.zx_handle);
          if (reg.prt) prt_dirty = true;
        } else {
          rc = vm.map(bo, va, cnt * ps, boff);
        }
        if (rc != 0) throw GpuError(std::string("DRM_AMDGPU_GEM_VA (mapping a peer's page ids) failed: ") + strerror(rc < 0 ? -rc : rc) +
                                    " - does the buffer hold " + std::to_string(first.k) + " lanes of " + std::to_string(R) + " pages?");
        d.rows_done = r + 1;
      }
      for (const Row &row : rows_)
        for (size_t c = 0; c < cnt; ++c) {
          row.r->handle[row.first + first.p + c] = d.h;
          row.r->seq[row.first + first.p + c] = 0;
          row.r->mapped[row.first + first.p + c] = 5;
        }
      d.settled = true;
      stats().seg[21] += (int64_t)R;
      i += cnt;
    }
    if (prt_dirty && !vm.refresh_prt_remainders()) throw GpuError("rewriting the remainders of split PRT mappings failed");
    if (zx_dirty && !vm.refresh_mappings_of(zx_dirty, ps)) KVC_LOG(LOG_ERROR, "rewriting the remaining mappings of the zero extent failed");
    if (replaced) {
      tlb_stale().store(true);
      need_epoch = ctx->next_flush_epoch();
    }
    ctx->ensure_flushed_through(need_epoch);
  } catch (...) {
    for (auto it = done.rbegin(); it != done.rend(); ++it) {
      for (size_t r = 0; r < it->rows_done; ++r) {
        KvRegion &reg = *rows_[r].r;
        if (reg.rest_direct())
          for_rest_pieces(reg, rows_[r].first + it->p, it->cnt, [&](size_t f0, size_t c) { (void)rest_replace(reg, f0, c); });
        else
          (void)vm.clear(reg.base + (rows_[r].first + it->p) * ps, it->cnt * ps);
      }
      if (it->settled)
        for (const Row &row : rows_)
          for (size_t c = 0; c < it->cnt; ++c) row.r->mapped[row.first + it->p + c] = 0;
    }
    (void)hipGetLastError();
    try {
      ctx->tlb_shootdown();
    } catch (...) {
    }
    for (auto h : bufs)
      if (h) (void)vmm_try_release(h);
    throw;
  }
  for (const Id &id : ids) ++peer_refs_[bufs[id.buf]];
  for (auto h : bufs) // (an fd that came along but backs none of the page ids)
    if (h && !peer_refs_.count(h)) (void)vmm_try_release(h);
  stats().pages_mapped += (int64_t)(n * R);
  return true;
}

bool KvAllocator::map_imported_slots(const offset_t *offsets, size_t n, const int *fds, size_t n_fds) {
  std::lock_guard<std::mutex> g(mu_);
  if (!dev_.is_gpu) throw NoGpuError("map_imported_slots needs a GPU device");
  if (num_layers_ == 0) return false;
  auto slots = slots_for(offsets, n);
  if (slots.size() != n_fds) throw InvalidError("fd count does not match the slot count of the offsets");
  ctx_->bind();
  std::vector<phys_handle_t> hs(n_fds);
  size_t i = 0;
  try {
    for (; i < n_fds; ++i) hs[i] = import_posix_fd(fds[i]);
  } catch (...) {
    for (size_t j = 0; j < i; ++j) (void)vmm_try_release(hs[j]);
    throw;
  }
  // map_slots consumes the handles of the slots it backs (they are released again by its own rollback if the batch
  // fails); what it skips - a slot that is already mapped - or never reaches must not stay imported: every such
  // handle pins a page of the peer's pool for the life of this process.
  std::vector<uint8_t> consumed(n_fds, 0);
  try {
    map_slots(slots, &hs, &consumed);
  } catch (...) {
    for (size_t j = 0; j < n_fds; ++j)
      if (!consumed[j]) (void)vmm_try_release(hs[j]);
    throw;
  }
  for (size_t j = 0; j < n_fds; ++j)
    if (!consumed[j]) (void)vmm_try_release(hs[j]);
  return true;
}

} // namespace kvc
