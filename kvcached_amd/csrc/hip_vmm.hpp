// hip_vmm.hpp — the only file that talks to ROCm's virtual-memory-management APIs.
//
// Two APIs reach the same driver: HIP's (hipMemAddressReserve / hipMemCreate / hipMemMap / hipMemSetAccess /
// hipMemUnmap / hipMemRelease) and ROCr's (hsa_amd_vmem_*), which HIP sits on. Every verb below exists in both
// forms; KVCACHED_VMM_BACKEND picks the combination (drm, hybrid, hip - see the comment above vmm_backend()).
// No CUDA branch anywhere (the reference's csrc/inc/gpu_vmm.hpp is a CUDA/HIP dual shim - this is not).
// Measured costs on MI355X / ROCm 7.2 that shaped the design (profiles/r01_*), per 2 MiB mapping:
//   HIP : create 3.4 us | map 3.1 us | set_access 3.3 us | unmap 12-15 us (10 of them a GPU-marker spin) | release 3-50 us
//   ROCr: create 3.3 us | map 2.3 us | set_access 3.0 us | unmap 2.8 us
//   creation is O(live handles) in ROCr's user space with either; the cost is per MAPPING, not per byte; no call
//   scales across threads of one process; no partial unmap and no working map offset.
//   DRM : (buffer object imported once per handle) map 2.2 us, no set_access | unmap 2.1 us   - DrmVm below
// Hence: recycle handles (PhysPool), never touch a slot twice, one batch per call with the fill kernel overlapped,
// and map/unmap with one GEM_VA ioctl each on VA that HIP has been introduced to once.
#pragma once

#include <hip/hip_runtime.h>
#include <hsa/hsa.h>
#include <hsa/hsa_ext_amd.h>

#include <dirent.h>
#include <dlfcn.h>
#include <fcntl.h>
#include <limits.h>
#include <strings.h>
#include <sys/ioctl.h>
#include <sys/syscall.h>
#include <unistd.h>

#include <algorithm>
#include <atomic>
#include <cerrno>
#include <cstring>
#include <iterator>
#include <functional>
#include <map>
#include <mutex>
#include <set>
#include <string>
#include <unordered_map>
#include <vector>

#include "common.hpp"
#include "extent_pool.hpp"

namespace kvc {

// Reference error text: "<file>:<line> <tok> failed in HIP runtime (<code>): <msg>"
// (csrc/inc/gpu_vmm.hpp:37-45). The reference then abort()s; we throw GpuError so that
// PageAllocator's rollback paths (page_allocator.cpp:215-224, 600-608) can actually run.
// KVCACHED_STRICT_ABORT=1 restores the abort.
inline void hip_check(hipError_t st, const char *tok, const char *file, int line) {
  if (st == hipSuccess) return;
  char buf[512];
  const char *base = strrchr(file, '/');
  snprintf(buf, sizeof buf, "%s:%d %s failed in HIP runtime (%u): %s", base ? base + 1 : file, line, tok,
           (unsigned)st, hipGetErrorString(st));
  (void)hipGetLastError(); // clear the sticky error
  if (env_bool("KVCACHED_STRICT_ABORT", false)) {
    fprintf(stderr, "%s\n", buf);
    std::abort();
  }
  throw GpuError(buf);
}
#define HIP_CHECK(expr) ::kvc::hip_check((expr), #expr, __FILE__, __LINE__)

// ---------------------------------------------------------------------------------------------------------
// Two ways to the same driver. The physical-memory API of HIP sits on ROCr's hsa_amd_vmem_*; on ROCm 7.x
// hipMemUnmap additionally pushes a marker through the GPU queue and spins on its signal - ~10 of its 12-15 us
// (tools/unmap_trace.cpp: 2.7 us of ioctls, the rest user-time spinning inside hsa_signal_wait). Talking to ROCr
// directly, map + set_access + unmap cost 2.3 + 3.0 + 2.8 us per 2 MiB mapping instead of 3.1 + 3.3 + 14.5
// (tools/hsa_vmm_probe.cpp, same box). The price: HIP never learns about such mappings. Kernels (raw pointers),
// device-to-device hipMemcpy, hipMemset and hipPointerGetAttributes work on them; a host<->device hipMemcpy takes
// the pointer for pageable host memory and crashes. Hence the hybrid backend (slots registered with HIP once, every
// torch operation works on the KV tensors) and, on top of it, the default drm backend (DrmVm below); =hip is HIP's API
// alone, the last resort of the fallback chain.
enum : int { kVmmHip = 0, kVmmHybrid = 2, kVmmDrm = 3 }; // (1 was a ROCr-only backend, dropped: HIP copies crashed on its memory)
// kVmmHybrid: VA reserved through HIP and every slot registered with HIP once (hipMemMap of a placeholder handle,
// taken away again through ROCr at once); from then on the slot is backed and unbacked with hsa_amd_vmem_* only.
// HIP keeps resolving the pointer (its copies use the VA, the hardware walks the page tables ROCr wrote): every
// hipMemcpy flavour works at full speed, and map/unmap run at ROCr's speed. See KvAllocator::register_slot.
inline std::atomic<int> &vmm_backend() { // set by KvAllocator::init from KVCACHED_VMM_BACKEND; fixed while handles exist
  static std::atomic<int> v{kVmmHip};
  return v;
}
inline bool vmm_uses_rocr() { return vmm_backend().load() != kVmmHip; }
// kVmmDrm: hybrid, plus this process's own pages - and pages imported from a peer - are mapped with ONE ioctl each
// and unmapped with one ranged ioctl per run of neighbours (DrmVm below) instead of ROCr's export + import + mmap +
// GEM_VA (+ GEM_CLOSE) per call, and are allocated straight from KFD. Zero aliases (compat mode) and the registration
// with HIP stay the hybrid backend's.
inline bool vmm_hip_registered() {
  const int b = vmm_backend().load();
  return b == kVmmHybrid || b == kVmmDrm;
}

inline void hsa_check(hsa_status_t st, const char *tok, const char *file, int line) {
  if (st == HSA_STATUS_SUCCESS) return;
  const char *msg = "unknown";
  (void)hsa_status_string(st, &msg);
  char buf[512];
  const char *base = strrchr(file, '/');
  snprintf(buf, sizeof buf, "%s:%d %s failed in ROCr (0x%x): %s", base ? base + 1 : file, line, tok, (unsigned)st, msg);
  if (env_bool("KVCACHED_STRICT_ABORT", false)) {
    fprintf(stderr, "%s\n", buf);
    std::abort();
  }
  throw GpuError(buf);
}
#define HSA_CHECK(expr) ::kvc::hsa_check((expr), #expr, __FILE__, __LINE__)

// The ROCr agent and its coarse-grained local pool behind a HIP device index (matched by PCI address).
struct HsaDevice {
  hsa_agent_t agent{};
  hsa_amd_memory_pool_t pool{};
  hsa_agent_t cpu{}; // a host agent, for mappings that HIP's host-side copy fallback may dereference
  bool have_cpu = false;
};
inline const HsaDevice &hsa_device(int hip_dev) {
  static std::mutex mu;
  static std::unordered_map<int, HsaDevice> cache;
  std::lock_guard<std::mutex> g(mu);
  auto it = cache.find(hip_dev);
  if (it != cache.end()) return it->second;
  HSA_CHECK(hsa_init()); // HIP has initialised ROCr already: this only takes a reference
  int bus = -1, dev = -1, dom = 0;
  HIP_CHECK(hipDeviceGetAttribute(&bus, hipDeviceAttributePciBusId, hip_dev));
  HIP_CHECK(hipDeviceGetAttribute(&dev, hipDeviceAttributePciDeviceId, hip_dev));
  (void)hipDeviceGetAttribute(&dom, hipDeviceAttributePciDomainID, hip_dev);
  (void)hipGetLastError();
  // bus + device identify the GPU on all but exotic hosts; the PCI domain only breaks ties (older HIP runtimes do not
  // report it, multi-domain hosts need it)
  struct Find {
    int bus, dev, dom;
    bool found = false, exact = false;
    int n_loose = 0, n_exact = 0;
    HsaDevice d;
  } f{bus, dev, dom};
  HSA_CHECK(hsa_iterate_agents(
      [](hsa_agent_t a, void *p) -> hsa_status_t {
        auto *f = static_cast<Find *>(p);
        hsa_device_type_t t;
        if (hsa_agent_get_info(a, HSA_AGENT_INFO_DEVICE, &t) != HSA_STATUS_SUCCESS) return HSA_STATUS_SUCCESS;
        if (t == HSA_DEVICE_TYPE_CPU && !f->d.have_cpu) {
          f->d.cpu = a;
          f->d.have_cpu = true;
        }
        if (t != HSA_DEVICE_TYPE_GPU) return HSA_STATUS_SUCCESS;
        uint32_t bdf = 0, domain = 0;
        (void)hsa_agent_get_info(a, (hsa_agent_info_t)HSA_AMD_AGENT_INFO_BDFID, &bdf);
        (void)hsa_agent_get_info(a, (hsa_agent_info_t)HSA_AMD_AGENT_INFO_DOMAIN, &domain);
        if ((int)((bdf >> 8) & 0xff) != f->bus || (int)((bdf >> 3) & 0x1f) != f->dev) return HSA_STATUS_SUCCESS;
        const bool exact = (int)domain == f->dom;
        ++f->n_loose;
        if (exact) ++f->n_exact;
        if (!f->found || (exact && !f->exact)) {
          f->d.agent = a;
          f->found = true;
          f->exact = exact;
        }
        return HSA_STATUS_SUCCESS;
      },
      &f));
  if (!f.found) throw GpuError("no ROCr agent matches HIP device " + std::to_string(hip_dev));
  // several agents behind one PCI address (partitioned GPU): never guess - the caller falls back to the HIP backend
  if (f.n_exact > 1 || (f.n_exact == 0 && f.n_loose > 1))
    throw GpuError("HIP device " + std::to_string(hip_dev) + " cannot be matched to ONE ROCr agent by PCI address");
  struct Pool {
    bool found = false;
    hsa_amd_memory_pool_t p{};
  } pool;
  HSA_CHECK(hsa_amd_agent_iterate_memory_pools(
      f.d.agent,
      [](hsa_amd_memory_pool_t p, void *q) -> hsa_status_t {
        auto *out = static_cast<Pool *>(q);
        hsa_amd_segment_t seg;
        uint32_t flags = 0;
        bool alloc = false;
        (void)hsa_amd_memory_pool_get_info(p, HSA_AMD_MEMORY_POOL_INFO_SEGMENT, &seg);
        (void)hsa_amd_memory_pool_get_info(p, HSA_AMD_MEMORY_POOL_INFO_GLOBAL_FLAGS, &flags);
        (void)hsa_amd_memory_pool_get_info(p, HSA_AMD_MEMORY_POOL_INFO_RUNTIME_ALLOC_ALLOWED, &alloc);
        if (seg == HSA_AMD_SEGMENT_GLOBAL && alloc && (flags & HSA_AMD_MEMORY_POOL_GLOBAL_FLAG_COARSE_GRAINED) && !out->found) {
          out->p = p;
          out->found = true;
        }
        return HSA_STATUS_SUCCESS;
      },
      &pool));
  if (!pool.found) throw GpuError("ROCr agent of HIP device " + std::to_string(hip_dev) + " has no coarse-grained local pool");
  f.d.pool = pool.p;
  return cache.emplace(hip_dev, f.d).first->second;
}


} // namespace kvc
#include "drm_vm.hpp" // class DrmVm
namespace kvc {

inline hipMemAllocationProp make_alloc_prop(int dev, bool exportable) {
  hipMemAllocationProp prop{};
  prop.type = hipMemAllocationTypePinned;
  prop.requestedHandleType = exportable ? hipMemHandleTypePosixFileDescriptor : hipMemHandleTypeNone;
  prop.location.type = hipMemLocationTypeDevice;
  prop.location.id = dev;
  return prop;
}
inline hipMemAccessDesc make_rw_access(int dev) {
  hipMemAccessDesc d{};
  d.location.type = hipMemLocationTypeDevice;
  d.location.id = dev;
  d.flags = hipMemAccessFlagsProtReadWrite;
  return d;
}
inline hipMemGenericAllocationHandle_t as_hip(phys_handle_t h) { return reinterpret_cast<hipMemGenericAllocationHandle_t>(h); }
inline hsa_amd_vmem_alloc_handle_t as_hsa(phys_handle_t h) { return hsa_amd_vmem_alloc_handle_t{h}; }

// ---- the VMM verbs, in a throwing form (vmm_*) and a quiet one for cleanup / rollback paths (vmm_try_*)
inline void *vmm_reserve(size_t size, size_t align, void *hint) {
  void *p = nullptr;
  HIP_CHECK(hipMemAddressReserve(&p, size, align, hint, 0)); // always through HIP, so that HIP knows the range
  return p;
}
inline bool vmm_try_address_free(void *va, size_t size) {
  return hipMemAddressFree(va, size) == hipSuccess;
}
// `direct`: with the drm backend, also import the handle into DRM so that vmm_map/vmm_unmap take the one-ioctl path
// (false for handles that are mapped many times over - the zero aliases - which stay with ROCr).
// `pages`: how many pages of the pool the buffer holds (an extent; only the drm backend with KFD allocation makes > 1)
inline phys_handle_t vmm_create(int dev, size_t size, bool exportable, bool direct = true, unsigned pages = 1) {
  if (vmm_uses_rocr()) {
    if (direct && vmm_backend() == kVmmDrm && DrmVm::instance().hip_dev() == dev && DrmVm::instance().kfd_ready())
      return DrmVm::instance().create(size, pages); // flat 5 us instead of O(live handles)
    if (pages > 1) throw GpuError("multi-page extents need buffers straight from KFD");
    hsa_amd_vmem_alloc_handle_t h{};
    HSA_CHECK(hsa_amd_vmem_handle_create(hsa_device(dev).pool, size, MEMORY_TYPE_PINNED, 0, &h));
    if (direct && vmm_backend() == kVmmDrm && DrmVm::instance().hip_dev() == dev && !DrmVm::instance().adopt(h.handle)) {
      (void)hsa_amd_vmem_handle_release(h);
      throw GpuError("importing a physical handle into DRM failed (KVCACHED_VMM_BACKEND=drm)");
    }
    return h.handle;
  }
  hipMemGenericAllocationHandle_t h{};
  auto prop = make_alloc_prop(dev, exportable);
  HIP_CHECK(hipMemCreate(&h, size, &prop, 0));
  return reinterpret_cast<phys_handle_t>(h);
}
inline bool vmm_try_release(phys_handle_t h) {
  if (vmm_backend() == kVmmDrm && DrmVm::instance().forget(h)) return true; // a buffer of our own: ROCr never saw it
  if (vmm_uses_rocr()) return hsa_amd_vmem_handle_release(as_hsa(h)) == HSA_STATUS_SUCCESS;
  return hipMemRelease(as_hip(h)) == hipSuccess;
}
// Set by every unmap below, cleared by GpuContext::tlb_shootdown(): "a translation that was valid has been removed and
// the GPU TLBs have not been invalidated since". While it is set, nothing may be mapped-and-touched, and no physical
// page may go back to the driver, without an invalidation first: the VMM calls (HIP's, ROCr's and DRM's alike) leave
// the old translation in the TLBs, and a later map at the same VA would be shadowed by it.
inline std::atomic<bool> &tlb_stale() {
  static std::atomic<bool> v{false};
  return v;
}
// `h`: the handle mapped at `va`, when the caller knows it (needed to undo a direct DRM mapping; 0 = an alias or
// a range, which are always ROCr's / HIP's).
// The flag is raised AFTER the driver call (and before it, for good measure): an invalidation that starts between
// "flag set" and "translation removed" clears the flag without covering this unmap.
struct StaleAfter {
  StaleAfter() { tlb_stale().store(true); }
  ~StaleAfter() { tlb_stale().store(true); }
};
// The DRM buffer object behind a handle and the page of it the handle names (DrmVm::resolve); nullptr: not a direct buffer.
inline void *vmm_direct_bo(phys_handle_t h, unsigned *piece = nullptr, unsigned *pages = nullptr) {
  unsigned p = 0;
  void *bo = vmm_backend() == kVmmDrm ? DrmVm::instance().resolve(h, &p, pages) : nullptr;
  if (piece) *piece = p;
  return bo;
}
// pages of the buffer a handle is a piece of (1: a whole buffer, or not ours to look into)
inline unsigned vmm_extent_pages(phys_handle_t h) {
  unsigned piece = 0, pages = 1;
  return vmm_direct_bo(h, &piece, &pages) ? pages : 1;
}
// `count` pieces of one extent with consecutive indices, starting with `h_first`, behind `count` adjacent slots: ONE ioctl.
inline void vmm_map_pieces(void *va, size_t piece_bytes, size_t count, phys_handle_t h_first) {
  unsigned piece = 0;
  void *bo = vmm_direct_bo(h_first, &piece);
  if (!bo) throw GpuError("vmm_map_pieces: not a direct DRM buffer");
  const int r = DrmVm::instance().map(bo, va, count * piece_bytes, static_cast<uint64_t>(piece) * piece_bytes);
  if (r != 0) throw GpuError(std::string("DRM_AMDGPU_GEM_VA map failed: ") + strerror(r < 0 ? -r : r));
}
// The same over slots that currently show something else (compat mode: aliases of the zero extent): whatever is mapped in
// the range is dropped or split and the pieces take its place, in ONE ioctl (AMDGPU_VA_OP_REPLACE).
// `live`: the translations being replaced may sit in a TLB (zero aliases; PRT entries too once they have been looked at).
inline void vmm_replace_pieces(void *va, size_t piece_bytes, size_t count, phys_handle_t h_first, bool live) {
  struct MaybeStale {
    bool on;
    explicit MaybeStale(bool b) : on(b) {
      if (on) tlb_stale().store(true);
    }
    ~MaybeStale() {
      if (on) tlb_stale().store(true);
    }
  } mark(live);
  unsigned piece = 0;
  void *bo = vmm_direct_bo(h_first, &piece);
  if (!bo) throw GpuError("vmm_replace_pieces: not a direct DRM buffer");
  const int r = DrmVm::instance().replace(bo, va, count * piece_bytes, static_cast<uint64_t>(piece) * piece_bytes);
  if (r != 0) throw GpuError(std::string("DRM_AMDGPU_GEM_VA replace failed: ") + strerror(r < 0 ? -r : r));
}
// Returns whether the mapping still needs vmm_set_access (a DRM mapping is made readable+writable in the same ioctl).
inline bool vmm_map(void *va, size_t size, phys_handle_t h) {
  unsigned piece = 0;
  if (void *bo = vmm_direct_bo(h, &piece)) {
    const int r = DrmVm::instance().map(bo, va, size, static_cast<uint64_t>(piece) * size);
    if (r != 0) throw GpuError(std::string("DRM_AMDGPU_GEM_VA map failed: ") + strerror(r < 0 ? -r : r));
    return false;
  }
  if (vmm_uses_rocr())
    HSA_CHECK(hsa_amd_vmem_map(va, size, 0, as_hsa(h), 0));
  else
    HIP_CHECK(hipMemMap(va, size, 0, as_hip(h), 0));
  return true;
}
inline bool vmm_try_map(void *va, size_t size, phys_handle_t h) {
  unsigned piece = 0;
  if (void *bo = vmm_direct_bo(h, &piece)) return DrmVm::instance().map(bo, va, size, static_cast<uint64_t>(piece) * size) == 0;
  if (vmm_uses_rocr()) return hsa_amd_vmem_map(va, size, 0, as_hsa(h), 0) == HSA_STATUS_SUCCESS;
  return hipMemMap(va, size, 0, as_hip(h), 0) == hipSuccess;
}
inline void vmm_set_access(void *va, size_t size, int dev) {
  if (vmm_uses_rocr()) {
    const HsaDevice &hd = hsa_device(dev);
    hsa_amd_memory_access_desc_t d{HSA_ACCESS_PERMISSION_RW, hd.agent}; // HIP resolves the pointers itself (registered VA)
    HSA_CHECK(hsa_amd_vmem_set_access(va, size, &d, 1));
  } else {
    const auto acc = make_rw_access(dev);
    HIP_CHECK(hipMemSetAccess(va, size, &acc, 1));
  }
}
inline bool vmm_try_set_access(void *va, size_t size, int dev) {
  if (vmm_uses_rocr()) {
    try {
      const HsaDevice &hd = hsa_device(dev);
      hsa_amd_memory_access_desc_t d{HSA_ACCESS_PERMISSION_RW, hd.agent};
      return hsa_amd_vmem_set_access(va, size, &d, 1) == HSA_STATUS_SUCCESS;
    } catch (...) {
      return false;
    }
  }
  const auto acc = make_rw_access(dev);
  return hipMemSetAccess(va, size, &acc, 1) == hipSuccess;
}
inline void vmm_unmap(void *va, size_t size, phys_handle_t h = 0) {
  StaleAfter mark;
  if (void *bo = h ? vmm_direct_bo(h) : nullptr) {
    const int r = DrmVm::instance().unmap(bo, va, size);
    if (r != 0) throw GpuError(std::string("DRM_AMDGPU_GEM_VA unmap failed: ") + strerror(r < 0 ? -r : r));
    return;
  }
  if (vmm_uses_rocr())
    HSA_CHECK(hsa_amd_vmem_unmap(va, size));
  else
    HIP_CHECK(hipMemUnmap(va, size));
}
// One call for a run of slots that are all direct DRM mappings of ours (drm backend only; the caller checked).
inline void vmm_unmap_run(void *va, size_t size) {
  StaleAfter mark;
  const int r = DrmVm::instance().clear(va, size);
  if (r != 0) throw GpuError(std::string("DRM_AMDGPU_GEM_VA clear failed: ") + strerror(r < 0 ? -r : r));
}
inline bool vmm_try_unmap(void *va, size_t size, phys_handle_t h = 0) {
  StaleAfter mark;
  if (void *bo = h ? vmm_direct_bo(h) : nullptr) return DrmVm::instance().unmap(bo, va, size) == 0;
  if (vmm_uses_rocr()) return hsa_amd_vmem_unmap(va, size) == HSA_STATUS_SUCCESS;
  const bool ok = hipMemUnmap(va, size) == hipSuccess;
  if (!ok) (void)hipGetLastError();
  return ok;
}

// Test hook: when >= 0, the (n+1)-th hipMemCreate from now fails with hipErrorOutOfMemory (option 104).
inline std::atomic<int64_t> &fail_after_creates() {
  static std::atomic<int64_t> v{-1};
  return v;
}

} // namespace kvc
