// hip_vmm.hpp — the only file that talks to ROCm's virtual-memory-management APIs.
//
// Two APIs reach the same driver: HIP's (hipMemAddressReserve / hipMemCreate / hipMemMap / hipMemSetAccess /
// hipMemUnmap / hipMemRelease) and ROCr's (hsa_amd_vmem_*), which HIP sits on. Every verb below exists in both
// forms; KVCACHED_VMM_BACKEND picks the combination (drm, hybrid, hip, hsa - see the comment above vmm_backend()).
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
// torch operation works on the KV tensors) and, on top of it, the default drm backend (DrmVm below);
// KVCACHED_VMM_BACKEND=hsa is for engines that touch KV memory from kernels only, =hip is HIP's API alone.
using phys_handle_t = uint64_t; // hipMemGenericAllocationHandle_t (a pointer), hsa_amd_vmem_alloc_handle_t::handle, or
                                // (drm backend, pages allocated straight from KFD) KFD's buffer handle

// Piece ids (drm backend, KVCACHED_PHYS_CHUNK_PAGES > 1): physical memory is allocated in chunks of k pages and a
// handle names ONE page-sized piece of a chunk: the chunk's buffer handle (KFD handles are < 2^48) with the piece index
// in the top byte. With k = 1 - every other backend, and the default - a piece id IS the handle.
constexpr int kPieceShift = 56;
inline phys_handle_t piece_id(phys_handle_t chunk, unsigned piece) { return chunk | (static_cast<uint64_t>(piece) << kPieceShift); }
inline phys_handle_t chunk_of(phys_handle_t h) { return h & ((1ull << kPieceShift) - 1); }
inline unsigned piece_of(phys_handle_t h) { return static_cast<unsigned>(h >> kPieceShift); }

enum : int { kVmmHip = 0, kVmmHsa = 1, kVmmHybrid = 2, kVmmDrm = 3 };
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
// hsa backend only. true (default): every mapping is also made accessible to the CPU agent. HIP, which takes a
// pointer it has never seen for host memory, then really can read and write it (through the PCIe BAR: host->device
// 11 GB/s, device->host 24 MB/s, coherent in tools/hsa_vmm_probe.cpp) instead of crashing. Costs 2.6 us per mapping
// (set_access 3.1 -> 4.1, unmap 2.8 -> 4.4). false: kernels only.
inline std::atomic<int> &hsa_cpu_access() {
  static std::atomic<int> v{1};
  return v;
}
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
  if (vmm_backend() == kVmmHsa) // hybrid: through HIP, so that HIP knows the range
    HSA_CHECK(hsa_amd_vmem_address_reserve_align(&p, size, reinterpret_cast<uint64_t>(hint), align, 0));
  else
    HIP_CHECK(hipMemAddressReserve(&p, size, align, hint, 0));
  return p;
}
inline bool vmm_try_address_free(void *va, size_t size) {
  if (vmm_backend() == kVmmHsa) return hsa_amd_vmem_address_free(va, size) == HSA_STATUS_SUCCESS;
  return hipMemAddressFree(va, size) == hipSuccess;
}
// `direct`: with the drm backend, also import the handle into DRM so that vmm_map/vmm_unmap take the one-ioctl path
// (false for handles that are mapped many times over - the zero aliases - which stay with ROCr).
inline phys_handle_t vmm_create(int dev, size_t size, bool exportable, bool direct = true) {
  if (vmm_uses_rocr()) {
    if (direct && vmm_backend() == kVmmDrm && DrmVm::instance().hip_dev() == dev && DrmVm::instance().kfd_ready())
      return DrmVm::instance().create(size); // flat 5 us instead of O(live handles)
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
inline void *vmm_direct_bo(phys_handle_t h) { return vmm_backend() == kVmmDrm ? DrmVm::instance().find(chunk_of(h)) : nullptr; }
// `count` pieces of one chunk with consecutive indices, starting with `h_first`, behind `count` adjacent slots: ONE ioctl.
inline void vmm_map_pieces(void *va, size_t piece_bytes, size_t count, phys_handle_t h_first) {
  void *bo = vmm_direct_bo(h_first);
  if (!bo) throw GpuError("vmm_map_pieces: not a direct DRM buffer");
  const int r = DrmVm::instance().map(bo, va, count * piece_bytes, static_cast<uint64_t>(piece_of(h_first)) * piece_bytes);
  if (r != 0) throw GpuError(std::string("DRM_AMDGPU_GEM_VA map failed: ") + strerror(r < 0 ? -r : r));
}
// Returns whether the mapping still needs vmm_set_access (a DRM mapping is made readable+writable in the same ioctl).
inline bool vmm_map(void *va, size_t size, phys_handle_t h) {
  if (void *bo = vmm_direct_bo(h)) {
    const int r = DrmVm::instance().map(bo, va, size, static_cast<uint64_t>(piece_of(h)) * size);
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
  if (void *bo = vmm_direct_bo(h)) return DrmVm::instance().map(bo, va, size, static_cast<uint64_t>(piece_of(h)) * size) == 0;
  if (vmm_uses_rocr()) return hsa_amd_vmem_map(va, size, 0, as_hsa(h), 0) == HSA_STATUS_SUCCESS;
  return hipMemMap(va, size, 0, as_hip(h), 0) == hipSuccess;
}
inline void vmm_set_access(void *va, size_t size, int dev) {
  if (vmm_uses_rocr()) {
    const HsaDevice &hd = hsa_device(dev);
    hsa_amd_memory_access_desc_t d[2] = {{HSA_ACCESS_PERMISSION_RW, hd.agent}, {HSA_ACCESS_PERMISSION_RW, hd.cpu}};
    HSA_CHECK(hsa_amd_vmem_set_access(va, size, d, hsa_cpu_access().load() && hd.have_cpu ? 2 : 1));
  } else {
    const auto acc = make_rw_access(dev);
    HIP_CHECK(hipMemSetAccess(va, size, &acc, 1));
  }
}
inline bool vmm_try_set_access(void *va, size_t size, int dev) {
  if (vmm_uses_rocr()) {
    try {
      const HsaDevice &hd = hsa_device(dev);
      hsa_amd_memory_access_desc_t d[2] = {{HSA_ACCESS_PERMISSION_RW, hd.agent}, {HSA_ACCESS_PERMISSION_RW, hd.cpu}};
      return hsa_amd_vmem_set_access(va, size, d, hsa_cpu_access().load() && hd.have_cpu ? 2 : 1) == HSA_STATUS_SUCCESS;
    } catch (...) {
      return false;
    }
  }
  const auto acc = make_rw_access(dev);
  return hipMemSetAccess(va, size, &acc, 1) == hipSuccess;
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

struct VmmCounters {
  std::atomic<int64_t> created{0}, released{0}, reused{0};
};

// A physical allocation plus the order in which it was created. The order matters on ROCm:
// hipMemRelease walks a creation-ordered list from its head (3.7 us for the oldest handles, 46 us
// for the newest, tools/create_diag.cpp) and the hole left at the head makes later hipMemCreate
// calls O(1) instead of O(live handles). Hence: always release oldest-first.
struct Phys {
  phys_handle_t h{};
  uint64_t seq = 0;
};

// A bounded set of idle physical allocations of one size on one device. hipMemCreate costs
// O(live allocations) on ROCm (4 us at 1k live handles, 70-200 us at 20-28k), so recycling is what
// keeps the map path flat (~10 us/page). Idle handles hold HBM, so the pool is bounded
// (KVCACHED_PHYS_POOL_MB, default 16384 = 5.5 % of the HBM; trim() empties it) and pressure-aware: when the device's
// free memory is below the allocator's own headroom (1 - KVCACHED_GPU_UTILIZATION of the total),
// released handles go straight back to the driver like the reference's GPUPage destructor
// (csrc/page.cpp:17) and the idle ones are drained (also checked at 10 Hz by the allocator's watcher thread,
// so an idle engine does not starve a co-located one). Eviction is oldest-created-first.
class PhysPool {
public:
  // `unit`: pages per handle (1, or k when the handles are chunks carved up by a PiecePool): the created/released
  // counters stay in pages; reuse is then counted per piece by the PiecePool.
  PhysPool(int dev, size_t granule, bool exportable, VmmCounters *ctr, unsigned unit = 1)
      : dev_(dev), granule_(granule), exportable_(exportable), ctr_(ctr), unit_(unit) {}
  ~PhysPool() { drain(0); }

  size_t granule() const { return granule_; }
  bool exportable() const { return exportable_; }
  void set_cap_bytes(size_t b) { cap_handles_ = b / granule_; }
  // Called once before a batch of handles is given back to the driver (their physical memory leaves this
  // process): the owner uses it to flush a TLB invalidation it had deferred.
  void set_before_driver_release(std::function<void()> fn) { before_driver_release_ = std::move(fn); }

  // An idle handle if there is one (never creates).
  bool try_acquire_idle(Phys *out) {
    std::lock_guard<std::mutex> g(mu_);
    if (idle_.empty()) return false;
    auto it = std::prev(idle_.end());
    *out = Phys{it->second, it->first};
    idle_.erase(it);
    low_water_ = std::min(low_water_, idle_.size());
    if (unit_ == 1) ctr_->reused++;
    return true;
  }

  // `recycled` tells the caller whether the memory may hold old data.
  Phys acquire(bool *recycled) {
    {
      std::lock_guard<std::mutex> g(mu_);
      if (!idle_.empty()) {
        auto it = std::prev(idle_.end()); // youngest: the old ones stay cheap to give back
        Phys p{it->second, it->first};
        idle_.erase(it);
        low_water_ = std::min(low_water_, idle_.size());
        if (unit_ == 1) ctr_->reused++;
        *recycled = true;
        return p;
      }
    }
    Phys p;
    if (fail_after_creates().load() >= 0 && fail_after_creates().fetch_sub(1) == 0) // fault injection (tests)
      hip_check(hipErrorOutOfMemory, "hipMemCreate(&p.h, granule_, &prop, 0) [injected]", __FILE__, __LINE__);
    p.h = vmm_create(dev_, granule_, exportable_);
    p.seq = next_seq_.fetch_add(1) + 1;
    ctr_->created += unit_;
    *recycled = false;
    return p;
  }

  void release(Phys p) { release_batch(&p, 1); }

  // Takes handles back. What exceeds the cap (or everything, under memory pressure) goes to the
  // driver in creation order, oldest first.
  // With a housekeeping thread around (defer_eviction), what exceeds the cap is NOT released here, on the
  // caller's free() path - hipMemRelease of a used handle costs 40-50 us, 200 ms for a 4096-slot free - but
  // by trim_to_cap() from that thread, a tick later. Under memory pressure the release is immediate either way.
  void set_defer_eviction(bool on) { defer_eviction_.store(on); }
  void release_batch(Phys *ps, size_t n) {
    if (n == 0) return;
    std::vector<Phys> victims;
    const bool pressure = under_pressure();
    {
      std::lock_guard<std::mutex> g(mu_);
      for (size_t i = 0; i < n; ++i) idle_.emplace(ps[i].seq, ps[i].h);
      const size_t keep = pressure ? 0 : (defer_eviction_.load() && cap_handles_ > 0 ? (size_t)-1 : cap_handles_);
      while (idle_.size() > keep) {
        auto it = idle_.begin();
        victims.push_back(Phys{it->second, it->first});
        idle_.erase(it);
      }
    }
    to_driver(victims);
  }

  // Housekeeping: bring the idle set back under its cap, at most `max_release` handles per call, oldest first.
  size_t trim_to_cap(size_t max_release) {
    std::vector<Phys> victims;
    {
      std::lock_guard<std::mutex> g(mu_);
      while (idle_.size() > cap_handles_ && victims.size() < max_release) {
        auto it = idle_.begin();
        victims.push_back(Phys{it->second, it->first});
        idle_.erase(it);
      }
      low_water_ = std::min(low_water_, idle_.size());
    }
    to_driver(victims);
    return victims.size();
  }

  // Give idle memory back to the driver, keeping at most `keep` (the youngest) handles.
  void drain(size_t keep) {
    std::vector<Phys> victims;
    {
      std::lock_guard<std::mutex> g(mu_);
      while (idle_.size() > keep) {
        auto it = idle_.begin();
        victims.push_back(Phys{it->second, it->first});
        idle_.erase(it);
      }
    }
    to_driver(victims);
  }

  // Idle-time decay (called from the allocator's 10 Hz watcher): the handles nobody needed during a whole
  // window of `idle_ns` - the low-water mark of the idle set over that window - go back to the driver, at
  // most `max_release` per call. The pool is a recycling buffer for alloc/free churn, not a place to keep
  // memory: a co-located engine computes what it may use from hipMemGetInfo and would never see what is
  // parked here (the reference releases on every unmap, csrc/page.cpp:17).
  size_t decay(int64_t now_ns, int64_t idle_ns, size_t max_release) {
    std::vector<Phys> victims;
    {
      std::lock_guard<std::mutex> g(mu_);
      if (window_start_ns_ == 0 || idle_.empty()) {
        window_start_ns_ = now_ns;
        low_water_ = idle_.size();
        return 0;
      }
      if (now_ns - window_start_ns_ < idle_ns) return 0;
      const size_t surplus = std::min(low_water_, idle_.size());
      const size_t n = std::min(surplus, max_release);
      for (size_t i = 0; i < n; ++i) {
        auto it = idle_.begin();
        victims.push_back(Phys{it->second, it->first});
        idle_.erase(it);
      }
      if (n < surplus) {
        low_water_ = surplus - n; // keep going at the next tick
      } else {
        window_start_ns_ = now_ns;
        low_water_ = idle_.size();
      }
    }
    to_driver(victims);
    return victims.size();
  }

  // hipMemGetInfo is ~0.25 us on MI355X: cheap enough to ask on every release batch.
  bool under_pressure() const {
    size_t free_b = 0, total_b = 0;
    if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) {
      (void)hipGetLastError();
      return false;
    }
    static const double util = []() {
      const char *e = std::getenv("KVCACHED_GPU_UTILIZATION");
      return e ? std::atof(e) : 0.95;
    }();
    return free_b < static_cast<size_t>(total_b * (1.0 - util));
  }
  size_t idle_count() {
    std::lock_guard<std::mutex> g(mu_);
    return idle_.size();
  }

  // Release handles that never enter the pool (imports, teardown), oldest first.
  void to_driver(std::vector<Phys> &v) {
    if (v.empty()) return;
    if (before_driver_release_) before_driver_release_();
    std::sort(v.begin(), v.end(), [](const Phys &a, const Phys &b) { return a.seq < b.seq; });
    for (auto &p : v) {
      if (!vmm_try_release(p.h)) KVC_LOG(LOG_ERROR, "releasing a physical handle failed");
      ctr_->released += unit_;
    }
    (void)hipGetLastError();
  }

private:
  int dev_;
  size_t granule_;
  bool exportable_;
  VmmCounters *ctr_;
  unsigned unit_ = 1;
  size_t cap_handles_ = 0;
  std::function<void()> before_driver_release_;
  std::atomic<bool> defer_eviction_{false};
  std::atomic<uint64_t> next_seq_{0};
  std::mutex mu_;
  std::multimap<uint64_t, phys_handle_t> idle_; // creation order -> handle
  size_t low_water_ = 0;                         // smallest idle_ size since window_start_ns_
  int64_t window_start_ns_ = 0;
};

// Page-sized pieces out of a PhysPool of chunks (k pages each; k = 1: a plain pass-through, which is the default and
// the only mode of the non-drm backends). GEM_VA maps at an offset into a buffer, so any piece can back any slot, and a
// run of adjacent slots backed by adjacent pieces of one chunk is ONE ioctl (tools/drm_chunk_probe.cpp: 0.18 us per
// page for runs of 16 against 2.2). Whole idle chunks live in the PhysPool (cap, decay, pressure, eviction all work on
// chunks); chunks with some pieces out are tracked here and go back to it when their last piece returns. Single pieces
// are served from partly used chunks first, so that whole chunks stay whole.
class PiecePool {
public:
  PiecePool(PhysPool *chunks, size_t piece_bytes, unsigned k, VmmCounters *ctr)
      : chunks_(chunks), piece_bytes_(piece_bytes), k_(k), ctr_(ctr), full_(k >= 64 ? ~0ull : ((1ull << k) - 1)) {}
  PhysPool *chunks() const { return chunks_; }
  unsigned pieces_per_chunk() const { return k_; }

  // Up to `want` pieces with consecutive indices in one chunk; at least one unless nothing is idle and !may_create
  // (then 0). *recycled: the memory may hold old data.
  size_t acquire_run(size_t want, Phys *out, bool *recycled, bool may_create) {
    if (k_ == 1) {
      if (may_create) {
        out[0] = chunks_->acquire(recycled);
        return 1;
      }
      *recycled = true;
      return chunks_->try_acquire_idle(&out[0]) ? 1 : 0;
    }
    want = std::min<size_t>(std::max<size_t>(want, 1), k_);
    {
      std::lock_guard<std::mutex> g(mu_);
      size_t looked = 0;
      for (auto it = partial_.begin(); it != partial_.end() && looked < 8; ++it, ++looked) { // a partly used chunk with room
        Chunk &c = tracked_[*it];
        const int first = find_run(c.free_mask, (unsigned)want);
        if (first < 0) continue;
        const phys_handle_t h = *it;
        const unsigned old = take(h, c, (unsigned)first, (unsigned)want, out);
        *recycled = old > 0;
        ctr_->reused += old;
        return want;
      }
    }
    // No partly used chunk has `want` neighbours free. A whole idle chunk costs nothing new; failing that, whatever
    // free pieces exist are used up - a shorter run now, the caller comes back for the rest - BEFORE a new chunk is
    // made: physical memory allocated never exceeds what is mapped by more than one chunk's worth of pieces
    // (opening a chunk per run tripled the footprint of the Poisson workload).
    Phys c;
    bool rec = true;
    if (!chunks_->try_acquire_idle(&c)) {
      {
        std::lock_guard<std::mutex> g(mu_);
        if (!partial_.empty()) {
          const phys_handle_t h = *partial_.begin();
          Chunk &pc = tracked_[h];
          unsigned n = (unsigned)want;
          int first = -1;
          while (n >= 1 && (first = find_run(pc.free_mask, n)) < 0) --n; // the longest run this chunk still has, up to `want`
          const unsigned old = take(h, pc, (unsigned)first, n, out);
          *recycled = old > 0;
          ctr_->reused += old;
          return n;
        }
      }
      if (!may_create) return 0;
      c = chunks_->acquire(&rec); // may throw: nothing of ours has changed yet
    }
    std::lock_guard<std::mutex> g(mu_);
    Chunk &t = tracked_[c.h];
    t.seq = c.seq;
    t.free_mask = full_;
    t.used_mask = rec ? full_ : 0; // a chunk from the idle pool has been used all over, a new one nowhere
    free_pieces_ += k_;
    const unsigned old = take(c.h, t, 0, (unsigned)want, out);
    *recycled = old > 0;
    ctr_->reused += old;
    return want;
  }

  void release(Phys p) { release_batch(&p, 1); }
  void release_batch(Phys *ps, size_t n) {
    if (n == 0) return;
    if (k_ == 1) {
      chunks_->release_batch(ps, n);
      return;
    }
    std::vector<Phys> whole;
    {
      std::lock_guard<std::mutex> g(mu_);
      for (size_t i = 0; i < n; ++i) {
        const phys_handle_t h = chunk_of(ps[i].h);
        auto it = tracked_.find(h);
        if (it == tracked_.end()) {
          KVC_LOG(LOG_ERROR, "a piece of an unknown chunk was released");
          continue;
        }
        it->second.free_mask |= 1ull << piece_of(ps[i].h);
        ++free_pieces_;
        if (it->second.free_mask == full_) {
          free_pieces_ -= k_; // the chunk goes back whole: counted by the chunk pool from here on
          whole.push_back(Phys{h, it->second.seq});
          partial_.erase(h);
          tracked_.erase(it);
        } else {
          partial_.insert(h);
        }
      }
    }
    if (!whole.empty()) chunks_->release_batch(whole.data(), whole.size());
  }
  // bytes of free pieces inside partly used chunks: ours to reuse, invisible to hipMemGetInfo
  size_t free_piece_bytes() {
    if (k_ == 1) return 0;
    std::lock_guard<std::mutex> g(mu_);
    return free_pieces_ * piece_bytes_;
  }

private:
  struct Chunk {
    uint64_t seq = 0;
    uint64_t free_mask = 0;
    uint64_t used_mask = 0; // pieces that have been handed out before (their memory may hold old data)
  };
  // lowest start of `want` consecutive free pieces, or -1
  int find_run(uint64_t mask, unsigned want) const {
    uint64_t m = mask;
    for (unsigned s = 1; s < want && m; ++s) m &= mask >> s;
    return m ? __builtin_ctzll(m) : -1;
  }
  // returns how many of the pieces taken had been in use before
  unsigned take(phys_handle_t h, Chunk &c, unsigned first, unsigned n, Phys *out) {
    unsigned old = 0;
    for (unsigned i = 0; i < n; ++i) {
      const uint64_t bit = 1ull << (first + i);
      c.free_mask &= ~bit;
      --free_pieces_;
      old += (c.used_mask & bit) != 0;
      c.used_mask |= bit;
      out[i] = Phys{piece_id(h, first + i), c.seq};
    }
    if (c.free_mask)
      partial_.insert(h);
    else
      partial_.erase(h);
    return old;
  }
  PhysPool *chunks_;
  size_t piece_bytes_;
  unsigned k_;
  VmmCounters *ctr_;
  uint64_t full_;
  std::mutex mu_;
  size_t free_pieces_ = 0;                            // free pieces inside tracked chunks
  std::unordered_map<phys_handle_t, Chunk> tracked_; // chunks with at least one piece handed out
  std::set<phys_handle_t> partial_;                   // ... of which some pieces are free
};

} // namespace kvc
