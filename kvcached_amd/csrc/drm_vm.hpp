// drm_vm.hpp — the process's GPU address space edited through libdrm_amdgpu (DRM_AMDGPU_GEM_VA) and physical memory
// allocated / exported / imported through KFD's ioctls directly: what the default `drm` VMM backend is made of
// (DESIGN.md §4.7/§4.8). Included by hip_vmm.hpp, which supplies phys_handle_t, GpuError and the logging macros.
#pragma once

namespace kvc {

// KFD's id of the device behind a PCI address (sysfs topology); 0: not found, or several KFD nodes behind one address
// (a partitioned GPU): never guess.
inline uint32_t kfd_gpu_id_for(unsigned domain, unsigned bus, unsigned dev, unsigned fn) {
  const unsigned long long want = (bus << 8) | (dev << 3) | fn;
  uint32_t found = 0;
  for (int n = 0; n < 256; n++) {
    char path[128];
    snprintf(path, sizeof path, "/sys/class/kfd/kfd/topology/nodes/%d/gpu_id", n);
    FILE *f = fopen(path, "r");
    if (!f) break;
    unsigned long id = 0;
    if (fscanf(f, "%lu", &id) != 1) id = 0;
    fclose(f);
    if (!id) continue; // a CPU node
    snprintf(path, sizeof path, "/sys/class/kfd/kfd/topology/nodes/%d/properties", n);
    if (!(f = fopen(path, "r"))) continue;
    char key[64];
    unsigned long long val, loc = ~0ull, dom = 0;
    while (fscanf(f, "%63s %llu", key, &val) == 2) {
      if (!strcmp(key, "location_id")) loc = val;
      if (!strcmp(key, "domain")) dom = val;
    }
    fclose(f);
    if (loc == want && dom == domain) {
      if (found) return 0;
      found = (uint32_t)id;
    }
  }
  return found;
}
inline uint32_t kfd_gpu_id_of_hip_device(int hip_dev) {
  char bdf[64] = {0};
  unsigned dom = 0, bus = 0, dv = 0, fn = 0;
  if (hipDeviceGetPCIBusId(bdf, sizeof bdf, hip_dev) != hipSuccess || sscanf(bdf, "%x:%x:%x.%x", &dom, &bus, &dv, &fn) != 4) {
    (void)hipGetLastError();
    return 0;
  }
  return kfd_gpu_id_for(dom, bus, dv, fn);
}

// ---- KfdTlbFlush: make the kernel invalidate this GPU's TLBs for our process - deterministically.
// ROCm's VMM calls (HIP's, ROCr's, DRM's GEM_VA) leave stale translations behind; what flushes is KFD's
// unmap path (kfd_ioctl_unmap_memory_from_gpu ends in a heavyweight flush of the process's VM on that GPU). Round 1
// reached it through hipMalloc(2 MiB) + hipFree and watched the clock to guess whether the runtime had served the block
// from a cache. This is the ioctl pair itself, on a 4 KiB buffer of our own at a VA nothing else will ever use, on our
// own fd of /dev/kfd (every open of /dev/kfd by a process attaches to the same kfd_process, so it is ROCr's address
// space): AMDKFD_IOC_UNMAP_MEMORY_FROM_GPU + AMDKFD_IOC_MAP_MEMORY_TO_GPU (in that order: see flush()). No user-space
// cache can answer it; it costs what the flush costs (0.18-0.24 ms; tools/drm_vmm_probe.cpp mode 1: 0 wrong words where
// no flush gives all wrong).
class KfdTlbFlush {
public:
  KfdTlbFlush() = default;
  KfdTlbFlush(const KfdTlbFlush &) = delete;
  ~KfdTlbFlush() { close(); }
  bool ready() const { return handle_ != 0; }
  bool open(int hip_dev, std::string *why) {
    if (ready()) return true;
    gpu_id_ = kfd_gpu_id_of_hip_device(hip_dev);
    if (!gpu_id_) {
      *why = "no single KFD topology node for the device";
      return false;
    }
    fd_ = ::open("/dev/kfd", O_RDWR | O_CLOEXEC);
    if (fd_ < 0) {
      *why = std::string("cannot open /dev/kfd: ") + strerror(errno);
      return false;
    }
    if (hipMemAddressReserve(&va_, kVaBytes, kVaBytes, nullptr, 0) != hipSuccess) {
      (void)hipGetLastError();
      *why = "no VA for the flush buffer";
      close();
      return false;
    }
    Alloc a{};
    a.va_addr = reinterpret_cast<uint64_t>(va_);
    a.size = 4096;
    a.gpu_id = gpu_id_;
    a.flags = (1u << 31) | (1u << 28) | 1u; // VRAM | WRITABLE | NO_SUBSTITUTE
    if (call(kAlloc, &a) != 0) {
      *why = std::string("AMDKFD_IOC_ALLOC_MEMORY_OF_GPU for the flush buffer: ") + strerror(errno);
      close();
      return false;
    }
    handle_ = a.handle;
    if (!flush()) { // once, so that a kernel that refuses the pair is found out now
      *why = std::string("AMDKFD_IOC_MAP/UNMAP_MEMORY of the flush buffer: ") + strerror(errno);
      close();
      return false;
    }
    return true;
  }
  // UNMAP, then MAP again: the buffer rests MAPPED. KFD's unmap ends in a heavyweight flush of the process's address
  // space on this GPU, XCC by XCC - that is the invalidation. KFD's map ends in a (legacy) flush too whenever the
  // address space has seen an update that asks for one since KFD last flushed, i.e. after every one of our REPLACE
  // ioctls: with the pair in the order MAP, UNMAP each half paid a flush (2 x 0.2 ms, measured: the MAP half was half of
  // the 0.39 ms); in this order the MAP finds nothing new - a mapping that becomes valid asks for no flush - and takes
  // microseconds. `remap_ns`: what the MAP half took (diagnostics).
  bool flush(int64_t *remap_ns = nullptr) {
    if (!ready()) return false;
    Map m{handle_, reinterpret_cast<uint64_t>(&gpu_id_), 1, 0};
    if (!mapped_) { // (first use, or a MAP that failed after the previous flush)
      if (call(kMap, &m) != 0) return false;
      mapped_ = true;
      m.n_success = 0;
    }
    if (call(kUnmap, &m) != 0) return false; // returns after the flush
    mapped_ = false;
    const auto t0 = std::chrono::steady_clock::now();
    m.n_success = 0;
    if (call(kMap, &m) == 0) mapped_ = true; // (a failure here only makes the next flush do it first)
    if (remap_ns) *remap_ns = std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - t0).count();
    return true;
  }
  void close() {
    if (handle_) {
      if (mapped_) {
        Map m{handle_, reinterpret_cast<uint64_t>(&gpu_id_), 1, 0};
        (void)call(kUnmap, &m);
        mapped_ = false;
      }
      Free f{handle_};
      (void)call(kFree, &f);
      handle_ = 0;
    }
    if (va_) {
      (void)hipMemAddressFree(va_, kVaBytes);
      (void)hipGetLastError();
      va_ = nullptr;
    }
    if (fd_ >= 0) ::close(fd_);
    fd_ = -1;
  }

private:
  struct Alloc { // kfd_ioctl_alloc_memory_of_gpu_args
    uint64_t va_addr, size, handle, mmap_offset;
    uint32_t gpu_id, flags;
  };
  struct Free {
    uint64_t handle;
  };
  struct Map { // kfd_ioctl_map_memory_to_gpu_args / unmap
    uint64_t handle, device_ids_array_ptr;
    uint32_t n_devices, n_success;
  };
  bool mapped_ = false;
  static constexpr unsigned long kAlloc = _IOWR('K', 0x16, Alloc), kFree = _IOW('K', 0x17, Free), kMap = _IOWR('K', 0x18, Map),
                                 kUnmap = _IOWR('K', 0x19, Map);
  static constexpr size_t kVaBytes = 2u << 20;
  int call(unsigned long req, void *arg) {
    int r;
    do r = (int)syscall(SYS_ioctl, fd_, req, arg);
    while (r == -1 && (errno == EINTR || errno == EAGAIN));
    return r;
  }
  int fd_ = -1;
  uint32_t gpu_id_ = 0;
  void *va_ = nullptr;
  uint64_t handle_ = 0;
};

// ---- DrmVm: the process's GPU address space, driven through libdrm_amdgpu directly.
// What ROCr does per hsa_amd_vmem_map / set_access / unmap (tools/ioctl_timer.c, profiles/r01_unmap_trace_ioctls.log):
// export the handle from KFD as a dmabuf, import it into DRM, query it, mmap; export and import AGAIN, then the one
// ioctl that edits the page tables (DRM_AMDGPU_GEM_VA, 2.8 us); on unmap GEM_VA (2.4 us) + GEM_CLOSE. 8.1 us per
// map+unmap cycle, 5.2 of them GEM_VA. Keeping the imported buffer object for the lifetime of the handle leaves
// exactly those two ioctls (tools/drm_vmm_probe.cpp: map 2.2 us, unmap 2.1 us per 2 MiB page).
// Why this is the right address space: ROCr itself edits it through libdrm_amdgpu's amdgpu_bo_va_op on the
// amdgpu_device its thunk initialised for the render node whose VM KFD acquired; libdrm keeps ONE amdgpu_device per
// node and process, so amdgpu_device_initialize() on our own fd of that node returns the same device. That is a
// property of the library, not a contract: open() proves it before any memory is touched (a map over a VA that
// ROCr has mapped must be refused - mappings of one VM may not overlap) and the backend falls back to hybrid otherwise.
// libdrm_amdgpu is loaded with dlopen (ROCr links it, so the very same instance is already in the process); the
// five prototypes below are its stable public ABI (amdgpu.h), restated so that no -dev package is needed to build.
class DrmVm {
public:
  static DrmVm &instance() {
    static DrmVm *v = new DrmVm; // never destroyed: GPU contexts that outlive main() (no shutdown call) still release through it
    return *v;
  }
  bool ready() const { return dev_ != nullptr; }
  int hip_dev() const { return hip_dev_; }

  // Opens the render node of HIP device `hip_dev` (matched by PCI address). Returns false with a reason.
  bool open(int hip_dev, std::string *why) {
    std::lock_guard<std::mutex> g(mu_);
    if (dev_ && hip_dev_ == hip_dev) return true;
    if (dev_) close_locked();
    if (!load_api(why)) return false;
    char bdf[64] = {0};
    if (hipDeviceGetPCIBusId(bdf, sizeof bdf, hip_dev) != hipSuccess) {
      (void)hipGetLastError();
      *why = "hipDeviceGetPCIBusId failed";
      return false;
    }
    const std::string node = render_node_for(bdf);
    if (node.empty()) {
      *why = std::string("no DRM render node for ") + bdf;
      return false;
    }
    fd_ = ::open(node.c_str(), O_RDWR | O_CLOEXEC);
    if (fd_ < 0) {
      *why = "cannot open " + node + ": " + strerror(errno);
      return false;
    }
    uint32_t major = 0, minor = 0;
    const int r = api_.device_initialize(fd_, &major, &minor, &dev_);
    if (r != 0 || !dev_) {
      *why = "amdgpu_device_initialize(" + node + ") failed: " + strerror(r < 0 ? -r : r);
      ::close(fd_);
      fd_ = -1;
      dev_ = nullptr;
      return false;
    }
    hip_dev_ = hip_dev;
    return true;
  }
  void close() {
    std::lock_guard<std::mutex> g(mu_);
    close_locked();
  }

  // Import the ROCr handle's memory into DRM once; from then on find(h) answers. False: the handle stays ROCr-only.
  bool adopt(phys_handle_t h) {
    int dmabuf = -1;
    if (hsa_amd_vmem_export_shareable_handle(&dmabuf, hsa_amd_vmem_alloc_handle_t{h}, 0) != HSA_STATUS_SUCCESS) return false;
    ImportResult res{};
    std::lock_guard<std::mutex> g(mu_);
    const int r = dev_ ? api_.bo_import(dev_, kHandleTypeDmaBufFd, (uint32_t)dmabuf, &res) : -ENODEV;
    ::close(dmabuf);
    if (r != 0 || !res.bo) return false;
    bo_[h] = Entry{res.bo, false, 1};
    return true;
  }
  void *find(phys_handle_t h) {
    std::lock_guard<std::mutex> g(mu_);
    auto it = bo_.find(h);
    return it == bo_.end() ? nullptr : it->second.bo;
  }
  // The buffer object behind a handle as the allocator stores it, and which page of it: a handle known as it stands
  // (page 0 of its buffer: ROCr handles adopted, our own one-page buffers, imports), or a piece id (extent_pool.hpp) of
  // one of OUR multi-page buffers - accepted only if that buffer exists, was made by create() and has exactly the size
  // the id claims. Anything else is not a direct buffer (the caller takes ROCr's / HIP's path with the handle as it is):
  // handles of other origins use their high bits freely and are never masked.
  void *resolve(phys_handle_t h, unsigned *piece, unsigned *pages = nullptr) {
    std::lock_guard<std::mutex> g(mu_);
    auto it = bo_.find(h);
    if (it != bo_.end()) {
      *piece = 0;
      if (pages) *pages = 1;
      return it->second.bo;
    }
    if (!is_piece(h)) return nullptr;
    it = bo_.find(chunk_of(h));
    if (it == bo_.end() || !it->second.kfd || it->second.pages != pages_of(h) || piece_of(h) >= it->second.pages) return nullptr;
    *piece = piece_of(h);
    if (pages) *pages = it->second.pages;
    return it->second.bo;
  }
  // Drops DRM's reference; a buffer of our own making (create()) is given back to KFD as well. Returns whether the
  // handle was ours alone (nothing left for ROCr to release).
  bool forget(phys_handle_t h) {
    Entry e{};
    {
      std::lock_guard<std::mutex> g(mu_);
      auto it = bo_.find(h);
      if (it == bo_.end()) return false;
      e = it->second;
      bo_.erase(it);
    }
    const int64_t t0 = now_ns();
    (void)api_.bo_free(e.bo);
    if (!e.kfd) return false;
    KfdFree f{h};
    if (kfd_ioctl(kKfdFree, &f) != 0) KVC_LOG(LOG_ERROR, "AMDKFD_IOC_FREE_MEMORY_OF_GPU failed: %s", strerror(errno));
    create_times_.free_ns += now_ns() - t0;
    create_times_.frees++;
    return true;
  }

  // ---- physical memory straight from KFD. hipMemCreate and hsa_amd_vmem_handle_create cost O(live handles) in the
  // runtime's user space (9 -> 146 us at 32k handles) around an ioctl that is flat (1.5 us, tools/kfd_alloc_probe.cpp);
  // with mapping already off the runtime's hands nothing else needs its handle. KFD attaches every open of /dev/kfd
  // by one process to the same kfd_process, so a buffer allocated on our fd lives in the same context as ROCr's.
  bool kfd_ready() const { return kfd_fd_ >= 0 && gpu_id_ != 0; }
  void disable_kfd() {
    std::lock_guard<std::mutex> g(mu_);
    if (kfd_fd_ >= 0) ::close(kfd_fd_);
    kfd_fd_ = -1;
  }
  // Opens /dev/kfd and looks up KFD's id of the device (sysfs topology, matched by PCI address).
  bool open_kfd(std::string *why) {
    std::lock_guard<std::mutex> g(mu_);
    if (kfd_fd_ >= 0 && gpu_id_) return true;
    char bdf[64] = {0};
    unsigned dom = 0, bus = 0, dv = 0, fn = 0;
    if (hipDeviceGetPCIBusId(bdf, sizeof bdf, hip_dev_) != hipSuccess || sscanf(bdf, "%x:%x:%x.%x", &dom, &bus, &dv, &fn) != 4) {
      (void)hipGetLastError();
      *why = "no PCI address for the device";
      return false;
    }
    gpu_id_ = kfd_gpu_id_for(dom, bus, dv, fn);
    if (!gpu_id_) {
      *why = std::string("no KFD topology node for ") + bdf;
      return false;
    }
    kfd_fd_ = ::open("/dev/kfd", O_RDWR | O_CLOEXEC);
    if (kfd_fd_ < 0) {
      *why = std::string("cannot open /dev/kfd: ") + strerror(errno);
      return false;
    }
    return true;
  }
  // One buffer of `size` bytes of this GPU's memory, imported into DRM: {handle, bo}. Throws GpuError.
  // Where a creation's time goes (ns sums since the process started; kvc_get_option 112-115): the KFD allocation, the
  // dmabuf export, the import into DRM (+ close of the fd).
  struct CreateTimes {
    std::atomic<int64_t> alloc_ns{0}, export_ns{0}, import_ns{0}, count{0}, free_ns{0}, frees{0};
  };
  CreateTimes &create_times() { return create_times_; }
  phys_handle_t create(size_t size, unsigned pages = 1) {
    KfdAlloc a{};
    a.size = size;
    a.gpu_id = gpu_id_;
    a.flags = kKfdVramFlags;
    const int64_t t0 = now_ns();
    if (kfd_ioctl(kKfdAlloc, &a) != 0)
      throw GpuError(std::string("AMDKFD_IOC_ALLOC_MEMORY_OF_GPU failed: ") + (errno == ENOMEM ? "out of memory" : strerror(errno)));
    const int64_t t1 = now_ns();
    KfdExport e{};
    e.handle = a.handle;
    e.flags = O_CLOEXEC | O_RDWR;
    ImportResult res{};
    int r = kfd_ioctl(kKfdExport, &e) != 0 ? -errno : 0;
    const int64_t t2 = now_ns();
    if (r == 0) {
      std::lock_guard<std::mutex> g(mu_);
      r = dev_ ? api_.bo_import(dev_, kHandleTypeDmaBufFd, e.dmabuf_fd, &res) : -ENODEV;
      ::close((int)e.dmabuf_fd);
      if (r == 0 && res.bo) bo_[a.handle] = Entry{res.bo, true, (uint16_t)pages};
    }
    create_times_.alloc_ns += t1 - t0;
    create_times_.export_ns += t2 - t1;
    create_times_.import_ns += now_ns() - t2;
    create_times_.count++;
    if (r != 0 || !res.bo) {
      KfdFree f{a.handle};
      (void)kfd_ioctl(kKfdFree, &f);
      throw GpuError(std::string("exporting a KFD buffer into DRM failed: ") + strerror(r < 0 ? -r : EIO));
    }
    return a.handle;
  }
  // The peer's side of the cross-process pool: take a dmabuf fd of somebody else's buffer into OUR KFD process
  // (AMDKFD_IOC_IMPORT_DMABUF - the kernel then tracks it like our own when this process is evicted and restored) and
  // into DRM, like create(). The handle is released with forget() like one of ours. 4.3 us per buffer instead of the
  // 7 us of hsa_amd_vmem_import_shareable_handle + map + set_access. Throws GpuError.
  phys_handle_t import_fd(int dmabuf_fd) {
    KfdImport m{};
    m.gpu_id = gpu_id_;
    m.dmabuf_fd = (uint32_t)dmabuf_fd;
    if (kfd_ioctl(kKfdImport, &m) != 0) throw GpuError(std::string("AMDKFD_IOC_IMPORT_DMABUF failed: ") + strerror(errno));
    KfdExport e{};
    e.handle = m.handle;
    e.flags = O_CLOEXEC | O_RDWR;
    ImportResult res{};
    int r = kfd_ioctl(kKfdExport, &e) != 0 ? -errno : 0;
    if (r == 0) {
      std::lock_guard<std::mutex> g(mu_);
      r = dev_ ? api_.bo_import(dev_, kHandleTypeDmaBufFd, e.dmabuf_fd, &res) : -ENODEV;
      ::close((int)e.dmabuf_fd);
      if (r == 0 && res.bo) bo_[m.handle] = Entry{res.bo, true, 1};
    }
    if (r != 0 || !res.bo) {
      KfdFree f{m.handle};
      (void)kfd_ioctl(kKfdFree, &f);
      throw GpuError(std::string("taking an imported buffer into DRM failed: ") + strerror(r < 0 ? -r : EIO));
    }
    return m.handle;
  }
  // A dmabuf fd of a buffer made by create() (for the cross-process pool); -1 if `h` is not one.
  int export_fd(phys_handle_t h) {
    {
      std::lock_guard<std::mutex> g(mu_);
      auto it = bo_.find(h);
      if (it == bo_.end() || !it->second.kfd) return -1;
    }
    KfdExport e{};
    e.handle = h;
    e.flags = O_CLOEXEC | O_RDWR;
    if (kfd_ioctl(kKfdExport, &e) != 0) throw GpuError(std::string("AMDKFD_IOC_EXPORT_DMABUF failed: ") + strerror(errno));
    return (int)e.dmabuf_fd;
  }
  size_t adopted() {
    std::lock_guard<std::mutex> g(mu_);
    return bo_.size();
  }
  // 0 or a negative errno. The kernel serialises page-table edits per VM; no lock of ours is held across the ioctl.
  // `offset`: where in the buffer the mapping starts (GEM_VA honours it; HIP rejects one, ROCr ignores it).
  int map(void *bo, void *va, size_t size, uint64_t offset = 0) {
    return api_.bo_va_op(bo, offset, size, reinterpret_cast<uint64_t>(va), 0, kVaOpMap);
  }
  // UNMAP needs the exact extent of a mapping; CLEAR drops whatever is mapped in the range, splitting a larger mapping
  // if need be (a page that was mapped together with its neighbours in one ioctl) at the same cost - used when available.
  int unmap(void *bo, void *va, size_t size) {
    if (can_clear()) return clear(va, size);
    return api_.bo_va_op(bo, 0, size, reinterpret_cast<uint64_t>(va), 0, kVaOpUnmap);
  }
  // Drop EVERY mapping inside [va, va+size) with one ioctl (AMDGPU_VA_OP_CLEAR; the kernel walks its interval tree:
  // 1.5-1.7 us per mapping for runs of 8 and more against 2.1 for one UNMAP each, tools/drm_vmm_probe.cpp). The caller
  // guarantees that everything mapped in the range is its own and meant to go.
  // After pieces were CLEARed out of a larger mapping of `bo`: the kernel keeps what is left of that mapping, but the
  // page-table entries of the survivors still carry the FRAGMENT size of the original extent - the TLB may go on
  // translating the whole extent, hole included, from a neighbour's entry (seen as a slot that showed its previous
  // page after being backed afresh: benchmarks/soak_manager.py with chunked memory). The kernel rewrites the
  // remainders (with fragments that fit them) the next time it updates this buffer's mappings, i.e. on any MAP of it:
  // map one page of it at a scratch VA and drop that again. The caller invalidates the TLBs afterwards.
  bool refresh_mappings_of(void *bo, size_t page_bytes) {
    std::lock_guard<std::mutex> g(scratch_mu_);
    if (!scratch_va_ || scratch_bytes_ < page_bytes) {
      if (scratch_va_) (void)hipMemAddressFree(scratch_va_, scratch_bytes_);
      scratch_va_ = nullptr;
      if (hipMemAddressReserve(&scratch_va_, page_bytes, page_bytes, nullptr, 0) != hipSuccess) {
        (void)hipGetLastError();
        scratch_va_ = nullptr;
        return false;
      }
      scratch_bytes_ = page_bytes;
    }
    if (map(bo, scratch_va_, page_bytes, 0) != 0) return false;
    return clear(scratch_va_, page_bytes) == 0;
  }
  // "Drop whatever is mapped in [va, va+size) and map this range of `bo` there": alias -> page and page -> alias of the
  // compat mode's zero extent in ONE ioctl each way, for a whole run of slots (AMDGPU_VA_OP_REPLACE; mappings that
  // straddle the range are split by the kernel, which is what refresh_mappings_of() is for).
  int replace(void *bo, void *va, size_t size, uint64_t offset) {
    return api_.bo_va_op(bo, offset, size, reinterpret_cast<uint64_t>(va), 0, kVaOpReplace);
  }
  // A range with no buffer behind it and AMDGPU_VM_PAGE_PRT ("partially resident"): the PTEs are invalid with the PRT bit
  // set - reads return 0, writes are dropped, nothing faults (what Vulkan's sparse resources rest on; checked for
  // compute kernels, blit and copy-engine hipMemcpy and hipMemset on gfx950 by tools/prt_probe.cpp). The rest state of
  // unbacked KV slots of a compat region: "reads as zeros" without a zero page. NB the TLBs cache such an entry once
  // something has looked at the address (tools/prt_tlb_probe.cpp): backing the slot owes an invalidation like replacing
  // a valid mapping does. `replace`: drop or split whatever is mapped in the range first.
  int map_prt(void *va, size_t size, bool replace = false) {
    if (!can_clear()) return -ENOSYS;
    return api_.bo_va_op_raw(dev_, nullptr, 0, size, reinterpret_cast<uint64_t>(va), kVmPagePrt, replace ? kVaOpReplace : kVaOpMap);
  }
  // After slots were REPLACEd (or CLEARed) out of a larger PRT mapping: what is left of it keeps page-table entries that
  // carry the FRAGMENT size of the original mapping, and a TLB that caches one of them - PRT entries are cached once
  // looked at - goes on answering for the whole fragment, backed slots included: they read as zeros and swallow writes
  // (tools/prt_tlb_probe.cpp, "neighbours": 255 of 256 backed slots shadowed, for good, invalidations or not). The kernel
  // has the remainders queued to be written again at the next update of their owner, and all PRT mappings of one DRM file
  // share one owner: ONE PRT operation anywhere does it (same probe: 0 wrong reads afterwards). The caller invalidates
  // the TLBs afterwards. Same hazard and same cure as refresh_mappings_of() for buffers.
  bool refresh_prt_remainders() {
    std::lock_guard<std::mutex> g(scratch_mu_);
    constexpr size_t kBytes = 2u << 20;
    if (!prt_scratch_va_) {
      if (hipMemAddressReserve(&prt_scratch_va_, kBytes, kBytes, nullptr, 0) != hipSuccess) {
        (void)hipGetLastError();
        prt_scratch_va_ = nullptr;
        return false;
      }
    }
    return map_prt(prt_scratch_va_, kBytes, /*replace=*/true) == 0; // (replace: whatever an earlier call left there goes first)
  }
  bool can_clear() const { return api_.bo_va_op_raw != nullptr && dev_ != nullptr; }
  int clear(void *va, size_t size) { return api_.bo_va_op_raw(dev_, nullptr, 0, size, reinterpret_cast<uint64_t>(va), 0, kVaOpClear); }

private:
  struct ImportResult { // struct amdgpu_bo_import_result
    void *bo;
    uint64_t alloc_size;
  };
  struct Entry {
    void *bo;  // amdgpu_bo_handle
    bool kfd;  // allocated by create(): the key is KFD's handle, not ROCr's
    uint16_t pages = 1; // ... and how many pages it has (an extent: extent_pool.hpp)
  };
  // include/uapi/linux/kfd_ioctl.h, restated (the image's header predates EXPORT_DMABUF)
  struct KfdAlloc { // kfd_ioctl_alloc_memory_of_gpu_args
    uint64_t va_addr, size, handle, mmap_offset;
    uint32_t gpu_id, flags;
  };
  struct KfdFree { // kfd_ioctl_free_memory_of_gpu_args
    uint64_t handle;
  };
  struct KfdExport { // kfd_ioctl_export_dmabuf_args
    uint64_t handle;
    uint32_t flags, dmabuf_fd;
  };
  struct KfdImport { // kfd_ioctl_import_dmabuf_args
    uint64_t va_addr, handle;
    uint32_t gpu_id, dmabuf_fd;
  };
  static constexpr unsigned long kKfdAlloc = _IOWR('K', 0x16, KfdAlloc), kKfdFree = _IOW('K', 0x17, KfdFree),
                                 kKfdExport = _IOWR('K', 0x24, KfdExport), kKfdImport = _IOWR('K', 0x1D, KfdImport);
  // VRAM | WRITABLE | PUBLIC | NO_SUBSTITUTE, va 0: exactly what ROCr passes for hsa_amd_vmem_handle_create on the
  // coarse-grained device pool (its calls logged by tools/kfd_alloc_probe.cpp, profiles/r01_kfd_alloc_probe.log)
  static constexpr uint32_t kKfdVramFlags = (1u << 31) | (1u << 29) | (1u << 28) | 1u;
  int kfd_ioctl(unsigned long req, void *arg) {
    int r;
    do r = (int)syscall(SYS_ioctl, kfd_fd_, req, arg);
    while (r == -1 && (errno == EINTR || errno == EAGAIN)); // as the thunk does
    return r;
  }
  static constexpr int kHandleTypeDmaBufFd = 2;        // amdgpu_bo_handle_type_dma_buf_fd
  static constexpr uint32_t kVaOpMap = 1, kVaOpUnmap = 2, kVaOpClear = 3, kVaOpReplace = 4; // AMDGPU_VA_OP_MAP / _UNMAP / _CLEAR / _REPLACE
  static constexpr uint64_t kVmPagePrt = 1u << 4;                                              // AMDGPU_VM_PAGE_PRT
  struct Api {
    int (*device_initialize)(int, uint32_t *, uint32_t *, void **) = nullptr;
    int (*device_deinitialize)(void *) = nullptr;
    int (*bo_import)(void *, int, uint32_t, ImportResult *) = nullptr;
    int (*bo_free)(void *) = nullptr;
    int (*bo_va_op)(void *, uint64_t, uint64_t, uint64_t, uint64_t, uint32_t) = nullptr;
    int (*bo_va_op_raw)(void *, void *, uint64_t, uint64_t, uint64_t, uint64_t, uint32_t) = nullptr; // optional
  };

  bool load_api(std::string *why) {
    if (api_.bo_va_op) return true;
    void *lib = dlopen("libdrm_amdgpu.so.1", RTLD_NOW | RTLD_GLOBAL);
    if (!lib) {
      *why = std::string("dlopen(libdrm_amdgpu.so.1): ") + dlerror();
      return false;
    }
    Api a;
    a.device_initialize = reinterpret_cast<decltype(a.device_initialize)>(dlsym(lib, "amdgpu_device_initialize"));
    a.device_deinitialize = reinterpret_cast<decltype(a.device_deinitialize)>(dlsym(lib, "amdgpu_device_deinitialize"));
    a.bo_import = reinterpret_cast<decltype(a.bo_import)>(dlsym(lib, "amdgpu_bo_import"));
    a.bo_free = reinterpret_cast<decltype(a.bo_free)>(dlsym(lib, "amdgpu_bo_free"));
    a.bo_va_op = reinterpret_cast<decltype(a.bo_va_op)>(dlsym(lib, "amdgpu_bo_va_op"));
    a.bo_va_op_raw = reinterpret_cast<decltype(a.bo_va_op_raw)>(dlsym(lib, "amdgpu_bo_va_op_raw"));
    if (!a.device_initialize || !a.device_deinitialize || !a.bo_import || !a.bo_free || !a.bo_va_op) {
      *why = "libdrm_amdgpu.so.1 lacks an expected symbol";
      return false;
    }
    api_ = a;
    return true;
  }
  static std::string render_node_for(const char *bdf) {
    std::string found;
    bool ambiguous = false;
    DIR *d = opendir("/sys/class/drm");
    if (!d) return found;
    while (dirent *e = readdir(d)) {
      if (strncmp(e->d_name, "renderD", 7) != 0) continue;
      char link[PATH_MAX], real[PATH_MAX];
      snprintf(link, sizeof link, "/sys/class/drm/%s/device", e->d_name);
      if (!realpath(link, real)) continue;
      const char *leaf = strrchr(real, '/');
      if (leaf && strcasecmp(leaf + 1, bdf) == 0) {
        if (!found.empty()) ambiguous = true; // several render nodes behind one PCI address (a partitioned GPU)
        found = std::string("/dev/dri/") + e->d_name;
      }
    }
    closedir(d);
    if (ambiguous) found.clear(); // do not guess: the caller falls back to the hybrid backend
    return found;
  }
  void close_locked() {
    for (auto &kv : bo_) {
      (void)api_.bo_free(kv.second.bo);
      if (kv.second.kfd) {
        KfdFree f{kv.first};
        (void)kfd_ioctl(kKfdFree, &f);
      }
    }
    bo_.clear();
    if (kfd_fd_ >= 0) ::close(kfd_fd_);
    kfd_fd_ = -1;
    gpu_id_ = 0;
    if (dev_) (void)api_.device_deinitialize(dev_);
    dev_ = nullptr;
    if (fd_ >= 0) ::close(fd_);
    fd_ = -1;
    hip_dev_ = -1;
  }

  std::mutex mu_;
  Api api_;
  void *dev_ = nullptr; // amdgpu_device_handle
  int fd_ = -1, hip_dev_ = -1;
  int kfd_fd_ = -1;      // our own open of /dev/kfd (same kfd_process as ROCr's)
  uint32_t gpu_id_ = 0;  // KFD's id of the device
  std::unordered_map<phys_handle_t, Entry> bo_; // ROCr handle or KFD handle -> buffer object
  CreateTimes create_times_;
  std::mutex scratch_mu_;
  void *scratch_va_ = nullptr; // one page of reserved VA for refresh_mappings_of()
  size_t scratch_bytes_ = 0;
  void *prt_scratch_va_ = nullptr; // 2 MiB of reserved VA that refresh_prt_remainders() keeps PRT-mapped
};

} // namespace kvc
