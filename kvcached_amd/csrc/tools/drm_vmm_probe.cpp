// drm_vmm_probe.cpp — ROCr's hsa_amd_vmem_map + set_access + unmap cost 2.3 + 3.0 + 2.8 us per 2 MiB page, and the ioctl
// trace (tools/ioctl_timer.c) shows why: every map exports the handle as a dmabuf again, imports it into DRM again
// and mmap()s; every set_access exports and imports once more before the ONE ioctl that does the work
// (DRM_AMDGPU_GEM_VA, 2.8 us); every unmap is GEM_VA (2.4 us) + GEM_CLOSE. If the imported buffer object is kept for
// the handle's lifetime, a map is one GEM_VA ioctl and so is an unmap.
//
// ROCr reaches DRM through libdrm_amdgpu (amdgpu_bo_import / amdgpu_bo_va_op on the device handle its thunk
// initialised for the render node whose VM KFD acquired). libdrm (2.4.113 here) keeps ONE amdgpu_device per
// render node and process: amdgpu_device_initialize() on our own fd of the same node hands back that device, i.e. the
// process's compute VM. This probe
//   1. proves the VM is the same WITHOUT touching memory: a GEM_VA map over a VA that ROCr has mapped must be
//      refused (per-VM interval tree); if it is accepted we are in some other VM and stop;
//   2. times map / unmap through amdgpu_bo_va_op, 2 MiB pages, handles permuted over the VAs every round;
//   3. checks with kernels that data follows the HANDLE (each page carries its handle's number from the previous
//      round) — mappings are real and no stale translation survives the usual invalidation;
//   4. checks HIP's copy engines on a slot registered with HIP the way the hybrid backend does it.
// build: hipcc --offload-arch=gfx950 -O2 -I/usr/include/libdrm -o drm_vmm_probe drm_vmm_probe.cpp -lhsa-runtime64 -ldrm_amdgpu -ldrm
#include <hip/hip_runtime.h>
#include <hsa/hsa.h>
#include <hsa/hsa_ext_amd.h>

#include <amdgpu.h>
#include <amdgpu_drm.h>
#include <dirent.h>
#include <fcntl.h>
#include <limits.h>
#include <sys/ioctl.h>
#include <sys/syscall.h>
#include <unistd.h>

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
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
#define DK(x)                                                                                                          \
  do {                                                                                                                 \
    int r_ = (x);                                                                                                      \
    if (r_ != 0) {                                                                                                     \
      fprintf(stderr, "%s:%d %s -> %d (%s)\n", __FILE__, __LINE__, #x, r_, strerror(r_ < 0 ? -r_ : r_));               \
      exit(1);                                                                                                         \
    }                                                                                                                  \
  } while (0)

static double now_us() {
  return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

static hsa_agent_t g_gpu;
static hsa_amd_memory_pool_t g_pool;
static bool g_have_gpu = false, g_have_pool = false;
static hsa_status_t on_agent(hsa_agent_t a, void *) {
  hsa_device_type_t t;
  hsa_agent_get_info(a, HSA_AGENT_INFO_DEVICE, &t);
  if (t == HSA_DEVICE_TYPE_GPU && !g_have_gpu) {
    g_gpu = a;
    g_have_gpu = true;
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

// page i: every word = tag(i)
__global__ void stamp_pages(unsigned *base, size_t words_per_page, const unsigned *tags) {
  unsigned *p = base + (size_t)blockIdx.y * words_per_page;
  const unsigned v = tags[blockIdx.y];
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < words_per_page; i += (size_t)gridDim.x * blockDim.x) p[i] = v;
}
__global__ void check_pages(const unsigned *base, size_t words_per_page, const unsigned *tags, unsigned long long *bad) {
  const unsigned *p = base + (size_t)blockIdx.y * words_per_page;
  const unsigned v = tags[blockIdx.y];
  unsigned long long c = 0;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < words_per_page; i += (size_t)gridDim.x * blockDim.x) c += p[i] != v;
  if (c) atomicAdd(bad, c);
}

static std::string render_node_for(const char *bdf) {
  DIR *d = opendir("/sys/class/drm");
  if (!d) return "";
  std::string found;
  while (dirent *e = readdir(d)) {
    if (strncmp(e->d_name, "renderD", 7) != 0) continue;
    char link[PATH_MAX], real[PATH_MAX];
    snprintf(link, sizeof link, "/sys/class/drm/%s/device", e->d_name);
    if (!realpath(link, real)) continue;
    const char *leaf = strrchr(real, '/');
    if (leaf && strcasecmp(leaf + 1, bdf) == 0) found = std::string("/dev/dri/") + e->d_name;
  }
  closedir(d);
  return found;
}

// KFD's gpu_id of the device at PCI domain:bus:dev.fn - /sys/class/kfd/kfd/topology/nodes/<n>/{gpu_id,properties}
static uint32_t kfd_gpu_id_for(unsigned domain, unsigned bus, unsigned dev, unsigned fn) {
  const unsigned want_loc = (bus << 8) | (dev << 3) | fn;
  for (int n = 0; n < 64; n++) {
    char path[256];
    snprintf(path, sizeof path, "/sys/class/kfd/kfd/topology/nodes/%d/gpu_id", n);
    FILE *f = fopen(path, "r");
    if (!f) break;
    unsigned long id = 0;
    if (fscanf(f, "%lu", &id) != 1) id = 0;
    fclose(f);
    if (!id) continue; // a CPU node
    snprintf(path, sizeof path, "/sys/class/kfd/kfd/topology/nodes/%d/properties", n);
    f = fopen(path, "r");
    if (!f) continue;
    char key[64];
    unsigned long long val;
    unsigned long long loc = ~0ull, dom = 0;
    while (fscanf(f, "%63s %llu", key, &val) == 2) {
      if (!strcmp(key, "location_id")) loc = val;
      if (!strcmp(key, "domain")) dom = val;
    }
    fclose(f);
    if (loc == want_loc && dom == domain) return (uint32_t)id;
  }
  return 0;
}


// TLB invalidation. mode 0: what the product did first - hipMalloc + hipFree of 2 MiB (the KFD unmap inside hipFree
// flushes). mode 1: the same flush with nothing around it - a small buffer of our own (allocated once on our own fd
// of /dev/kfd, at a VA we reserved) is mapped to the GPU and unmapped again: AMDKFD_IOC_MAP_MEMORY_TO_GPU +
// AMDKFD_IOC_UNMAP_MEMORY_FROM_GPU, the second of which ends in the heavyweight flush. mode 2: none (control: the
// data check below must FAIL then, or it proves nothing).
struct kfd_alloc_args {
  uint64_t va_addr, size, handle, mmap_offset;
  uint32_t gpu_id, flags;
};
struct kfd_map_args {
  uint64_t handle, device_ids_array_ptr;
  uint32_t n_devices, n_success;
};
#define KFD_ALLOC _IOWR('K', 0x16, kfd_alloc_args)
#define KFD_MAP _IOWR('K', 0x18, kfd_map_args)
#define KFD_UNMAP _IOWR('K', 0x19, kfd_map_args)
static int g_mode = 0, g_kfd = -1;
static uint32_t g_gpu_id = 0;
static uint64_t g_flush_handle = 0;
static void tlb_flush_now() {
  void *p = nullptr;
  CK(hipMalloc(&p, 2u << 20));
  CK(hipFree(p));
}
static void tlb_shootdown() {
  if (g_mode == 2) return;
  if (g_mode == 0 || g_mode == 3) {
    tlb_flush_now();
    return;
  }
  kfd_map_args m{g_flush_handle, (uint64_t)(uintptr_t)&g_gpu_id, 1, 0};
  if (syscall(SYS_ioctl, g_kfd, KFD_MAP, &m) != 0) { perror("KFD_MAP"); exit(1); }
  m.n_success = 0;
  if (syscall(SYS_ioctl, g_kfd, KFD_UNMAP, &m) != 0) { perror("KFD_UNMAP"); exit(1); }
}

int main(int argc, char **argv) {
  const int n = argc > 1 ? atoi(argv[1]) : 1024;
  const int rounds = argc > 2 ? atoi(argv[2]) : 4;
  g_mode = argc > 3 ? atoi(argv[3]) : 0;
  const int clear_run = argc > 4 ? atoi(argv[4]) : 0; // > 0: unmap with AMDGPU_VA_OP_CLEAR over runs of this many slots
  const size_t PAGE = 2u << 20;
  CK(hipSetDevice(0));
  CK(hipFree(nullptr));
  HK(hsa_init());
  HK(hsa_iterate_agents(on_agent, nullptr));
  HK(hsa_amd_agent_iterate_memory_pools(g_gpu, on_pool, nullptr));
  if (!g_have_gpu || !g_have_pool) return fprintf(stderr, "no GPU agent / pool\n"), 1;

  char bdf[64] = {0};
  CK(hipDeviceGetPCIBusId(bdf, sizeof bdf, 0));
  const std::string node = render_node_for(bdf);
  printf("device 0 is %s -> %s\n", bdf, node.c_str());
  if (node.empty()) return fprintf(stderr, "no render node\n"), 2;
  { // which fds of this process already point at render nodes (the thunk's is among them)
    DIR *d = opendir("/proc/self/fd");
    while (dirent *e = readdir(d)) {
      char l[PATH_MAX], t[PATH_MAX];
      snprintf(l, sizeof l, "/proc/self/fd/%s", e->d_name);
      ssize_t k = readlink(l, t, sizeof t - 1);
      if (k > 0) {
        t[k] = 0;
        if (strstr(t, "/dev/dri/") || strstr(t, "/dev/kfd")) printf("  fd %s -> %s\n", e->d_name, t);
      }
    }
    closedir(d);
  }
  const int fd = open(node.c_str(), O_RDWR | O_CLOEXEC);
  if (fd < 0) return perror("open render node"), 2;
  uint32_t maj = 0, min = 0;
  amdgpu_device_handle dev = nullptr;
  DK(amdgpu_device_initialize(fd, &maj, &min, &dev));
  printf("libdrm_amdgpu %u.%u: our fd %d, device's fd %d\n", maj, min, fd, amdgpu_device_get_fd(dev));

  if (g_mode == 1) {
    unsigned dom = 0, bus = 0, dv = 0, fn = 0;
    sscanf(bdf, "%x:%x:%x.%x", &dom, &bus, &dv, &fn);
    g_gpu_id = kfd_gpu_id_for(dom, bus, dv, fn);
    g_kfd = open("/dev/kfd", O_RDWR | O_CLOEXEC);
    void *fva = nullptr;
    CK(hipMemAddressReserve(&fva, PAGE, PAGE, nullptr, 0)); // nothing else will ever use this VA
    kfd_alloc_args a{};
    a.va_addr = (uint64_t)fva;
    a.size = 4096;
    a.gpu_id = g_gpu_id;
    a.flags = (1u << 31) | (1u << 28) | 1u; // VRAM | WRITABLE | NO_SUBSTITUTE
    if (g_kfd < 0 || !g_gpu_id || syscall(SYS_ioctl, g_kfd, KFD_ALLOC, &a) != 0) return perror("flush buffer"), 6;
    g_flush_handle = a.handle;
    double t = now_us();
    for (int i = 0; i < 200; i++) tlb_shootdown();
    printf("KFD map+unmap of a 4 KiB buffer: %.1f us per pair\n", (now_us() - t) / 200);
  } else if (g_mode == 0) {
    double t = now_us();
    for (int i = 0; i < 200; i++) tlb_shootdown();
    printf("hipMalloc+hipFree of 2 MiB: %.1f us per pair\n", (now_us() - t) / 200);
  }
  void *va0 = nullptr;
  CK(hipMemAddressReserve(&va0, (size_t)n * PAGE, PAGE, nullptr, 0));
  char *va = (char *)va0;
  std::vector<hsa_amd_vmem_alloc_handle_t> h(n);
  std::vector<amdgpu_bo_handle> bo(n);
  double t0 = now_us();
  for (int i = 0; i < n; i++) HK(hsa_amd_vmem_handle_create(g_pool, PAGE, MEMORY_TYPE_PINNED, 0, &h[i]));
  double t1 = now_us();
  for (int i = 0; i < n; i++) {
    int dfd = -1;
    HK(hsa_amd_vmem_export_shareable_handle(&dfd, h[i], 0));
    amdgpu_bo_import_result res{};
    DK(amdgpu_bo_import(dev, amdgpu_bo_handle_type_dma_buf_fd, (uint32_t)dfd, &res));
    close(dfd);
    if (res.alloc_size < PAGE) return fprintf(stderr, "imported bo too small: %llu\n", (unsigned long long)res.alloc_size), 3;
    bo[i] = res.buf_handle;
  }
  double t2 = now_us();
  printf("create %.2f us/page, export+import (once per handle) %.2f us/page\n", (t1 - t0) / n, (t2 - t1) / n);

  // ---- 1. same VM? (no memory access involved)
  hsa_amd_memory_access_desc_t acc{HSA_ACCESS_PERMISSION_RW, g_gpu};
  HK(hsa_amd_vmem_map(va, PAGE, 0, h[0], 0));
  HK(hsa_amd_vmem_set_access(va, PAGE, &acc, 1));
  int r = amdgpu_bo_va_op(bo[1], 0, PAGE, (uint64_t)va, 0, AMDGPU_VA_OP_MAP);
  printf("GEM_VA map over a VA ROCr has mapped: %d (%s)\n", r, r ? "refused: same VM" : "ACCEPTED: this is not the compute VM");
  if (r == 0) {
    (void)amdgpu_bo_va_op(bo[1], 0, PAGE, (uint64_t)va, 0, AMDGPU_VA_OP_UNMAP);
    return 4;
  }
  HK(hsa_amd_vmem_unmap(va, PAGE));
  r = amdgpu_bo_va_op(bo[1], 0, PAGE, (uint64_t)va, 0, AMDGPU_VA_OP_MAP);
  printf("the same map after ROCr's unmap: %d\n", r);
  if (r != 0) return 5;
  DK(amdgpu_bo_va_op(bo[1], 0, PAGE, (uint64_t)va, 0, AMDGPU_VA_OP_UNMAP));
  tlb_shootdown();

  // ---- 2./3. timed rounds, data follows the handle
  unsigned *tags;
  unsigned long long *cnt;
  CK(hipMalloc(&tags, n * sizeof(unsigned)));
  CK(hipMalloc(&cnt, 8));
  hipStream_t s;
  CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  std::vector<unsigned> host_tags(n);
  if (g_mode == 3) {
    // Is the invalidation after a MAP needed at all? KFD itself does not flush when it maps on this GPU family (it
    // flushes after unmap only): a translation that was invalid cannot sit in a TLB on GFX9+. Adversarial version:
    // even slots stay mapped and are read by a kernel while the odd slots are unmapped (so the page-table lines that
    // hold the odd slots' invalid entries are fetched again and again), then the odd slots are re-backed with
    // permuted pages and checked IMMEDIATELY, with no invalidation since the one that followed their unmap.
    const int half = n / 2;
    auto odd_handle = [&](int j, int rd) { return 2 * (int)(((long)j * 37 + rd * 101) % half) + 1; }; // handle of odd slot 2j+1
    for (int i = 0; i < n; i++) DK(amdgpu_bo_va_op(bo[i], 0, PAGE, (uint64_t)(va + (size_t)i * PAGE), 0, AMDGPU_VA_OP_MAP));
    tlb_flush_now();
    for (int i = 0; i < n; i++) host_tags[i] = 0x70000000u | (unsigned)i; // page i carries its handle number
    CK(hipMemcpyAsync(tags, host_tags.data(), n * sizeof(unsigned), hipMemcpyHostToDevice, s));
    stamp_pages<<<dim3(8, n), 256, 0, s>>>((unsigned *)va, PAGE / 4, tags);
    CK(hipStreamSynchronize(s));
    std::vector<int> cur(half); // handle currently behind odd slot 2j+1
    for (int j = 0; j < half; j++) cur[j] = 2 * j + 1;
    for (int round = 1; round <= rounds; round++) {
      for (int j = 0; j < half; j++) DK(amdgpu_bo_va_op(bo[cur[j]], 0, PAGE, (uint64_t)(va + (size_t)(2 * j + 1) * PAGE), 0, AMDGPU_VA_OP_UNMAP));
      tlb_flush_now(); // the one invalidation of the cycle
      // neighbours at work while the odd slots are unbacked
      unsigned long long bad_even = 0, bad_odd = 0;
      CK(hipMemsetAsync(cnt, 0, 8, s));
      for (int rep = 0; rep < 3; rep++)
        for (int j = 0; j < half; j++) check_pages<<<dim3(8, 1), 256, 0, s>>>((const unsigned *)(va + (size_t)(2 * j) * PAGE), PAGE / 4, tags + 2 * j, cnt);
      CK(hipMemcpyAsync(&bad_even, cnt, 8, hipMemcpyDeviceToHost, s));
      CK(hipStreamSynchronize(s));
      double a = now_us();
      for (int j = 0; j < half; j++) {
        cur[j] = odd_handle(j, round);
        DK(amdgpu_bo_va_op(bo[cur[j]], 0, PAGE, (uint64_t)(va + (size_t)(2 * j + 1) * PAGE), 0, AMDGPU_VA_OP_MAP));
      }
      double b = now_us();
      // no invalidation here: every odd slot must show the page that was just put behind it
      std::vector<unsigned> want(half);
      for (int j = 0; j < half; j++) want[j] = 0x70000000u | (unsigned)cur[j];
      unsigned *wt;
      CK(hipMalloc(&wt, half * sizeof(unsigned)));
      CK(hipMemcpyAsync(wt, want.data(), half * sizeof(unsigned), hipMemcpyHostToDevice, s));
      CK(hipMemsetAsync(cnt, 0, 8, s));
      for (int j = 0; j < half; j++) check_pages<<<dim3(8, 1), 256, 0, s>>>((const unsigned *)(va + (size_t)(2 * j + 1) * PAGE), PAGE / 4, wt + j, cnt);
      CK(hipMemcpyAsync(&bad_odd, cnt, 8, hipMemcpyDeviceToHost, s));
      CK(hipStreamSynchronize(s));
      CK(hipFree(wt)); // (a small block: served from the runtime's cache, no driver trip)
      printf("round %d: odd slots re-backed (map %.2f us/page) and read with NO invalidation after the map: wrong words %llu (even neighbours: %llu)\n",
             round, (b - a) / half, bad_odd, bad_even);
      fflush(stdout);
    }
    for (int i = 0; i < n; i++) {
      const int hnd = (i & 1) ? cur[i / 2] : i;
      DK(amdgpu_bo_va_op(bo[hnd], 0, PAGE, (uint64_t)(va + (size_t)i * PAGE), 0, AMDGPU_VA_OP_UNMAP));
    }
    tlb_flush_now();
    for (int i = 0; i < n; i++) {
      DK(amdgpu_bo_free(bo[i]));
      HK(hsa_amd_vmem_handle_release(h[i]));
    }
    printf("done\n");
    return 0;
  }
  for (int round = 0; round < rounds; round++) {
    auto handle_at = [&](int i, int rd) { return (int)(((long)i * 37 + rd * 101) % n); }; // 37 coprime with 1024
    double a = now_us();
    for (int i = 0; i < n; i++) DK(amdgpu_bo_va_op(bo[handle_at(i, round)], 0, PAGE, (uint64_t)(va + (size_t)i * PAGE), 0, AMDGPU_VA_OP_MAP));
    double b = now_us();
    tlb_shootdown();
    double c = now_us();
    unsigned long long bad = 0;
    if (round > 0) { // every page must still hold what was written through its HANDLE last round
      for (int i = 0; i < n; i++) host_tags[i] = ((unsigned)(round - 1) << 20) | (unsigned)handle_at(i, round);
      CK(hipMemcpyAsync(tags, host_tags.data(), n * sizeof(unsigned), hipMemcpyHostToDevice, s));
      CK(hipMemsetAsync(cnt, 0, 8, s));
      check_pages<<<dim3(8, n), 256, 0, s>>>((const unsigned *)va, PAGE / 4, tags, cnt);
      CK(hipMemcpyAsync(&bad, cnt, 8, hipMemcpyDeviceToHost, s));
      CK(hipStreamSynchronize(s));
    }
    for (int i = 0; i < n; i++) host_tags[i] = ((unsigned)round << 20) | (unsigned)handle_at(i, round);
    CK(hipMemcpyAsync(tags, host_tags.data(), n * sizeof(unsigned), hipMemcpyHostToDevice, s));
    stamp_pages<<<dim3(8, n), 256, 0, s>>>((unsigned *)va, PAGE / 4, tags);
    CK(hipStreamSynchronize(s));
    double d = now_us();
    if (clear_run > 0) { // one ioctl drops every mapping in the range (the kernel walks its interval tree)
      for (int i = 0; i < n; i += clear_run)
        DK(amdgpu_bo_va_op_raw(dev, nullptr, 0, (uint64_t)std::min(clear_run, n - i) * PAGE, (uint64_t)(va + (size_t)i * PAGE), 0, AMDGPU_VA_OP_CLEAR));
    } else {
      for (int i = 0; i < n; i++) DK(amdgpu_bo_va_op(bo[handle_at(i, round)], 0, PAGE, (uint64_t)(va + (size_t)i * PAGE), 0, AMDGPU_VA_OP_UNMAP));
    }
    double e = now_us();
    tlb_shootdown();
    printf("round %d: map %.2f  unmap %.2f us/page (shootdown %.0f us);  words that did not follow their handle: %llu\n", round,
           (b - a) / n, (e - d) / n, c - b, bad);
    fflush(stdout);
  }

  // ---- 4. a slot registered with HIP (hybrid backend style), backed through DRM: HIP's copy paths
  hipMemGenericAllocationHandle_t shell{};
  hipMemAllocationProp prop{};
  prop.type = hipMemAllocationTypePinned;
  prop.location.type = hipMemLocationTypeDevice;
  prop.location.id = 0;
  CK(hipMemCreate(&shell, PAGE, &prop, 0));
  const int kReg = n < 32 ? n : 32;
  for (int i = 0; i < kReg; i++) {
    CK(hipMemMap(va + (size_t)i * PAGE, PAGE, 0, shell, 0));
    HK(hsa_amd_vmem_unmap(va + (size_t)i * PAGE, PAGE));
    DK(amdgpu_bo_va_op(bo[i], 0, PAGE, (uint64_t)(va + (size_t)i * PAGE), 0, AMDGPU_VA_OP_MAP));
  }
  tlb_shootdown();
  for (int i = 0; i < kReg; i++) host_tags[i] = 0xC0DE0000u + i;
  CK(hipMemcpyAsync(tags, host_tags.data(), kReg * sizeof(unsigned), hipMemcpyHostToDevice, s));
  stamp_pages<<<dim3(8, kReg), 256, 0, s>>>((unsigned *)va, PAGE / 4, tags);
  CK(hipStreamSynchronize(s));
  std::vector<unsigned> back((size_t)kReg * PAGE / 4);
  double ta = now_us();
  CK(hipMemcpy(back.data(), va, (size_t)kReg * PAGE, hipMemcpyDeviceToHost));
  double tb = now_us();
  size_t wrong = 0;
  for (int i = 0; i < kReg; i++)
    for (size_t w = 0; w < PAGE / 4; w += 4099) wrong += back[(size_t)i * PAGE / 4 + w] != 0xC0DE0000u + i;
  printf("hipMemcpy D2H of %d registered+DRM-mapped pages: %.1f GB/s, %zu sampled words wrong\n", kReg,
         (double)kReg * PAGE / ((tb - ta) * 1e3), wrong);
  for (auto &w : back) w = 0x5EED5EEDu;
  CK(hipMemcpy(va, back.data(), (size_t)kReg * PAGE, hipMemcpyHostToDevice));
  for (int i = 0; i < kReg; i++) host_tags[i] = 0x5EED5EEDu;
  CK(hipMemcpyAsync(tags, host_tags.data(), kReg * sizeof(unsigned), hipMemcpyHostToDevice, s));
  CK(hipMemsetAsync(cnt, 0, 8, s));
  check_pages<<<dim3(8, kReg), 256, 0, s>>>((const unsigned *)va, PAGE / 4, tags, cnt);
  unsigned long long bad = ~0ull;
  CK(hipMemcpyAsync(&bad, cnt, 8, hipMemcpyDeviceToHost, s));
  CK(hipStreamSynchronize(s));
  printf("hipMemcpy H2D into them, checked by a kernel: %llu words wrong\n", bad);
  CK(hipMemset(va, 0, (size_t)kReg * PAGE));
  CK(hipDeviceSynchronize()); // hipMemset on device memory is asynchronous, and `s` does not wait for the null stream
  for (int i = 0; i < kReg; i++) host_tags[i] = 0;
  CK(hipMemcpyAsync(tags, host_tags.data(), kReg * sizeof(unsigned), hipMemcpyHostToDevice, s));
  CK(hipMemsetAsync(cnt, 0, 8, s));
  check_pages<<<dim3(8, kReg), 256, 0, s>>>((const unsigned *)va, PAGE / 4, tags, cnt);
  CK(hipMemcpyAsync(&bad, cnt, 8, hipMemcpyDeviceToHost, s));
  CK(hipStreamSynchronize(s));
  printf("hipMemset over them, checked by a kernel: %llu words wrong\n", bad);

  // ---- teardown in the order the product would use: our mappings, then let HIP unmap a stand-in, then the handles
  for (int i = 0; i < kReg; i++) {
    DK(amdgpu_bo_va_op(bo[i], 0, PAGE, (uint64_t)(va + (size_t)i * PAGE), 0, AMDGPU_VA_OP_UNMAP));
    HK(hsa_amd_vmem_map(va + (size_t)i * PAGE, PAGE, 0, h[0], 0));
    CK(hipMemUnmap(va + (size_t)i * PAGE, PAGE));
  }
  CK(hipMemRelease(shell));
  double r0 = now_us();
  for (int i = 0; i < n; i++) {
    DK(amdgpu_bo_free(bo[i]));
    HK(hsa_amd_vmem_handle_release(h[i]));
  }
  printf("bo_free + handle_release: %.2f us/page\n", (now_us() - r0) / n);
  CK(hipMemAddressFree(va0, (size_t)n * PAGE));
  amdgpu_device_deinitialize(dev);
  close(fd);
  printf("done\n");
  return 0;
}
