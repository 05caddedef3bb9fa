// kfd_alloc_probe.cpp — creating a physical handle costs O(live handles) with HIP and with ROCr alike (9 -> 146 us at
// 32k handles, tools/create_scaling_probe.cpp) although the KFD ioctl underneath is flat (1.7-2.8 us): the growth path
// of the allocator is bound by user-space bookkeeping of the runtime. With the drm backend the runtime is no longer
// needed to MAP a page (tools/drm_vmm_probe.cpp) - is it needed to CREATE one?
//   1. what does ROCr pass to AMDKFD_IOC_ALLOC_MEMORY_OF_GPU for hsa_amd_vmem_handle_create? (ioctl() is interposed
//      by this executable; the arguments of ROCr's own calls are printed)
//   2. the same ioctl on our own fd of /dev/kfd (KFD attaches every open of a process to the same kfd_process), then
//      AMDKFD_IOC_EXPORT_DMABUF -> amdgpu_bo_import -> GEM_VA map: does a kernel see the memory, does data follow the
//      buffer, what do create / free cost at 32k live buffers?
// The KFD structs are restated from the kernel's uAPI header (include/uapi/linux/kfd_ioctl.h); the image's copy of the
// header predates EXPORT_DMABUF (0x24).
// build: hipcc --offload-arch=gfx950 -O2 -rdynamic -I/usr/include/libdrm -o kfd_alloc_probe kfd_alloc_probe.cpp -lhsa-runtime64 -ldrm_amdgpu -ldrm
#include <hip/hip_runtime.h>
#include <hsa/hsa.h>
#include <hsa/hsa_ext_amd.h>

#include <amdgpu.h>
#include <amdgpu_drm.h>
#include <dirent.h>
#include <fcntl.h>
#include <limits.h>
#include <stdarg.h>
#include <sys/ioctl.h>
#include <sys/syscall.h>
#include <unistd.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

struct kfd_alloc_args { // struct kfd_ioctl_alloc_memory_of_gpu_args
  uint64_t va_addr, size, handle, mmap_offset;
  uint32_t gpu_id, flags;
};
struct kfd_free_args {
  uint64_t handle;
};
struct kfd_export_args { // struct kfd_ioctl_export_dmabuf_args
  uint64_t handle;
  uint32_t flags, dmabuf_fd;
};
#define KFD_ALLOC _IOWR('K', 0x16, kfd_alloc_args)
#define KFD_FREE _IOW('K', 0x17, kfd_free_args)
#define KFD_EXPORT _IOWR('K', 0x24, kfd_export_args)

static bool g_log = false;
extern "C" int ioctl(int fd, unsigned long req, ...) {
  va_list ap;
  va_start(ap, req);
  void *arg = va_arg(ap, void *);
  va_end(ap);
  const bool alloc = g_log && _IOC_TYPE(req) == 'K' && _IOC_NR(req) == 0x16;
  kfd_alloc_args before{};
  if (alloc) before = *static_cast<kfd_alloc_args *>(arg);
  const long r = syscall(SYS_ioctl, fd, req, arg);
  if (alloc) {
    const auto *a = static_cast<kfd_alloc_args *>(arg);
    fprintf(stderr, "  [ROCr] ALLOC_MEMORY_OF_GPU fd %d: va 0x%llx size %llu gpu_id %u flags 0x%08x -> rc %ld handle 0x%llx mmap_offset 0x%llx\n",
            fd, (unsigned long long)before.va_addr, (unsigned long long)before.size, before.gpu_id, before.flags, r,
            (unsigned long long)a->handle, (unsigned long long)a->mmap_offset);
  } else if (g_log && _IOC_TYPE(req) == 'K') {
    fprintf(stderr, "  [ROCr] KFD ioctl nr 0x%02x size %u -> rc %ld\n", (unsigned)_IOC_NR(req), (unsigned)_IOC_SIZE(req), r);
  }
  return (int)r;
}

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
      fprintf(stderr, "%s:%d %s -> 0x%x\n", __FILE__, __LINE__, #x, (unsigned)s_);                                     \
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

static void tlb_shootdown() {
  void *p = nullptr;
  CK(hipMalloc(&p, 2u << 20));
  CK(hipFree(p));
}

int main(int argc, char **argv) {
  const int n_big = argc > 1 ? atoi(argv[1]) : 32768; // live buffers for the scaling part
  const int n = 1024;                                 // mapped and checked
  const size_t PAGE = 2u << 20;
  CK(hipSetDevice(0));
  CK(hipFree(nullptr));
  HK(hsa_init());
  HK(hsa_iterate_agents(on_agent, nullptr));
  HK(hsa_amd_agent_iterate_memory_pools(g_gpu, on_pool, nullptr));
  if (!g_have_gpu || !g_have_pool) return fprintf(stderr, "no GPU agent / pool\n"), 1;

  // ---- 1. what ROCr asks KFD for
  fprintf(stderr, "hsa_amd_vmem_handle_create(2 MiB) x2, then release:\n");
  hsa_amd_vmem_alloc_handle_t probe[2];
  g_log = true;
  HK(hsa_amd_vmem_handle_create(g_pool, PAGE, MEMORY_TYPE_PINNED, 0, &probe[0]));
  HK(hsa_amd_vmem_handle_create(g_pool, PAGE, MEMORY_TYPE_PINNED, 0, &probe[1]));
  HK(hsa_amd_vmem_handle_release(probe[0]));
  HK(hsa_amd_vmem_handle_release(probe[1]));
  g_log = false;

  char bdf[64] = {0};
  CK(hipDeviceGetPCIBusId(bdf, sizeof bdf, 0));
  unsigned dom = 0, bus = 0, dv = 0, fn = 0;
  sscanf(bdf, "%x:%x:%x.%x", &dom, &bus, &dv, &fn);
  const uint32_t gpu_id = kfd_gpu_id_for(dom, bus, dv, fn);
  printf("device 0 is %s; KFD gpu_id from sysfs: %u\n", bdf, gpu_id);
  if (!gpu_id) return 2;
  const uint32_t flags = argc > 2 ? (uint32_t)strtoul(argv[2], nullptr, 0) : 0xD0000001u; // VRAM | WRITABLE | EXECUTABLE | NO_SUBSTITUTE
  printf("our flags: 0x%08x (compare with ROCr's above)\n", flags);

  const int kfd = open("/dev/kfd", O_RDWR | O_CLOEXEC);
  if (kfd < 0) return perror("open /dev/kfd"), 2;
  const std::string node = render_node_for(bdf);
  const int rfd = open(node.c_str(), O_RDWR | O_CLOEXEC);
  if (rfd < 0) return perror("open render node"), 2;
  uint32_t maj, min;
  amdgpu_device_handle dev = nullptr;
  DK(amdgpu_device_initialize(rfd, &maj, &min, &dev));

  // ---- 2. our own buffers
  std::vector<uint64_t> h(n_big);
  std::vector<double> chunk_us;
  double t0 = now_us();
  for (int i = 0; i < n_big; i++) {
    if (i && i % 4096 == 0) {
      double t = now_us();
      chunk_us.push_back((t - t0) / 4096);
      t0 = t;
    }
    kfd_alloc_args a{};
    a.size = PAGE;
    a.gpu_id = gpu_id;
    a.flags = flags;
    if (syscall(SYS_ioctl, kfd, KFD_ALLOC, &a) != 0) {
      fprintf(stderr, "ALLOC #%d failed: %s\n", i, strerror(errno));
      return 3;
    }
    h[i] = a.handle;
  }
  chunk_us.push_back((now_us() - t0) / (n_big % 4096 ? n_big % 4096 : 4096));
  printf("KFD alloc, us per 2 MiB buffer by 4096-chunk (live buffers grow):");
  for (double c : chunk_us) printf(" %.2f", c);
  printf("\n");

  std::vector<amdgpu_bo_handle> bo(n);
  t0 = now_us();
  for (int i = 0; i < n; i++) {
    kfd_export_args e{};
    e.handle = h[i];
    e.flags = O_CLOEXEC | O_RDWR;
    if (syscall(SYS_ioctl, kfd, KFD_EXPORT, &e) != 0) {
      fprintf(stderr, "EXPORT_DMABUF failed: %s\n", strerror(errno));
      return 4;
    }
    amdgpu_bo_import_result res{};
    DK(amdgpu_bo_import(dev, amdgpu_bo_handle_type_dma_buf_fd, e.dmabuf_fd, &res));
    close((int)e.dmabuf_fd);
    bo[i] = res.buf_handle;
  }
  printf("export + import: %.2f us per buffer\n", (now_us() - t0) / n);

  void *va0 = nullptr;
  CK(hipMemAddressReserve(&va0, (size_t)n * PAGE, PAGE, nullptr, 0));
  char *va = (char *)va0;
  unsigned *tags;
  unsigned long long *cnt;
  CK(hipMalloc(&tags, n * sizeof(unsigned)));
  CK(hipMalloc(&cnt, 8));
  hipStream_t s;
  CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  std::vector<unsigned> host_tags(n);
  for (int round = 0; round < 3; round++) {
    auto buf_at = [&](int i, int rd) { return (int)(((long)i * 37 + rd * 101) % n); };
    double a = now_us();
    for (int i = 0; i < n; i++) DK(amdgpu_bo_va_op(bo[buf_at(i, round)], 0, PAGE, (uint64_t)(va + (size_t)i * PAGE), 0, AMDGPU_VA_OP_MAP));
    double b = now_us();
    tlb_shootdown();
    unsigned long long bad = 0;
    if (round > 0) {
      for (int i = 0; i < n; i++) host_tags[i] = ((unsigned)(round - 1) << 20) | (unsigned)buf_at(i, round);
      CK(hipMemcpyAsync(tags, host_tags.data(), n * sizeof(unsigned), hipMemcpyHostToDevice, s));
      CK(hipMemsetAsync(cnt, 0, 8, s));
      check_pages<<<dim3(8, n), 256, 0, s>>>((const unsigned *)va, PAGE / 4, tags, cnt);
      CK(hipMemcpyAsync(&bad, cnt, 8, hipMemcpyDeviceToHost, s));
      CK(hipStreamSynchronize(s));
    }
    for (int i = 0; i < n; i++) host_tags[i] = ((unsigned)round << 20) | (unsigned)buf_at(i, round);
    CK(hipMemcpyAsync(tags, host_tags.data(), n * sizeof(unsigned), hipMemcpyHostToDevice, s));
    stamp_pages<<<dim3(8, n), 256, 0, s>>>((unsigned *)va, PAGE / 4, tags);
    CK(hipStreamSynchronize(s));
    double d = now_us();
    for (int i = 0; i < n; i++) DK(amdgpu_bo_va_op(bo[buf_at(i, round)], 0, PAGE, (uint64_t)(va + (size_t)i * PAGE), 0, AMDGPU_VA_OP_UNMAP));
    double e = now_us();
    tlb_shootdown();
    printf("round %d: map %.2f  unmap %.2f us/page;  words that did not follow their buffer: %llu\n", round, (b - a) / n, (e - d) / n, bad);
  }
  size_t free_b = 0, total_b = 0;
  CK(hipMemGetInfo(&free_b, &total_b));
  printf("hipMemGetInfo with %d buffers live: %.1f GiB used\n", n_big, (total_b - free_b) / 1073741824.0);

  // ---- free: the touched ones (first n) and the untouched rest, oldest first
  for (int i = 0; i < n; i++) DK(amdgpu_bo_free(bo[i]));
  t0 = now_us();
  for (int i = 0; i < n; i++) {
    kfd_free_args f{h[i]};
    if (syscall(SYS_ioctl, kfd, KFD_FREE, &f) != 0) return fprintf(stderr, "FREE failed: %s\n", strerror(errno)), 5;
  }
  double t1 = now_us();
  for (int i = n; i < n_big; i++) {
    kfd_free_args f{h[i]};
    if (syscall(SYS_ioctl, kfd, KFD_FREE, &f) != 0) return fprintf(stderr, "FREE failed: %s\n", strerror(errno)), 5;
  }
  double t2 = now_us();
  printf("KFD free: %.2f us per touched buffer, %.2f us per untouched buffer\n", (t1 - t0) / n, (t2 - t1) / (n_big - n > 0 ? n_big - n : 1));
  CK(hipMemGetInfo(&free_b, &total_b));
  printf("hipMemGetInfo after freeing: %.1f GiB used\n", (total_b - free_b) / 1073741824.0);
  CK(hipMemAddressFree(va0, (size_t)n * PAGE));
  amdgpu_device_deinitialize(dev);
  close(rfd);
  close(kfd);
  printf("done\n");
  return 0;
}
