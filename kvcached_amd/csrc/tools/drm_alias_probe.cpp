// drm_alias_probe.cpp — the reference aliases every unbacked slot to ONE physical page of zeros (ftensor.cpp:160-176).
// Through HIP/ROCr that is expensive on ROCm: unmapping an alias walks the handle's mapping list in user space
// (O(aliases), DESIGN.md 4.2 - hence sharded zero pages), and swapping alias <-> page is two or three calls.
// DRM_AMDGPU_GEM_VA has two more operations: REPLACE (drop whatever is mapped in the range, map this buffer: one
// ioctl for "alias out, page in" and one for "page out, alias in") and CLEAR (drop every mapping in a range).
// Questions: what does an alias cost when ONE buffer carries 32k of them (map, REPLACE away, REPLACE back)? Does the
// cost grow with the number of aliases? Is the data right (pages keep their contents, aliases read zeros)?
// What does CLEAR of the whole range cost?
// build: hipcc --offload-arch=gfx950 -O2 -I/usr/include/libdrm -o drm_alias_probe drm_alias_probe.cpp -lhsa-runtime64 -ldrm_amdgpu -ldrm
#include <hip/hip_runtime.h>
#include <hsa/hsa.h>
#include <hsa/hsa_ext_amd.h>

#include <amdgpu.h>
#include <amdgpu_drm.h>
#include <dirent.h>
#include <fcntl.h>
#include <limits.h>
#include <unistd.h>

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

static void tlb_shootdown() { // what the product does after a batch of (un)maps: a KFD free that the driver flushes for
  void *p = nullptr;
  CK(hipMalloc(&p, 2u << 20));
  CK(hipFree(p));
}

int main(int argc, char **argv) {
  const int n = argc > 1 ? atoi(argv[1]) : 32768; // slots (aliases of the one zero page)
  const int k = 1024;                             // pages swapped in and out per round
  const size_t PAGE = 2u << 20;
  CK(hipSetDevice(0));
  CK(hipFree(nullptr));
  HK(hsa_init());
  HK(hsa_iterate_agents(on_agent, nullptr));
  HK(hsa_amd_agent_iterate_memory_pools(g_gpu, on_pool, nullptr));
  char bdf[64] = {0};
  CK(hipDeviceGetPCIBusId(bdf, sizeof bdf, 0));
  const std::string node = render_node_for(bdf);
  const int fd = open(node.c_str(), O_RDWR | O_CLOEXEC);
  if (fd < 0) return perror("open render node"), 2;
  uint32_t maj = 0, min = 0;
  amdgpu_device_handle dev = nullptr;
  DK(amdgpu_device_initialize(fd, &maj, &min, &dev));

  void *va0 = nullptr;
  CK(hipMemAddressReserve(&va0, (size_t)n * PAGE, PAGE, nullptr, 0));
  char *va = (char *)va0;
  std::vector<hsa_amd_vmem_alloc_handle_t> h(k + 1);
  std::vector<amdgpu_bo_handle> bo(k + 1); // [k] = the zero page
  for (int i = 0; i <= k; i++) {
    HK(hsa_amd_vmem_handle_create(g_pool, PAGE, MEMORY_TYPE_PINNED, 0, &h[i]));
    int dfd = -1;
    HK(hsa_amd_vmem_export_shareable_handle(&dfd, h[i], 0));
    amdgpu_bo_import_result res{};
    DK(amdgpu_bo_import(dev, amdgpu_bo_handle_type_dma_buf_fd, (uint32_t)dfd, &res));
    close(dfd);
    bo[i] = res.buf_handle;
  }
  amdgpu_bo_handle zero = bo[k];

  // ---- all slots alias the zero page
  printf("alias map, us per slot by 4096-chunk (aliases of the one buffer grow):");
  double t0 = now_us();
  for (int i = 0; i < n; i++) {
    if (i && i % 4096 == 0) {
      double t = now_us();
      printf(" %.2f", (t - t0) / 4096);
      t0 = t;
    }
    DK(amdgpu_bo_va_op(zero, 0, PAGE, (uint64_t)(va + (size_t)i * PAGE), 0, AMDGPU_VA_OP_MAP));
  }
  printf(" %.2f\n", (now_us() - t0) / (n % 4096 ? n % 4096 : 4096));
  tlb_shootdown();
  unsigned *tags;
  unsigned long long *cnt;
  CK(hipMalloc(&tags, k * sizeof(unsigned)));
  CK(hipMalloc(&cnt, 8));
  hipStream_t s;
  CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  CK(hipMemsetAsync(tags, 0, 4, s));
  stamp_pages<<<dim3(8, 1), 256, 0, s>>>((unsigned *)va, PAGE / 4, tags); // the zero page, through its first alias (HIP has not
  CK(hipStreamSynchronize(s));                                            // been told about this VA: kernels only)
  std::vector<unsigned> host_tags(k);
  std::vector<int> slot(k);
  for (int round = 0; round < 3; round++) {
    for (int j = 0; j < k; j++) slot[j] = (int)(((long)j * 7919 + round * 104729 + 13) % n); // scattered, distinct (7919 coprime with n)
    double a = now_us();
    for (int j = 0; j < k; j++) DK(amdgpu_bo_va_op(bo[j], 0, PAGE, (uint64_t)(va + (size_t)slot[j] * PAGE), 0, AMDGPU_VA_OP_REPLACE));
    double b = now_us();
    tlb_shootdown();
    // pages carry their number; written through the slot VA, checked through it, while the neighbours must read zero
    unsigned long long bad = 0, bad_zero = 0;
    for (int j = 0; j < k; j++) host_tags[j] = 0xA0000000u | ((unsigned)round << 16) | (unsigned)j;
    CK(hipMemcpyAsync(tags, host_tags.data(), k * sizeof(unsigned), hipMemcpyHostToDevice, s));
    for (int j = 0; j < k; j++) stamp_pages<<<dim3(8, 1), 256, 0, s>>>((unsigned *)(va + (size_t)slot[j] * PAGE), PAGE / 4, tags + j);
    CK(hipMemsetAsync(cnt, 0, 8, s));
    for (int j = 0; j < k; j++) check_pages<<<dim3(8, 1), 256, 0, s>>>((const unsigned *)(va + (size_t)slot[j] * PAGE), PAGE / 4, tags + j, cnt);
    CK(hipMemcpyAsync(&bad, cnt, 8, hipMemcpyDeviceToHost, s));
    CK(hipStreamSynchronize(s));
    double c = now_us();
    for (int j = 0; j < k; j++) DK(amdgpu_bo_va_op(zero, 0, PAGE, (uint64_t)(va + (size_t)slot[j] * PAGE), 0, AMDGPU_VA_OP_REPLACE));
    double d = now_us();
    tlb_shootdown();
    // every one of those slots reads zeros again (the zero page was never written through a stale translation)
    unsigned zt = 0;
    CK(hipMemcpyAsync(tags, &zt, 4, hipMemcpyHostToDevice, s));
    CK(hipMemsetAsync(cnt, 0, 8, s));
    for (int j = 0; j < k; j++) check_pages<<<dim3(8, 1), 256, 0, s>>>((const unsigned *)(va + (size_t)slot[j] * PAGE), PAGE / 4, tags, cnt);
    CK(hipMemcpyAsync(&bad_zero, cnt, 8, hipMemcpyDeviceToHost, s));
    CK(hipStreamSynchronize(s));
    printf("round %d: REPLACE alias->page %.2f  page->alias %.2f us/slot;  wrong words in pages %llu, non-zero words behind aliases %llu\n",
           round, (b - a) / k, (d - c) / k, bad, bad_zero);
    fflush(stdout);
  }
  // plain UNMAP of an alias (the kernel finds the mapping in the buffer's list): cost with ~32k aliases
  {
    double a = now_us();
    for (int j = 0; j < 256; j++) DK(amdgpu_bo_va_op(zero, 0, PAGE, (uint64_t)(va + (size_t)(n - 1 - j) * PAGE), 0, AMDGPU_VA_OP_UNMAP));
    double b = now_us();
    for (int j = 0; j < 256; j++) DK(amdgpu_bo_va_op(zero, 0, PAGE, (uint64_t)(va + (size_t)j * PAGE), 0, AMDGPU_VA_OP_UNMAP));
    double c = now_us();
    printf("plain UNMAP of an alias: newest 256 aliases %.2f us each, oldest 256 aliases %.2f us each\n", (b - a) / 256, (c - b) / 256);
  }
  double a = now_us();
  DK(amdgpu_bo_va_op_raw(dev, nullptr, 0, (uint64_t)n * PAGE, (uint64_t)va, 0, AMDGPU_VA_OP_CLEAR));
  printf("CLEAR of the whole %d-slot range: %.1f us in one ioctl\n", n, now_us() - a);
  tlb_shootdown();
  // the range really is empty: a map at any slot is accepted again
  DK(amdgpu_bo_va_op(bo[0], 0, PAGE, (uint64_t)(va + (size_t)(n / 2) * PAGE), 0, AMDGPU_VA_OP_MAP));
  DK(amdgpu_bo_va_op(bo[0], 0, PAGE, (uint64_t)(va + (size_t)(n / 2) * PAGE), 0, AMDGPU_VA_OP_UNMAP));
  for (int i = 0; i <= k; i++) {
    DK(amdgpu_bo_free(bo[i]));
    HK(hsa_amd_vmem_handle_release(h[i]));
  }
  CK(hipMemAddressFree(va0, (size_t)n * PAGE));
  amdgpu_device_deinitialize(dev);
  close(fd);
  printf("done\n");
  return 0;
}
