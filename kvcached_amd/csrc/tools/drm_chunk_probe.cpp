// drm_chunk_probe.cpp — DRM_AMDGPU_GEM_VA takes an offset into the buffer (HIP rejects a non-zero map offset, ROCr
// accepts and ignores it: tools/hsa_offset_probe.cpp). So physical memory could be allocated in CHUNKS of k x 2 MiB,
// any 2 MiB piece of a chunk mapped at any slot (offset = piece x 2 MiB), and - the point - a run of adjacent slots
// backed by adjacent pieces of one chunk mapped with ONE ioctl. What does that buy, and is the data right?
//   1. whole chunks at runs of k adjacent slots: us per 2 MiB page for k = 1, 4, 16, 64;
//   2. pieces scattered: every piece of every chunk at an arbitrary slot (one ioctl per piece, with offset);
//   3. after each step a kernel checks that every slot shows exactly the piece that was put behind it.
// build: hipcc --offload-arch=gfx950 -O2 -I/usr/include/libdrm -o drm_chunk_probe drm_chunk_probe.cpp -lhsa-runtime64 -ldrm_amdgpu -ldrm
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
  const int n = 1024; // slots
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
  unsigned *tags;
  unsigned long long *cnt;
  CK(hipMalloc(&tags, n * sizeof(unsigned)));
  CK(hipMalloc(&cnt, 8));
  hipStream_t s;
  CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  std::vector<unsigned> host_tags(n);

  for (int k : {1, 4, 16, 64}) {
    const int chunks = n / k;
    std::vector<hsa_amd_vmem_alloc_handle_t> h(chunks);
    std::vector<amdgpu_bo_handle> bo(chunks);
    double t0 = now_us();
    for (int c = 0; c < chunks; c++) {
      HK(hsa_amd_vmem_handle_create(g_pool, (size_t)k * PAGE, MEMORY_TYPE_PINNED, 0, &h[c]));
      int dfd = -1;
      HK(hsa_amd_vmem_export_shareable_handle(&dfd, h[c], 0));
      amdgpu_bo_import_result res{};
      DK(amdgpu_bo_import(dev, amdgpu_bo_handle_type_dma_buf_fd, (uint32_t)dfd, &res));
      close(dfd);
      bo[c] = res.buf_handle;
    }
    double t1 = now_us();
    // ---- 1. runs: chunk c (permuted) behind slots [j*k, (j+1)*k), one ioctl
    auto chunk_at = [&](int j, int rd) { return (int)(((long)j * 37 + rd * 11) % chunks); };
    double map_us = 0, unmap_us = 0;
    unsigned long long bad_total = 0;
    for (int round = 0; round < 3; round++) {
      double a = now_us();
      for (int j = 0; j < chunks; j++)
        DK(amdgpu_bo_va_op(bo[chunk_at(j, round)], 0, (uint64_t)k * PAGE, (uint64_t)(va + (size_t)j * k * PAGE), 0, AMDGPU_VA_OP_MAP));
      double b = now_us();
      tlb_shootdown();
      unsigned long long bad = 0;
      if (round > 0) { // slot i shows piece (i % k) of the chunk that sits there now, stamped last round
        for (int i = 0; i < n; i++) host_tags[i] = ((unsigned)(round - 1) << 24) | ((unsigned)chunk_at(i / k, round) << 8) | (unsigned)(i % k);
        CK(hipMemcpyAsync(tags, host_tags.data(), n * sizeof(unsigned), hipMemcpyHostToDevice, s));
        CK(hipMemsetAsync(cnt, 0, 8, s));
        check_pages<<<dim3(8, n), 256, 0, s>>>((const unsigned *)va, PAGE / 4, tags, cnt);
        CK(hipMemcpyAsync(&bad, cnt, 8, hipMemcpyDeviceToHost, s));
        CK(hipStreamSynchronize(s));
      }
      for (int i = 0; i < n; i++) host_tags[i] = ((unsigned)round << 24) | ((unsigned)chunk_at(i / k, round) << 8) | (unsigned)(i % k);
      CK(hipMemcpyAsync(tags, host_tags.data(), n * sizeof(unsigned), hipMemcpyHostToDevice, s));
      stamp_pages<<<dim3(8, n), 256, 0, s>>>((unsigned *)va, PAGE / 4, tags);
      CK(hipStreamSynchronize(s));
      double c = now_us();
      for (int j = 0; j < n; j += 16)
        DK(amdgpu_bo_va_op_raw(dev, nullptr, 0, (uint64_t)16 * PAGE, (uint64_t)(va + (size_t)j * PAGE), 0, AMDGPU_VA_OP_CLEAR));
      double d = now_us();
      tlb_shootdown();
      if (round > 0) {
        map_us += (b - a) / n / 2;
        unmap_us += (d - c) / n / 2;
      }
      bad_total += bad;
    }
    // ---- 2. pieces scattered: piece p of chunk c at slot perm(c*k+p), one ioctl each, with offset
    double a = now_us();
    for (int i = 0; i < n; i++) {
      const int slot = (int)(((long)i * 389 + 17) % n); // 389 coprime with 1024
      DK(amdgpu_bo_va_op(bo[i / k], (uint64_t)(i % k) * PAGE, PAGE, (uint64_t)(va + (size_t)slot * PAGE), 0, AMDGPU_VA_OP_MAP));
      host_tags[slot] = (2u << 24) | ((unsigned)(i / k) << 8) | (unsigned)(i % k); // what round 2 stamped into that piece
    }
    double b = now_us();
    tlb_shootdown();
    unsigned long long bad = 0;
    CK(hipMemcpyAsync(tags, host_tags.data(), n * sizeof(unsigned), hipMemcpyHostToDevice, s));
    CK(hipMemsetAsync(cnt, 0, 8, s));
    check_pages<<<dim3(8, n), 256, 0, s>>>((const unsigned *)va, PAGE / 4, tags, cnt);
    CK(hipMemcpyAsync(&bad, cnt, 8, hipMemcpyDeviceToHost, s));
    CK(hipStreamSynchronize(s));
    DK(amdgpu_bo_va_op_raw(dev, nullptr, 0, (uint64_t)n * PAGE, (uint64_t)va, 0, AMDGPU_VA_OP_CLEAR));
    tlb_shootdown();
    printf("chunks of %2d x 2 MiB: create+import %.2f us/page | runs of %2d adjacent slots per ioctl: map %.2f us/page, ranged unmap %.2f, wrong words %llu | "
           "pieces scattered (one ioctl each, offset): map %.2f us/page, wrong words %llu\n",
           k, (t1 - t0) / n, k, map_us, unmap_us, bad_total, (b - a) / n, bad);
    fflush(stdout);
    for (int c = 0; c < chunks; c++) {
      DK(amdgpu_bo_free(bo[c]));
      HK(hsa_amd_vmem_handle_release(h[c]));
    }
  }
  CK(hipMemAddressFree(va0, (size_t)n * PAGE));
  amdgpu_device_deinitialize(dev);
  close(fd);
  printf("done\n");
  return 0;
}
