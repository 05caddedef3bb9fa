// engine_ioctl_probe.cpp — what one page id costs the kernel in the geometry engines use on ROCm.
// One page id of the non-contiguous layout is 64 single 2 MiB slots in 64 places of the address space (32 layers x K/V):
// 64 GEM_VA ioctls whatever user space does. Questions, each answered per ioctl in microseconds:
//   A  MAP page r of ONE 64-page buffer at row r (lazy mode's map), CLEAR it again            - the floor
//   B  the same with 64 one-page buffers                                                        - does the buffer matter?
//   C  A with AMDGPU_VM_DELAY_UPDATE on the first 63 ioctls and a plain 64th                    - is the cost the syscall or the page-table write?
//   D  rows at rest in ONE PRT mapping of 64 slots each: REPLACE slot s by a page (splits the PRT mapping), then the one
//      PRT operation that makes the kernel rewrite the remainders (timed by itself), then REPLACE back to PRT
//   E  rows at rest in single-slot PRT mappings: REPLACE is exact, nothing is split
//   F  E with DELAY_UPDATE on the first 63                                                       - and does the data arrive?
//   G  runs of 8 adjacent slots per row, one ioctl each (pages r*8 .. r*8+7 of a 512-page buffer)
// C and F read the rows back through a kernel ONLY where the rest state is PRT (a page-table entry that was never written
// then reads 0 instead of faulting).
// build: hipcc --offload-arch=gfx950 -O2 -I/usr/include/libdrm -o engine_ioctl_probe engine_ioctl_probe.cpp -lhsa-runtime64 -ldrm_amdgpu -ldrm
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

static double now_us() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

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
static void tlb_shootdown() {
  void *p = nullptr;
  CK(hipMalloc(&p, 2u << 20));
  CK(hipFree(p));
}

// one word per row: out[r] = *(row r, slot s, word 0); stamp: writes v + r there
__global__ void touch_rows(char *base, size_t row_stride, size_t slot_off, unsigned *out, unsigned v, int do_write) {
  const int r = blockIdx.x;
  if (threadIdx.x == 0) {
    unsigned *p = (unsigned *)(base + (size_t)r * row_stride + slot_off);
    if (do_write) __hip_atomic_store(p, v + r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    out[r] = __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  }
}

static amdgpu_device_handle dev;
static const uint64_t RWX = AMDGPU_VM_PAGE_READABLE | AMDGPU_VM_PAGE_WRITEABLE | AMDGPU_VM_PAGE_EXECUTABLE;

struct Buf {
  hsa_amd_vmem_alloc_handle_t h;
  amdgpu_bo_handle bo;
};
static Buf make(size_t bytes) {
  Buf b{};
  HK(hsa_amd_vmem_handle_create(g_pool, bytes, MEMORY_TYPE_PINNED, 0, &b.h));
  int dfd = -1;
  HK(hsa_amd_vmem_export_shareable_handle(&dfd, b.h, 0));
  amdgpu_bo_import_result res{};
  DK(amdgpu_bo_import(dev, amdgpu_bo_handle_type_dma_buf_fd, (uint32_t)dfd, &res));
  close(dfd);
  b.bo = res.buf_handle;
  return b;
}

int main() {
  const int R = 64, S = 64; // rows (layer x K/V), slots per row
  const size_t PAGE = 2u << 20, ROW = 1ull << 30; // rows 1 GiB apart, as regions of an engine are
  CK(hipSetDevice(0));
  CK(hipFree(nullptr));
  HK(hsa_init());
  HK(hsa_iterate_agents(on_agent, nullptr));
  HK(hsa_amd_agent_iterate_memory_pools(g_gpu, on_pool, nullptr));
  char bdf[64] = {0};
  CK(hipDeviceGetPCIBusId(bdf, sizeof bdf, 0));
  const int fd = open(render_node_for(bdf).c_str(), O_RDWR | O_CLOEXEC);
  if (fd < 0) return perror("open render node"), 2;
  uint32_t maj = 0, min = 0;
  DK(amdgpu_device_initialize(fd, &maj, &min, &dev));
  void *va0 = nullptr;
  CK(hipMemAddressReserve(&va0, (size_t)R * ROW, PAGE, nullptr, 0));
  char *va = (char *)va0;
  auto at = [&](int r, int s) { return (uint64_t)(va + (size_t)r * ROW + (size_t)s * PAGE); };
  unsigned *out;
  CK(hipMalloc(&out, R * sizeof(unsigned)));
  std::vector<unsigned> host(R);
  hipStream_t st;
  CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));

  Buf big = make((size_t)R * PAGE);      // one page id = one buffer, page r behind row r
  std::vector<Buf> ones;
  for (int r = 0; r < R; r++) ones.push_back(make(PAGE));
  Buf wide = make((size_t)R * 8 * PAGE); // 8 page ids in one buffer, row-major: pages r*8 .. r*8+7 behind row r
  const int ROUNDS = 6;

  { // ---- A, B, C: unmapped rest state
    double a = 0, b = 0, c_delay = 0, c_last = 0, clr = 0;
    for (int round = 0; round < ROUNDS; round++) {
      const int s = 3 + round * 7;
      double t0 = now_us();
      for (int r = 0; r < R; r++) DK(amdgpu_bo_va_op_raw(dev, big.bo, (uint64_t)r * PAGE, PAGE, at(r, s), RWX, AMDGPU_VA_OP_MAP));
      double t1 = now_us();
      for (int r = 0; r < R; r++) DK(amdgpu_bo_va_op_raw(dev, nullptr, 0, PAGE, at(r, s), 0, AMDGPU_VA_OP_CLEAR));
      double t2 = now_us();
      for (int r = 0; r < R; r++) DK(amdgpu_bo_va_op_raw(dev, ones[r].bo, 0, PAGE, at(r, s), RWX, AMDGPU_VA_OP_MAP));
      double t3 = now_us();
      for (int r = 0; r < R; r++) DK(amdgpu_bo_va_op_raw(dev, nullptr, 0, PAGE, at(r, s), 0, AMDGPU_VA_OP_CLEAR));
      double t4 = now_us();
      for (int r = 0; r < R - 1; r++) DK(amdgpu_bo_va_op_raw(dev, big.bo, (uint64_t)r * PAGE, PAGE, at(r, s), RWX | AMDGPU_VM_DELAY_UPDATE, AMDGPU_VA_OP_MAP));
      double t5 = now_us();
      DK(amdgpu_bo_va_op_raw(dev, big.bo, (uint64_t)(R - 1) * PAGE, PAGE, at(R - 1, s), RWX, AMDGPU_VA_OP_MAP));
      double t6 = now_us();
      for (int r = 0; r < R; r++) DK(amdgpu_bo_va_op_raw(dev, nullptr, 0, PAGE, at(r, s), 0, AMDGPU_VA_OP_CLEAR));
      tlb_shootdown();
      if (round) a += t1 - t0, clr += t2 - t1, b += t3 - t2, c_delay += t5 - t4, c_last += t6 - t5;
    }
    const double n = (ROUNDS - 1);
    printf("A one 64-page buffer, MAP per row: %.2f us/ioctl | CLEAR per row: %.2f | B 64 one-page buffers, MAP: %.2f | "
           "C MAP with DELAY_UPDATE: %.2f us/ioctl for the first 63, the plain 64th: %.1f us (whole page id %.0f us against %.0f)\n",
           a / n / R, clr / n / R, b / n / R, c_delay / n / (R - 1), c_last / n, (c_delay + c_last) / n, a / n);
    fflush(stdout);
  }

  auto read_rows = [&](int s, unsigned v, int write) {
    touch_rows<<<R, 64, 0, st>>>(va, ROW, (size_t)s * PAGE, out, v, write);
    CK(hipMemcpyAsync(host.data(), out, R * sizeof(unsigned), hipMemcpyDeviceToHost, st));
    CK(hipStreamSynchronize(st));
  };

  { // ---- D: group PRT mappings (64 slots per row in one mapping), split by every map
    for (int r = 0; r < R; r++) DK(amdgpu_bo_va_op_raw(dev, nullptr, 0, (uint64_t)S * PAGE, at(r, 0), AMDGPU_VM_PAGE_PRT, AMDGPU_VA_OP_MAP));
    void *scratch = nullptr;
    CK(hipMemAddressReserve(&scratch, PAGE, PAGE, nullptr, 0));
    DK(amdgpu_bo_va_op_raw(dev, nullptr, 0, PAGE, (uint64_t)scratch, AMDGPU_VM_PAGE_PRT, AMDGPU_VA_OP_MAP));
    double rep = 0, rewrite = 0, back = 0, rep_again = 0, rewrite_again = 0;
    unsigned long long wrong = 0;
    for (int round = 0; round < ROUNDS; round++) {
      const int s = 3 + round * 7; // a fresh slot in the middle of what is left of the group: two remainders per row
      double t0 = now_us();
      for (int r = 0; r < R; r++) DK(amdgpu_bo_va_op_raw(dev, big.bo, (uint64_t)r * PAGE, PAGE, at(r, s), RWX, AMDGPU_VA_OP_REPLACE));
      double t1 = now_us();
      DK(amdgpu_bo_va_op_raw(dev, nullptr, 0, PAGE, (uint64_t)scratch, AMDGPU_VM_PAGE_PRT, AMDGPU_VA_OP_REPLACE));
      double t2 = now_us();
      tlb_shootdown();
      read_rows(s, 0x1000u * (round + 1), 1);
      for (int r = 0; r < R; r++) wrong += host[r] != 0x1000u * (round + 1) + r;
      double t3 = now_us();
      for (int r = 0; r < R; r++) DK(amdgpu_bo_va_op_raw(dev, nullptr, 0, PAGE, at(r, s), AMDGPU_VM_PAGE_PRT, AMDGPU_VA_OP_REPLACE));
      double t4 = now_us();
      tlb_shootdown();
      // the same slot again: its PRT mapping is a single slot now, nothing to split
      double t5 = now_us();
      for (int r = 0; r < R; r++) DK(amdgpu_bo_va_op_raw(dev, big.bo, (uint64_t)r * PAGE, PAGE, at(r, s), RWX, AMDGPU_VA_OP_REPLACE));
      double t6 = now_us();
      DK(amdgpu_bo_va_op_raw(dev, nullptr, 0, PAGE, (uint64_t)scratch, AMDGPU_VM_PAGE_PRT, AMDGPU_VA_OP_REPLACE));
      double t7 = now_us();
      tlb_shootdown();
      for (int r = 0; r < R; r++) DK(amdgpu_bo_va_op_raw(dev, nullptr, 0, PAGE, at(r, s), AMDGPU_VM_PAGE_PRT, AMDGPU_VA_OP_REPLACE));
      tlb_shootdown();
      if (round) rep += t1 - t0, rewrite += t2 - t1, back += t4 - t3, rep_again += t6 - t5, rewrite_again += t7 - t6;
    }
    const double n = ROUNDS - 1;
    printf("D rows at rest in one PRT mapping of %d slots: REPLACE by a page (splits it) %.2f us/ioctl, then the rewrite of the %d remainders: %.1f us in one ioctl "
           "(%.2f us per remainder); REPLACE back to PRT %.2f us/ioctl; the same slot again (exact, nothing to split): %.2f us/ioctl, rewrite ioctl %.1f us; wrong words %llu\n",
           S, rep / n / R, 2 * R, rewrite / n, rewrite / n / (2 * R), back / n / R, rep_again / n / R, rewrite_again / n, wrong);
    fflush(stdout);

    // ---- E, F: by now slots 3+7k are single-slot PRT mappings. E = "the same slot again" above. F: DELAY_UPDATE on the first 63
    double f_delay = 0, f_last = 0, fb_delay = 0, fb_last = 0;
    unsigned long long stale = 0, stale_back = 0;
    for (int round = 0; round < ROUNDS; round++) {
      const int s = 3 + round * 7;
      double t0 = now_us();
      for (int r = 0; r < R - 1; r++) DK(amdgpu_bo_va_op_raw(dev, big.bo, (uint64_t)r * PAGE, PAGE, at(r, s), RWX | AMDGPU_VM_DELAY_UPDATE, AMDGPU_VA_OP_REPLACE));
      double t1 = now_us();
      DK(amdgpu_bo_va_op_raw(dev, big.bo, (uint64_t)(R - 1) * PAGE, PAGE, at(R - 1, s), RWX, AMDGPU_VA_OP_REPLACE));
      double t2 = now_us();
      tlb_shootdown();
      read_rows(s, 0x100000u * (round + 1), 1); // a row whose entry was never written is still PRT: the write is dropped, the read is 0
      for (int r = 0; r < R; r++) stale += host[r] != 0x100000u * (round + 1) + r;
      double t3 = now_us();
      for (int r = 0; r < R - 1; r++) DK(amdgpu_bo_va_op_raw(dev, nullptr, 0, PAGE, at(r, s), AMDGPU_VM_PAGE_PRT | AMDGPU_VM_DELAY_UPDATE, AMDGPU_VA_OP_REPLACE));
      double t4 = now_us();
      DK(amdgpu_bo_va_op_raw(dev, nullptr, 0, PAGE, at(R - 1, s), AMDGPU_VM_PAGE_PRT, AMDGPU_VA_OP_REPLACE));
      double t5 = now_us();
      tlb_shootdown();
      read_rows(s, 0, 0); // back to PRT: every row reads 0 (a row still showing its page reads the stamp)
      for (int r = 0; r < R; r++) stale_back += host[r] != 0;
      if (round) f_delay += t1 - t0, f_last += t2 - t1, fb_delay += t4 - t3, fb_last += t5 - t4;
    }
    printf("F exact REPLACE with DELAY_UPDATE: %.2f us/ioctl for the first 63, the plain 64th %.1f us (page id %.0f us); rows NOT showing their page afterwards: %llu of %d | "
           "back to PRT with DELAY_UPDATE: %.2f us/ioctl, 64th %.1f us; rows still showing a page: %llu\n",
           f_delay / n / (R - 1), f_last / n, (f_delay + f_last) / n, stale, R * ROUNDS, fb_delay / n / (R - 1), fb_last / n, stale_back);
    fflush(stdout);

    // ---- G: runs of 8 slots per row, one ioctl per row; the slots are single-slot or group PRT mappings, whatever D left
    double g_rep = 0, g_rw = 0, g_back = 0;
    for (int round = 0; round < ROUNDS; round++) {
      const int s = 48 + (round % 2) * 8; // two places, each visited three times: the first visit splits, later ones are exact
      double t0 = now_us();
      for (int r = 0; r < R; r++) DK(amdgpu_bo_va_op_raw(dev, wide.bo, (uint64_t)r * 8 * PAGE, 8 * PAGE, at(r, s), RWX, AMDGPU_VA_OP_REPLACE));
      double t1 = now_us();
      DK(amdgpu_bo_va_op_raw(dev, nullptr, 0, PAGE, (uint64_t)scratch, AMDGPU_VM_PAGE_PRT, AMDGPU_VA_OP_REPLACE));
      double t2 = now_us();
      tlb_shootdown();
      double t2b = now_us();
      for (int r = 0; r < R; r++) DK(amdgpu_bo_va_op_raw(dev, nullptr, 0, 8 * PAGE, at(r, s), AMDGPU_VM_PAGE_PRT, AMDGPU_VA_OP_REPLACE));
      double t3 = now_us();
      tlb_shootdown();
      if (round >= 2) g_rep += t1 - t0, g_rw += t2 - t1, g_back += t3 - t2b;
    }
    printf("G runs of 8 slots per row (exact): REPLACE by 8 pages %.2f us/ioctl (%.2f us per 2 MiB), rewrite ioctl %.1f us, back to PRT %.2f us/ioctl\n",
           g_rep / (ROUNDS - 2) / R, g_rep / (ROUNDS - 2) / R / 8, g_rw / (ROUNDS - 2), g_back / (ROUNDS - 2) / R);
    fflush(stdout);
    // ---- H: is there a way from PRT to a page WITHOUT an instant in which the entry is invalid?
    //   REPLACE is clear-then-map inside one ioctl (amdgpu_vm_bo_replace_map: the old mapping goes to the freed list, whose
    //   entries are written as INVALID by amdgpu_vm_clear_freed before amdgpu_vm_bo_update writes the new ones); DELAY_UPDATE
    //   only postpones both passes to the next plain ioctl, where they still run in that order (F above: the data arrives, but
    //   between the two passes - now 64 ranges apart instead of one - the entries are invalid for longer, not shorter).
    //   The remaining candidate: MAP the page OVER the PRT mapping, so that the entry goes PRT -> valid in one write.
    {
      const int rc = amdgpu_bo_va_op_raw(dev, big.bo, 0, PAGE, at(0, 40), RWX, AMDGPU_VA_OP_MAP);
      printf("H MAP of a page over a PRT mapping (no REPLACE): %s - mappings of one address space may not overlap, so there is no sequence without an\n"
             "  invalid window; the window is the time between two page-table passes of ONE ioctl (~2 us per range), and nothing legitimate looks at\n"
             "  a slot that is being backed or given up\n", rc == 0 ? "ACCEPTED (unexpected)" : strerror(rc < 0 ? -rc : rc));
      if (rc == 0) DK(amdgpu_bo_va_op_raw(dev, nullptr, 0, PAGE, at(0, 40), AMDGPU_VM_PAGE_PRT, AMDGPU_VA_OP_REPLACE));
      fflush(stdout);
    }
    for (int r = 0; r < R; r++) DK(amdgpu_bo_va_op_raw(dev, nullptr, 0, (uint64_t)S * PAGE, at(r, 0), 0, AMDGPU_VA_OP_CLEAR));
    DK(amdgpu_bo_va_op_raw(dev, nullptr, 0, PAGE, (uint64_t)scratch, 0, AMDGPU_VA_OP_CLEAR));
    tlb_shootdown();
  }
  printf("done\n");
  return 0;
}
