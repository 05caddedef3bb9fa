// fill_bench: explore zero-fill kernel shapes on VMM-mapped 2 MiB pages (pure HIP, gfx950).
// hipcc --offload-arch=gfx950 -O3 -std=c++17 fill_bench.cpp -o fill_bench
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <numeric>
#include <random>
#include <vector>
#define CK(e) do { hipError_t s_ = (e); if (s_ != hipSuccess) { printf("FAIL %s -> %s\n", #e, hipGetErrorString(s_)); exit(1);} } while (0)
typedef unsigned int v4u __attribute__((ext_vector_type(4)));
static const size_t PAGE = 2u << 20;
template <int N> struct Table { void *p[N]; };

// one workgroup per SLAB bytes of one page; page pointer from the kernarg table
template <int THREADS, int SLAB, bool NT, int TBL, int XCD>
__global__ __launch_bounds__(THREADS) void fill_k(Table<TBL> t, unsigned slabs_per_page, unsigned total) {
  constexpr int STORES = SLAB / 16 / THREADS;
  unsigned b = blockIdx.x;
  unsigned page, slab;
  if (XCD == 1) { // blocks are dealt round-robin to the 8 XCDs: give XCD x the x-th contiguous eighth of the work
    unsigned per = total >> 3;
    b = (b & 7) * per + (b >> 3);
    page = b / slabs_per_page, slab = b - page * slabs_per_page;
  } else if (XCD == 2) { // page-granular: XCD x writes pages x, x+8, x+16, ... completely
    unsigned x = b & 7, i = b >> 3;
    unsigned q = i / slabs_per_page;
    slab = i - q * slabs_per_page;
    page = q * 8 + x;
  } else {
    page = b / slabs_per_page, slab = b - page * slabs_per_page;
  }
  v4u *dst = reinterpret_cast<v4u *>(static_cast<char *>(t.p[page]) + (size_t)slab * SLAB) + threadIdx.x;
  const v4u z = {0u, 0u, 0u, 0u};
#pragma unroll
  for (int i = 0; i < STORES; ++i) {
    if (NT) __builtin_nontemporal_store(z, dst + i * THREADS); else dst[i * THREADS] = z;
  }
}
// contiguous range, grid-stride persistent (reference point)
__global__ void fill_stride(v4u *p, size_t n16) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x, st = (size_t)gridDim.x * blockDim.x;
  const v4u z = {0u, 0u, 0u, 0u};
  for (; i < n16; i += st) p[i] = z;
}
// contiguous range, one WG per 64 KiB, no table (what the first probe measured at 6.9 TB/s)
__global__ __launch_bounds__(512) void fill_flat(v4u *p) {
  v4u *dst = p + (size_t)blockIdx.x * 4096 + threadIdx.x;
  const v4u z = {0u, 0u, 0u, 0u};
#pragma unroll
  for (int i = 0; i < 8; ++i) dst[i * 512] = z;
}

int main() {
  CK(hipSetDevice(0));
  const int NP = 1024;
  hipMemAllocationProp prop{}; prop.type = hipMemAllocationTypePinned; prop.location.type = hipMemLocationTypeDevice;
  hipMemAccessDesc acc{}; acc.location = prop.location; acc.flags = hipMemAccessFlagsProtReadWrite;
  char *va; CK(hipMemAddressReserve((void **)&va, NP * PAGE, PAGE, nullptr, 0));
  std::vector<hipMemGenericAllocationHandle_t> h(NP);
  for (int i = 0; i < NP; i++) { CK(hipMemCreate(&h[i], PAGE, &prop, 0)); CK(hipMemMap(va + i * PAGE, PAGE, 0, h[i], 0)); CK(hipMemSetAccess(va + i * PAGE, PAGE, &acc, 1)); }
  void *fl; CK(hipMalloc(&fl, PAGE)); CK(hipFree(fl)); // TLB shootdown
  hipStream_t s; CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  std::vector<int> order(NP); std::iota(order.begin(), order.end(), 0);
  std::vector<int> shuf = order; std::shuffle(shuf.begin(), shuf.end(), std::mt19937(0));

  auto time_it = [&](const char *name, size_t bytes_per_rep, auto fn) {
    for (int i = 0; i < 3; i++) fn();
    CK(hipStreamSynchronize(s));
    const int R = 20;
    CK(hipEventRecord(e0, s));
    for (int i = 0; i < R; i++) fn();
    CK(hipEventRecord(e1, s));
    CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    printf("%-44s %8.1f us/rep  %7.1f GB/s\n", name, ms * 1e3 / R, bytes_per_rep / (ms / R * 1e-3) / 1e9);
    fflush(stdout);
  };

  for (int npages : {256, 1024}) {
    size_t bytes = (size_t)npages * PAGE;
    printf("---- %d pages (%zu MiB) per rep\n", npages, bytes >> 20);
    time_it("flat contiguous 64KiB/WG (no table)", bytes, [&] { fill_flat<<<(unsigned)(bytes / 65536), 512, 0, s>>>((v4u *)va); });
    time_it("grid-stride 2048x512", bytes, [&] { fill_stride<<<2048, 512, 0, s>>>((v4u *)va, bytes / 16); });
    time_it("hipMemsetAsync", bytes, [&] { (void)hipMemsetAsync(va, 0, bytes, s); });
    for (int sh = 0; sh < 2; sh++) {
      std::vector<int> ord(npages); std::iota(ord.begin(), ord.end(), 0);
      if (sh) std::shuffle(ord.begin(), ord.end(), std::mt19937(0));
      auto run = [&](const char *nm, auto kern, int tbl, int threads, int slab) {
        char name[96]; snprintf(name, sizeof name, "%s %s", nm, sh ? "shuffled" : "seq");
        time_it(name, bytes, [&] {
          for (int off = 0; off < npages; off += tbl) {
            int k = std::min(tbl, npages - off);
            kern(off, k, threads, slab);
          }
        });
      };
#define KERN(THREADS, SLAB, NT, TBL, XCD)                                                           \
  [&](int off, int k, int, int) {                                                                   \
    Table<TBL> t{};                                                                                 \
    for (int i = 0; i < k; i++) t.p[i] = va + (size_t)ord[off + i] * PAGE;               \
    unsigned spp = PAGE / SLAB, total = spp * k;                                                    \
    fill_k<THREADS, SLAB, NT, TBL, XCD><<<total, THREADS, 0, s>>>(t, spp, total);                   \
  }
      run("tbl256  512thr  64K   x0   ", KERN(512, 65536, false, 256, 0), 256, 512, 65536);
      run("tbl1024 512thr  64K   x0   ", KERN(512, 65536, false, 1024, 0), 1024, 512, 65536);
      run("tbl1024 512thr  64K   x1   ", KERN(512, 65536, false, 1024, 1), 1024, 512, 65536);
      run("tbl1024 512thr  64K   x2   ", KERN(512, 65536, false, 1024, 2), 1024, 512, 65536);
      run("tbl256  512thr  64K   x2   ", KERN(512, 65536, false, 256, 2), 256, 512, 65536);
      run("tbl1024 512thr  128K  x2   ", KERN(512, 131072, false, 1024, 2), 1024, 512, 131072);
      run("tbl1024 1024thr 256K  x2   ", KERN(1024, 262144, false, 1024, 2), 1024, 1024, 262144);
      run("tbl1024 512thr  512K  x2   ", KERN(512, 524288, false, 1024, 2), 1024, 512, 524288);
      run("tbl1024 1024thr 2M    x2   ", KERN(1024, 2097152, false, 1024, 2), 1024, 1024, 2097152);
      run("tbl1024 512thr  2M    x0   ", KERN(512, 2097152, false, 1024, 0), 1024, 512, 2097152);
      run("tbl1024 256thr  64K   x2   ", KERN(256, 65536, false, 1024, 2), 1024, 256, 65536);
      run("tbl1024 1024thr 64K   x2   ", KERN(1024, 65536, false, 1024, 2), 1024, 1024, 65536);
    }
  }
  return 0;
}
