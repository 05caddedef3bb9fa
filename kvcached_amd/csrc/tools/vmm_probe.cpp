// vmm_probe: characterise the ROCm VMM driver calls on one MI355X before the
// allocator design is fixed. Every step tolerates failure (prints and goes on);
// nothing here touches memory that is not mapped.
//
// Build: hipcc --offload-arch=gfx950 -O2 -std=c++17 -pthread vmm_probe.cpp -o vmm_probe
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <numeric>
#include <thread>
#include <unistd.h>
#include <vector>

using clk = std::chrono::steady_clock;
static double us_since(clk::time_point t0) {
  return std::chrono::duration<double, std::micro>(clk::now() - t0).count();
}

#define TRY(expr)                                                              \
  ([&]() {                                                                     \
    hipError_t e_ = (expr);                                                    \
    if (e_ != hipSuccess)                                                      \
      printf("  !! %s -> %d (%s)\n", #expr, (int)e_, hipGetErrorString(e_));   \
    return e_;                                                                 \
  }())

struct Stat {
  std::vector<double> v;
  void add(double x) { v.push_back(x); }
  void print(const char *name) {
    if (v.empty()) {
      printf("%-28s (no samples)\n", name);
      return;
    }
    std::sort(v.begin(), v.end());
    double sum = std::accumulate(v.begin(), v.end(), 0.0);
    auto q = [&](double p) { return v[std::min(v.size() - 1, (size_t)(p * v.size()))]; };
    printf("%-28s n=%zu avg=%.2f p50=%.2f p90=%.2f p99=%.2f max=%.2f us\n", name,
           v.size(), sum / v.size(), q(0.5), q(0.9), q(0.99), v.back());
  }
};

static const size_t MiB = 1ull << 20;
static const size_t GiB = 1ull << 30;
static const size_t PAGE = 2 * MiB;

__global__ void fill_plain(uint4 *p, size_t n16) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  size_t stride = (size_t)gridDim.x * blockDim.x;
  uint4 z = {0, 0, 0, 0};
  for (; i < n16; i += stride) p[i] = z;
}
__global__ void fill_nt(uint4 *p, size_t n16) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  size_t stride = (size_t)gridDim.x * blockDim.x;
  typedef unsigned int v4 __attribute__((ext_vector_type(4)));
  v4 z = {0, 0, 0, 0};
  for (; i < n16; i += stride) __builtin_nontemporal_store(z, (v4 *)p + i);
}
__global__ void poison(uint4 *p, size_t n16) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  size_t stride = (size_t)gridDim.x * blockDim.x;
  uint4 z = {0xA5A5A5A5u, 0xA5A5A5A5u, 0xA5A5A5A5u, 0xA5A5A5A5u};
  for (; i < n16; i += stride) p[i] = z;
}
__global__ void count_nonzero(const uint4 *p, size_t n16, unsigned long long *out) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  size_t stride = (size_t)gridDim.x * blockDim.x;
  unsigned long long c = 0;
  for (; i < n16; i += stride) {
    uint4 v = p[i];
    c += (v.x != 0) + (v.y != 0) + (v.z != 0) + (v.w != 0);
  }
  if (c) atomicAdd(out, c);
}

static hipMemAllocationProp make_prop(int dev, bool exportable) {
  hipMemAllocationProp prop{};
  prop.type = hipMemAllocationTypePinned;
  prop.requestedHandleType =
      exportable ? hipMemHandleTypePosixFileDescriptor : hipMemHandleTypeNone;
  prop.location.type = hipMemLocationTypeDevice;
  prop.location.id = dev;
  return prop;
}
static hipMemAccessDesc make_access(int dev) {
  hipMemAccessDesc d{};
  d.location.type = hipMemLocationTypeDevice;
  d.location.id = dev;
  d.flags = hipMemAccessFlagsProtReadWrite;
  return d;
}

int main(int argc, char **argv) {
  int dev = 0;
  int N = argc > 1 ? atoi(argv[1]) : 1024;
  TRY(hipInit(0));
  TRY(hipSetDevice(dev));
  hipDeviceProp_t dp;
  TRY(hipGetDeviceProperties(&dp, dev));
  printf("device: %s arch=%s CUs=%d totalGlobalMem=%.1f GiB\n", dp.name, dp.gcnArchName,
         dp.multiProcessorCount, dp.totalGlobalMem / (double)GiB);

  int vmm = 0;
  TRY(hipDeviceGetAttribute(&vmm, hipDeviceAttributeVirtualMemoryManagementSupported, dev));
  printf("VMM supported: %d\n", vmm);
  auto prop = make_prop(dev, false);
  auto propx = make_prop(dev, true);
  auto acc = make_access(dev);
  size_t gmin = 0, grec = 0;
  TRY(hipMemGetAllocationGranularity(&gmin, &prop, hipMemAllocationGranularityMinimum));
  TRY(hipMemGetAllocationGranularity(&grec, &prop, hipMemAllocationGranularityRecommended));
  printf("granularity: min=%zu recommended=%zu\n", gmin, grec);

  {
    Stat s;
    for (int i = 0; i < 200; i++) {
      size_t f, t;
      auto t0 = clk::now();
      (void)hipMemGetInfo(&f, &t);
      s.add(us_since(t0));
      if (i == 0) printf("hipMemGetInfo: free=%.2f GiB total=%.2f GiB\n", f / (double)GiB, t / (double)GiB);
    }
    s.print("hipMemGetInfo");
  }

  // --- 1. reserve with hint
  void *hint = (void *)0x1f000000000ull;
  void *va = nullptr;
  size_t va_size = 64 * GiB;
  {
    auto t0 = clk::now();
    TRY(hipMemAddressReserve(&va, va_size, PAGE, hint, 0));
    printf("address_reserve 64 GiB: %.1f us, got %p (hint %p, honored=%d)\n", us_since(t0), va, hint, va == hint);
  }
  if (!va) return 1;
  char *base = (char *)va;

  // --- 2. per-call latencies, 2 MiB pages
  std::vector<hipMemGenericAllocationHandle_t> h(N);
  {
    Stat sc, sm, sa, su, sr;
    for (int i = 0; i < N; i++) {
      auto t0 = clk::now();
      TRY(hipMemCreate(&h[i], PAGE, &prop, 0));
      sc.add(us_since(t0));
    }
    for (int i = 0; i < N; i++) {
      auto t0 = clk::now();
      TRY(hipMemMap(base + (size_t)i * PAGE, PAGE, 0, h[i], 0));
      sm.add(us_since(t0));
    }
    for (int i = 0; i < N; i++) {
      auto t0 = clk::now();
      TRY(hipMemSetAccess(base + (size_t)i * PAGE, PAGE, &acc, 1));
      sa.add(us_since(t0));
    }
    // freshly created memory: zero?
    unsigned long long *cnt;
    TRY(hipMalloc(&cnt, 8));
    TRY(hipMemset(cnt, 0, 8));
    count_nonzero<<<2048, 256>>>((const uint4 *)base, (size_t)N * PAGE / 16, cnt);
    unsigned long long hc = 0;
    TRY(hipMemcpy(&hc, cnt, 8, hipMemcpyDeviceToHost));
    printf("fresh hipMemCreate pages: nonzero dwords = %llu of %zu\n", hc, (size_t)N * PAGE / 4);
    // poison, so recycled pages are visibly dirty
    poison<<<2048, 256>>>((uint4 *)base, (size_t)N * PAGE / 16);
    TRY(hipDeviceSynchronize());

    for (int i = 0; i < N; i++) {
      auto t0 = clk::now();
      TRY(hipMemUnmap(base + (size_t)i * PAGE, PAGE));
      su.add(us_since(t0));
    }
    // remap same handles (pool reuse): map + one coalesced set_access
    {
      auto t0 = clk::now();
      for (int i = 0; i < N; i++) TRY(hipMemMap(base + (size_t)i * PAGE, PAGE, 0, h[i], 0));
      double tm = us_since(t0);
      t0 = clk::now();
      hipError_t e = TRY(hipMemSetAccess(base, (size_t)N * PAGE, &acc, 1));
      double ta = us_since(t0);
      printf("pool remap %d pages: map total %.1f us (%.2f/page); ONE set_access over range: %.1f us (%.3f/page) ok=%d\n",
             N, tm, tm / N, ta, ta / N, e == hipSuccess);
      TRY(hipMemset(cnt, 0, 8));
      count_nonzero<<<2048, 256>>>((const uint4 *)base, (size_t)N * PAGE / 16, cnt);
      TRY(hipMemcpy(&hc, cnt, 8, hipMemcpyDeviceToHost));
      printf("recycled (poisoned, unmapped, remapped) pages: nonzero dwords = %llu (expect all = %zu)\n", hc, (size_t)N * PAGE / 4);
      // one unmap over whole range?
      t0 = clk::now();
      e = hipMemUnmap(base, (size_t)N * PAGE);
      printf("ONE unmap over %d mappings: %s in %.1f us\n", N, hipGetErrorString(e), us_since(t0));
      if (e != hipSuccess)
        for (int i = 0; i < N; i++) TRY(hipMemUnmap(base + (size_t)i * PAGE, PAGE));
    }
    for (int i = 0; i < N; i++) {
      auto t0 = clk::now();
      TRY(hipMemRelease(h[i]));
      sr.add(us_since(t0));
    }
    sc.print("mem_create 2MiB");
    sm.print("mem_map 2MiB");
    sa.print("set_access 2MiB");
    su.print("mem_unmap 2MiB");
    sr.print("mem_release 2MiB");
    TRY(hipFree(cnt));
  }

  // --- 3. big handles
  for (size_t sz : {8 * MiB, 64 * MiB, 128 * MiB, 1024 * MiB}) {
    hipMemGenericAllocationHandle_t hb;
    auto t0 = clk::now();
    if (TRY(hipMemCreate(&hb, sz, &prop, 0)) != hipSuccess) continue;
    double tc = us_since(t0);
    t0 = clk::now();
    TRY(hipMemMap(base, sz, 0, hb, 0));
    double tm = us_since(t0);
    t0 = clk::now();
    TRY(hipMemSetAccess(base, sz, &acc, 1));
    double ta = us_since(t0);
    // partial unmap?
    t0 = clk::now();
    hipError_t e = hipMemUnmap(base, PAGE);
    double tpu = us_since(t0);
    printf("handle %4zu MiB: create %.1f map %.1f set_access %.1f us; partial unmap(2MiB) -> %s (%.1f us)\n",
           sz / MiB, tc, tm, ta, hipGetErrorString(e), tpu);
    t0 = clk::now();
    if (e == hipSuccess) {
      // unmap the rest
      TRY(hipMemUnmap(base + PAGE, sz - PAGE));
    } else {
      TRY(hipMemUnmap(base, sz));
    }
    double tu = us_since(t0);
    // sub-range map of a big handle with an offset
    e = hipMemMap(base, PAGE, PAGE, hb, 0);
    printf("   unmap %.1f us; map(size=2MiB, offset=2MiB) of big handle -> %s\n", tu, hipGetErrorString(e));
    if (e == hipSuccess) TRY(hipMemUnmap(base, PAGE));
    e = hipMemMap(base, PAGE, 0, hb, 0);
    printf("   map(size=2MiB < handle size, offset=0) -> %s\n", hipGetErrorString(e));
    if (e == hipSuccess) TRY(hipMemUnmap(base, PAGE));
    t0 = clk::now();
    TRY(hipMemRelease(hb));
    printf("   release %.1f us\n", us_since(t0));
  }

  // --- 4. one handle aliased at many VAs (the reference's zero page)
  {
    hipMemGenericAllocationHandle_t hz;
    TRY(hipMemCreate(&hz, PAGE, &prop, 0));
    int K = 256;
    auto t0 = clk::now();
    int ok = 0;
    for (int i = 0; i < K; i++) ok += hipMemMap(base + (size_t)i * PAGE, PAGE, 0, hz, 0) == hipSuccess;
    double tm = us_since(t0);
    t0 = clk::now();
    hipError_t e = TRY(hipMemSetAccess(base, (size_t)K * PAGE, &acc, 1));
    double ta = us_since(t0);
    printf("zero-page aliasing: %d/%d maps ok, %.2f us/map, one set_access %.1f us (%s)\n", ok, K, tm / K, ta, hipGetErrorString(e));
    if (e == hipSuccess) {
      // write through alias 0, read through alias K-1
      unsigned int v = 0xDEADBEEF, r = 0;
      TRY(hipMemcpy(base + 64, &v, 4, hipMemcpyHostToDevice));
      TRY(hipMemcpy(&r, base + (size_t)(K - 1) * PAGE + 64, 4, hipMemcpyDeviceToHost));
      printf("   alias readback 0x%x (aliased=%d)\n", r, r == v);
    }
    t0 = clk::now();
    for (int i = 0; i < K; i++) hipMemUnmap(base + (size_t)i * PAGE, PAGE);
    printf("   unmap %.2f us/page\n", us_since(t0) / K);
    TRY(hipMemRelease(hz));
  }

  // --- 5. exportable handles
  {
    hipMemGenericAllocationHandle_t hx, hi;
    auto t0 = clk::now();
    if (TRY(hipMemCreate(&hx, PAGE, &propx, 0)) == hipSuccess) {
      double tc = us_since(t0);
      int fd = -1;
      t0 = clk::now();
      hipError_t e = TRY(hipMemExportToShareableHandle(&fd, hx, hipMemHandleTypePosixFileDescriptor, 0));
      double te = us_since(t0);
      printf("exportable create %.1f us; export -> fd=%d (%s) %.1f us\n", tc, fd, hipGetErrorString(e), te);
      if (e == hipSuccess) {
        t0 = clk::now();
        e = TRY(hipMemImportFromShareableHandle(&hi, (void *)(uintptr_t)fd, hipMemHandleTypePosixFileDescriptor));
        printf("import (same process) -> %s %.1f us\n", hipGetErrorString(e), us_since(t0));
        if (e == hipSuccess) {
          TRY(hipMemMap(base, PAGE, 0, hx, 0));
          TRY(hipMemMap(base + PAGE, PAGE, 0, hi, 0));
          TRY(hipMemSetAccess(base, 2 * PAGE, &acc, 1));
          unsigned int v = 0x12345678, r = 0;
          TRY(hipMemcpy(base + 128, &v, 4, hipMemcpyHostToDevice));
          TRY(hipMemcpy(&r, base + PAGE + 128, 4, hipMemcpyDeviceToHost));
          printf("   imported alias readback 0x%x (shared=%d)\n", r, r == v);
          TRY(hipMemUnmap(base, PAGE));
          TRY(hipMemUnmap(base + PAGE, PAGE));
          TRY(hipMemRelease(hi));
        }
        close(fd);
      }
      TRY(hipMemRelease(hx));
    }
  }

  // --- 6. threads: T threads each create+map+access N/T pages in disjoint VA slices
  for (int T : {1, 2, 4, 8}) {
    std::vector<hipMemGenericAllocationHandle_t> hh(N);
    for (int i = 0; i < N; i++) TRY(hipMemCreate(&hh[i], PAGE, &prop, 0));
    auto t0 = clk::now();
    std::vector<std::thread> th;
    for (int t = 0; t < T; t++)
      th.emplace_back([&, t]() {
        (void)hipSetDevice(dev);
        int lo = N * t / T, hi = N * (t + 1) / T;
        for (int i = lo; i < hi; i++) (void)hipMemMap(base + (size_t)i * PAGE, PAGE, 0, hh[i], 0);
        (void)hipMemSetAccess(base + (size_t)lo * PAGE, (size_t)(hi - lo) * PAGE, &acc, 1);
      });
    for (auto &x : th) x.join();
    double tmap = us_since(t0);
    t0 = clk::now();
    th.clear();
    for (int t = 0; t < T; t++)
      th.emplace_back([&, t]() {
        (void)hipSetDevice(dev);
        int lo = N * t / T, hi = N * (t + 1) / T;
        for (int i = lo; i < hi; i++) (void)hipMemUnmap(base + (size_t)i * PAGE, PAGE);
      });
    for (auto &x : th) x.join();
    double tun = us_since(t0);
    printf("threads=%d: map+access %d pages %.1f us (%.2f us/page, %.1f GB/s backed); unmap %.1f us (%.2f/page)\n", T, N,
           tmap, tmap / N, (double)N * PAGE / tmap / 1e3, tun, tun / N);
    for (int i = 0; i < N; i++) TRY(hipMemRelease(hh[i]));
  }

  // --- 7. fill-kernel rates on a mapped 2 GiB range (N pages)
  {
    std::vector<hipMemGenericAllocationHandle_t> hh(N);
    for (int i = 0; i < N; i++) TRY(hipMemCreate(&hh[i], PAGE, &prop, 0));
    for (int i = 0; i < N; i++) TRY(hipMemMap(base + (size_t)i * PAGE, PAGE, 0, hh[i], 0));
    TRY(hipMemSetAccess(base, (size_t)N * PAGE, &acc, 1));
    size_t bytes = (size_t)N * PAGE;
    hipEvent_t e0, e1;
    TRY(hipEventCreate(&e0));
    TRY(hipEventCreate(&e1));
    auto timeit = [&](const char *name, auto fn) {
      for (int i = 0; i < 3; i++) fn();
      TRY(hipDeviceSynchronize());
      TRY(hipEventRecord(e0, 0));
      int R = 10;
      for (int i = 0; i < R; i++) fn();
      TRY(hipEventRecord(e1, 0));
      TRY(hipEventSynchronize(e1));
      float ms = 0;
      TRY(hipEventElapsedTime(&ms, e0, e1));
      printf("fill %-28s %.1f us / %zu MiB = %.1f GB/s\n", name, ms * 1e3 / R, bytes / MiB, bytes / (ms / R * 1e-3) / 1e9);
    };
    for (int grid : {1024, 2048, 4096, 8192, 32768})
      for (int blk : {256, 512}) {
        char nm[64];
        snprintf(nm, sizeof nm, "plain g=%d b=%d", grid, blk);
        timeit(nm, [&]() { fill_plain<<<grid, blk>>>((uint4 *)base, bytes / 16); });
        snprintf(nm, sizeof nm, "nt    g=%d b=%d", grid, blk);
        timeit(nm, [&]() { fill_nt<<<grid, blk>>>((uint4 *)base, bytes / 16); });
      }
    timeit("hipMemsetAsync", [&]() { (void)hipMemsetAsync(base, 0, bytes, 0); });
    for (int i = 0; i < N; i++) TRY(hipMemUnmap(base + (size_t)i * PAGE, PAGE));
    for (int i = 0; i < N; i++) TRY(hipMemRelease(hh[i]));
  }

  TRY(hipMemAddressFree(va, va_size));
  printf("probe done\n");
  return 0;
}
