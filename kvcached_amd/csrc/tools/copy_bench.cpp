// copy_bench.cpp — what is the copy (read+write) ceiling of one MI355X, and which kernel shape reaches it?
// Contiguous 2 GiB -> 2 GiB device copies: hipMemcpyAsync D2D vs kernels that differ in bytes in flight per
// thread, tile size per workgroup, temporal hints and block->tile mapping. Guides compact_blocks' tiling.
// build: hipcc --offload-arch=gfx950 -O3 -o copy_bench copy_bench.cpp
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x)                                                                                                          \
  do {                                                                                                                 \
    hipError_t e_ = (x);                                                                                               \
    if (e_ != hipSuccess) {                                                                                            \
      fprintf(stderr, "%s:%d %s -> %s\n", __FILE__, __LINE__, #x, hipGetErrorString(e_));                              \
      exit(1);                                                                                                         \
    }                                                                                                                  \
  } while (0)

typedef unsigned int v4u __attribute__((ext_vector_type(4)));

// Each workgroup copies TILE bytes: THREADS lanes x UNROLL x 16 B per pass, all loads of a pass issued before the stores.
template <int THREADS, int UNROLL, int TILE, bool NT_LOAD, bool NT_STORE, bool XCD>
__global__ __launch_bounds__(THREADS) void copy_kernel(const v4u *__restrict__ src, v4u *__restrict__ dst, unsigned n_tiles) {
  unsigned tile = blockIdx.x;
  if (XCD) { // XCD x owns a contiguous eighth of the tiles
    const unsigned x = blockIdx.x & 7u, i = blockIdx.x >> 3, per = (n_tiles + 7) / 8;
    tile = x * per + i;
    if (i >= per || tile >= n_tiles) return;
  }
  constexpr int VEC_PER_TILE = TILE / 16;
  constexpr int PASSES = VEC_PER_TILE / (THREADS * UNROLL);
  const v4u *s = src + (size_t)tile * VEC_PER_TILE + threadIdx.x;
  v4u *d = dst + (size_t)tile * VEC_PER_TILE + threadIdx.x;
#pragma unroll
  for (int p = 0; p < PASSES; ++p) {
    v4u r[UNROLL];
#pragma unroll
    for (int u = 0; u < UNROLL; ++u)
      r[u] = NT_LOAD ? __builtin_nontemporal_load(s + (p * UNROLL + u) * THREADS) : s[(p * UNROLL + u) * THREADS];
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) {
      if (NT_STORE)
        __builtin_nontemporal_store(r[u], d + (p * UNROLL + u) * THREADS);
      else
        d[(p * UNROLL + u) * THREADS] = r[u];
    }
  }
}

// grid-stride persistent variant
template <int THREADS, int UNROLL>
__global__ __launch_bounds__(THREADS) void copy_persistent(const v4u *__restrict__ src, v4u *__restrict__ dst, size_t n_vec) {
  const size_t stride = (size_t)gridDim.x * THREADS * UNROLL;
  for (size_t base = (size_t)blockIdx.x * THREADS * UNROLL + threadIdx.x; base < n_vec; base += stride) {
    v4u r[UNROLL];
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) r[u] = src[base + (size_t)u * THREADS];
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) dst[base + (size_t)u * THREADS] = r[u];
  }
}

template <class F> static void timeit(const char *name, size_t bytes, hipStream_t s, F &&f) {
  hipEvent_t a, b;
  CK(hipEventCreate(&a));
  CK(hipEventCreate(&b));
  for (int i = 0; i < 3; ++i) f();
  CK(hipStreamSynchronize(s));
  const int reps = 10;
  CK(hipEventRecord(a, s));
  for (int i = 0; i < reps; ++i) f();
  CK(hipEventRecord(b, s));
  CK(hipEventSynchronize(b));
  float ms = 0;
  CK(hipEventElapsedTime(&ms, a, b));
  printf("%-52s %8.1f us/rep  %7.1f GB/s (read+write)\n", name, ms / reps * 1e3, 2.0 * bytes * reps / (ms * 1e-3) / 1e9);
  fflush(stdout);
}

#define RUN(THREADS, UNROLL, TILE, NTL, NTS, XCD)                                                                      \
  timeit("kernel thr=" #THREADS " unroll=" #UNROLL " tile=" #TILE " ntl=" #NTL " nts=" #NTS " xcd=" #XCD, bytes, s, [&] { \
    const unsigned n_tiles = (unsigned)(bytes / (TILE));                                                               \
    const unsigned grid = (XCD) ? ((n_tiles + 7) / 8) * 8 : n_tiles;                                                   \
    copy_kernel<THREADS, UNROLL, TILE, NTL, NTS, XCD><<<grid, THREADS, 0, s>>>((const v4u *)src, (v4u *)dst, n_tiles);  \
  })

int main() {
  CK(hipSetDevice(0));
  hipStream_t s;
  CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  const size_t bytes = 2ull << 30;
  char *src, *dst;
  CK(hipMalloc(&src, bytes));
  CK(hipMalloc(&dst, bytes));
  CK(hipMemsetAsync(src, 1, bytes, s));
  CK(hipMemsetAsync(dst, 2, bytes, s));
  CK(hipStreamSynchronize(s));
  timeit("hipMemcpyAsync D2D", bytes, s, [&] { CK(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, s)); });
  RUN(256, 4, 16384, false, false, false);
  RUN(256, 4, 16384, false, false, true);
  RUN(256, 8, 32768, false, false, false);
  RUN(256, 8, 32768, false, false, true);
  RUN(256, 4, 32768, false, false, true);
  RUN(256, 8, 65536, false, false, true);
  RUN(256, 16, 65536, false, false, true);
  RUN(512, 4, 32768, false, false, true);
  RUN(512, 8, 65536, false, false, true);
  RUN(1024, 4, 65536, false, false, true);
  RUN(256, 8, 32768, true, false, true);
  RUN(256, 8, 32768, false, true, true);
  RUN(256, 8, 32768, true, true, true);
  RUN(256, 2, 8192, false, false, true);
  RUN(64, 8, 8192, false, false, true);
  RUN(128, 8, 16384, false, false, true);
  for (int grid : {2048, 4096, 8192}) {
    char nm[64];
    snprintf(nm, sizeof nm, "persistent grid=%d thr=256 unroll=8", grid);
    timeit(nm, bytes, s, [&] { copy_persistent<256, 8><<<grid, 256, 0, s>>>((const v4u *)src, (v4u *)dst, bytes / 16); });
  }
  timeit("hipMemcpyAsync D2D (again)", bytes, s, [&] { CK(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, s)); });
  return 0;
}
