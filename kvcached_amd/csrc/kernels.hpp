// kernels.hpp — host-side launch interface of the two gfx950 kernels (kernels.hip).
#pragma once

#include <hip/hip_runtime.h>

#include <cstddef>
#include <cstdint>

namespace kvc {

// Bytes one workgroup zeroes / the alignment unit of zero_fill_pages.
static constexpr size_t kFillSlabBytes = 64 * 1024;
// Pointers (pages / regions) and moves one launch carries in its kernarg segment.
static constexpr int kMaxPtrsPerLaunch = 256;
static constexpr int kMaxRegionsPerLaunch = 128;
static constexpr int kMaxMovesPerLaunch = 448; // 448 x 16 B + 128 x 8 B = 8 KiB of kernarg

// Zero `n` pages of `page_bytes` each (n <= kMaxPtrsPerLaunch). Asynchronous on `stream`.
// variant: 0 = default; others are tuning variants kept for A/B runs.
hipError_t launch_zero_fill_pages(void *const *pages, int n, size_t page_bytes, hipStream_t stream, int variant = 0);

// Copy blocks inside each region: for r < n_regions, m < n_moves:
//   bases[r] + dst[m]*block_bytes  <-  bases[r] + src[m]*block_bytes
// n_regions <= kMaxRegionsPerLaunch, n_moves <= kMaxMovesPerLaunch. variant 0 = LDS-staged
// (LDS-DMA in, ds_read_b128 + global_store out) with XCD-aware block placement, 1 = register-staged
// XCD-aware, 2/3 = the same two with the plain interleaved placement (A/B runs).
hipError_t launch_compact_blocks(void *const *bases, int n_regions, const int64_t *src, const int64_t *dst, int n_moves,
                                 size_t block_bytes, hipStream_t stream, int variant = 0);

} // namespace kvc
