// kernels.hpp — host-side launch interface of the gfx950 kernels (kernels.hip: byte movers; index_kernels.hip:
// block id <-> token index glue).
#pragma once

#include <hip/hip_runtime.h>

#include <cstddef>
#include <cstdint>

namespace kvc {

// Bytes one workgroup zeroes / the alignment unit of zero_fill_pages.
static constexpr size_t kFillSlabBytes = 64 * 1024;
// Pointers (pages / regions) and moves one launch carries in its kernarg segment.
static constexpr int kMaxPtrsPerLaunch = 1024; // 8 KiB of kernarg: a whole 1024-page batch is ONE launch (no drain/ramp between 4)
static constexpr int kMaxRegionsPerLaunch = 128;
static constexpr int kMaxMovesPerLaunch = 448; // 448 x 16 B + 128 x 8 B = 8 KiB of kernarg (32 KiB / 1984 moves per launch was tried in round 3: fewer launches, but no faster on 32 KiB blocks and 8-10 % slower on 16-18 KiB blocks - every workgroup reads its three entries through the scalar cache)
static constexpr int kMaxIdsPerLaunch = 1024; // block ids one index-kernel launch carries in its kernarg (8 KiB)

// Zero `n` pages of `page_bytes` each (n <= kMaxPtrsPerLaunch). Asynchronous on `stream`.
// variant: 0 = default; others are tuning variants kept for A/B runs.
hipError_t launch_zero_fill_pages(void *const *pages, int n, size_t page_bytes, hipStream_t stream, int variant = 0);

// Copy blocks inside each region: for r < n_regions, m < n_moves:
//   bases[r] + dst[m]*block_bytes  <-  bases[r] + src[m]*block_bytes
// n_regions <= kMaxRegionsPerLaunch, n_moves <= kMaxMovesPerLaunch. variant 0 (= 9) = LDS-staged
// (LDS-DMA in, ds_read_b128 + global_store out), a contiguous eighth of the (region, move) pairs per XCD, non-temporal accesses,
// 32 KiB tiles; 10 = the same register-staged; 6 / 7 = LDS / register-staged with every eighth pair per XCD (the default of most of round 3); 8 and 1 = LDS / register-staged with 16 KiB tiles (8: the default until round 3); 2/3 = LDS/register-staged
// with the plain interleaved placement and temporal accesses; 4/5 = LDS/register-staged XCD-aware with temporal accesses (A/B runs).
hipError_t launch_compact_blocks(void *const *bases, int n_regions, const int64_t *src, const int64_t *dst, int n_moves,
                                 size_t block_bytes, hipStream_t stream, int variant = 0);

// One lane: optionally store `value` to *p, then copy *p to *out_dev (the TLB self test's view of memory).
hipError_t launch_peek_poke(void *p, unsigned *out_dev, unsigned value, bool do_write, hipStream_t stream);

// ---- index_kernels.hip. Block ids come either from the host (`ids_host`, <= kMaxIdsPerLaunch, carried in the
// kernarg) or from device memory (`ids_dev` != nullptr, any count). All asynchronous on `stream`.
// out[i*tpb + j] = ids[i]*tpb + j
hipError_t launch_expand_block_ids(const int64_t *ids_host, const int64_t *ids_dev, size_t n_ids, int64_t tpb,
                                   int64_t *out, hipStream_t stream);
// token slots for requests growing pre_lens[r] -> seq_lens[r] (see the kernel); out has out_len entries
hipError_t launch_alloc_extend(const int64_t *ids_host, const int64_t *ids_dev, size_t n_ids, const int64_t *pre_lens,
                               const int64_t *seq_lens, const int64_t *last_loc, size_t bs, int64_t tpb, int64_t *out,
                               size_t out_len, hipStream_t stream);
// token slot of the one new token of every request; out has bs entries
hipError_t launch_alloc_decode(const int64_t *ids_host, const int64_t *ids_dev, size_t n_ids, const int64_t *seq_lens,
                               const int64_t *last_loc, size_t bs, int64_t tpb, int64_t *out, hipStream_t stream);
// result[0] = number of distinct blocks (or -(number of out-of-range indices)), result[1..] = ascending block ids.
// `bitmap` (>= ceil(n_blocks/32) zeroed words) and `header` (16 bytes {0xffffffff,0,0,0}) are left in that state.
hipError_t launch_unique_block_ids(const int64_t *idx, size_t n, int64_t tpb, int64_t n_blocks, unsigned *bitmap,
                                   void *header, int64_t *result, size_t cap, hipStream_t stream);

} // namespace kvc
