// kernels.hip — the two device kernels of the path, written for gfx950 (CDNA4) only.
//
// Both are pure byte movers and HBM-bound (no MFMA: there is no contraction anywhere on this
// path). What matters on MI355X, and what the code below does:
//   * every wave-instruction touches 64 lanes x 16 B = 1 KiB of contiguous memory
//     (global_store_dwordx4 / global_load_dwordx4 / global_load_lds_dwordx4);
//   * one workgroup per 64 KiB slab (fill) or per <=32 KiB tile (compaction) and NO grid-stride
//     loop: with 256 CUs in 8 XCDs a batch is tens of thousands of workgroups, and the probe
//     showed short-lived workgroups beating a persistent 2048-block grid (6.9 vs 6.1 TB/s);
//   * the page / region tables ride in the kernarg segment (scalar loads, no table in HBM,
//     no extra copy on the launch path);
//   * the blockIdx -> work mapping is XCD-aware: each XCD (own L2) owns whole pages instead of
//     every XCD touching every page (+12 % measured, see zero_fill_pages below).
//
// Algorithmic bytes: zero_fill_pages writes page_bytes per page (2 097 152 B
// for a 2 MiB page) and reads nothing; compact_blocks reads + writes block_bytes per
// (region, move) = 2 x block_bytes x n_regions per moved block.

#include "kernels.hpp"

namespace kvc {

typedef unsigned int v4u __attribute__((ext_vector_type(4)));

struct PageTable {
  void *p[kMaxPtrsPerLaunch];
};

// ---------------------------------------------------------------------------- zero_fill_pages
// One workgroup zeroes one 64 KiB slab: THREADS x STORES x 16 B, lane l of the workgroup stores at
// slab + (i*THREADS + l)*16, so every store instruction of a wave covers one aligned 1 KiB run.
//
// blockIdx -> (page, slab) is XCD-aware. Workgroups are dealt round-robin over the 8 XCDs (block b
// runs on XCD b % 8), so the naive "consecutive blocks take consecutive slabs" makes all eight
// XCD L2s write into every page at once. Here XCD x owns pages x, x+8, x+16, ... completely:
//     x = b % 8, i = b / 8, page = (i / slabs_per_page) * 8 + x, slab = i % slabs_per_page
// Measured on MI355X (tools/fill_bench.cpp, 2 GiB of 2 MiB pages, sequential and shuffled lists):
// 6.2 TB/s with this mapping vs 5.5 TB/s interleaved; hipMemsetAsync reaches 6.4 TB/s on the same
// (contiguous) range. The placement is a speed heuristic only: any block->XCD assignment is correct.
// XCD_BLOCKED (variant 4, A/B): XCD x owns the pages [x * ceil(n/8), (x+1) * ceil(n/8)) - a contiguous eighth of the list (the
// list arrives sorted into runs of up to 64 adjacent pages of one physical extent) - instead of every eighth page.
template <int THREADS, bool NT, bool XCD_PAGES, bool XCD_BLOCKED = false>
__global__ __launch_bounds__(THREADS) void zero_fill_pages_kernel(PageTable pages, unsigned slabs_per_page,
                                                                   unsigned n_pages) {
  constexpr int STORES = (int)(kFillSlabBytes / 16 / THREADS);
  unsigned page, slab;
  if (XCD_PAGES) {
    const unsigned x = blockIdx.x & 7u, i = blockIdx.x >> 3;
    const unsigned q = i / slabs_per_page;
    slab = i - q * slabs_per_page;
    page = XCD_BLOCKED ? x * ((n_pages + 7u) >> 3) + q : q * 8u + x;
    if (page >= n_pages) return; // the grid is padded to a multiple of 8 pages (wave-uniform exit)
  } else {
    page = blockIdx.x / slabs_per_page;
    slab = blockIdx.x - page * slabs_per_page;
  }
  v4u *dst = reinterpret_cast<v4u *>(static_cast<char *>(pages.p[page]) + (size_t)slab * kFillSlabBytes) + threadIdx.x;
  const v4u z = {0u, 0u, 0u, 0u};
#pragma unroll
  for (int i = 0; i < STORES; ++i) {
    if (NT)
      __builtin_nontemporal_store(z, dst + i * THREADS);
    else
      dst[i * THREADS] = z;
  }
}

hipError_t launch_zero_fill_pages(void *const *pages, int n, size_t page_bytes, hipStream_t stream, int variant) {
  if (n <= 0) return hipSuccess;
  if (n > kMaxPtrsPerLaunch || page_bytes == 0 || page_bytes % kFillSlabBytes != 0) return hipErrorInvalidValue;
  // The order of the table decides which pages are written AT THE SAME TIME (the ~64 pages the chip has in flight are consecutive
  // table entries), and the memory system likes those scattered: a list of adjacent pages in address order fills at 6.2 TB/s, the
  // same pages shuffled at 7.0; runs of 64 adjacent pages in shuffled order - what the allocator's sorted batches look like - at
  // 6.6-6.7 (benchmarks/probe_fill_rate_per_buffer.py). A regular stride is not enough (i * 0.618 n mod n: 6.4 on the sorted
  // list), so the table is laid out in a fixed pseudo-random order (Fisher-Yates with a constant seed, the same for every
  // launch of the same size): whatever order the caller's list has, the kernel sees a shuffled one. Zeroing does not care about
  // the order. Variant 5 keeps the caller's order (A/B runs).
  PageTable t;
  static thread_local int perm_n = 0;
  static thread_local unsigned short perm[kMaxPtrsPerLaunch];
  const bool shuffle = variant != 5 && n > 16;
  if (shuffle && perm_n != n) {
    for (int i = 0; i < n; ++i) perm[i] = (unsigned short)i;
    uint64_t x = 0x9e3779b97f4a7c15ull; // (splitmix-style steps; any fixed sequence will do)
    for (int i = n - 1; i > 0; --i) {
      x += 0x9e3779b97f4a7c15ull;
      uint64_t z = x;
      z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull;
      z = (z ^ (z >> 27)) * 0x94d049bb133111ebull;
      z ^= z >> 31;
      const int j = (int)(z % (uint64_t)(i + 1));
      const unsigned short tmp = perm[i];
      perm[i] = perm[j];
      perm[j] = tmp;
    }
    perm_n = n;
  }
  for (int i = 0; i < n; ++i) {
    if (reinterpret_cast<uintptr_t>(pages[i]) % 16 != 0) return hipErrorInvalidValue;
    t.p[shuffle ? perm[i] : i] = pages[i];
  }
  const unsigned slabs = (unsigned)(page_bytes / kFillSlabBytes);
  const size_t padded = ((size_t)n + 7) / 8 * 8;
  const size_t grid_xcd = (size_t)slabs * padded, grid_flat = (size_t)slabs * (size_t)n;
  if (grid_xcd > 0x7fffffffull) return hipErrorInvalidValue;
  switch (variant) {
  case 1: // interleaved slabs (the pre-XCD-aware mapping), kept for A/B runs
    zero_fill_pages_kernel<512, false, false><<<dim3((unsigned)grid_flat), dim3(512), 0, stream>>>(t, slabs, (unsigned)n);
    break;
  case 2: // non-temporal stores: slower on gfx950 (5.2 vs 6.2 TB/s), kept as evidence
    zero_fill_pages_kernel<512, true, true><<<dim3((unsigned)grid_xcd), dim3(512), 0, stream>>>(t, slabs, (unsigned)n);
    break;
  case 3:
    zero_fill_pages_kernel<1024, false, true><<<dim3((unsigned)grid_xcd), dim3(1024), 0, stream>>>(t, slabs, (unsigned)n);
    break;
  case 5: // the default kernel on the caller's order of pages (no stride permutation of the table)
    zero_fill_pages_kernel<512, false, true><<<dim3((unsigned)grid_xcd), dim3(512), 0, stream>>>(t, slabs, (unsigned)n);
    break;
  case 4: // a contiguous eighth of the page list per XCD
    zero_fill_pages_kernel<512, false, true, true><<<dim3((unsigned)grid_xcd), dim3(512), 0, stream>>>(t, slabs, (unsigned)n);
    break;
  default:
    zero_fill_pages_kernel<512, false, true><<<dim3((unsigned)grid_xcd), dim3(512), 0, stream>>>(t, slabs, (unsigned)n);
  }
  return hipGetLastError();
}

// ---------------------------------------------------------------------------- peek_poke (init self test)
// One lane writes `value` to *p (if do_write) and copies *p to *out, with system-scope loads/stores that bypass nothing
// they should not: what the TLB self test of KvAllocator::init looks at memory with (through the GPU's own address
// translation, which is the thing under test - a host copy would go another way).
__global__ void peek_poke_kernel(unsigned *p, unsigned *out, unsigned value, int do_write) {
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    if (do_write) __hip_atomic_store(p, value, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    *out = __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  }
}
hipError_t launch_peek_poke(void *p, unsigned *out_dev, unsigned value, bool do_write, hipStream_t stream) {
  peek_poke_kernel<<<dim3(1), dim3(64), 0, stream>>>(static_cast<unsigned *>(p), out_dev, value, do_write ? 1 : 0);
  return hipGetLastError();
}

// ---------------------------------------------------------------------------- compact_blocks
struct CompactArgs {
  void *base[kMaxRegionsPerLaunch];
  int64_t src[kMaxMovesPerLaunch];
  int64_t dst[kMaxMovesPerLaunch];
};
static_assert(sizeof(CompactArgs) <= 8192, "kernarg segment budget");

static constexpr int kCompactThreads = 256;          // 4 waves
static constexpr int kCompactWaves = kCompactThreads / 64;
static constexpr int kCompactTile = 16 * 1024;       // bytes per workgroup: 4 waves x 4 x 1 KiB
static constexpr int kPiece = 1024;                  // one wave-instruction
static constexpr int kPiecesPerWave = kCompactTile / kPiece / (kCompactThreads / 64);
// variants 6 / 7 (A/B): twice the bytes in flight per wave - 32 KiB per workgroup, 8 pieces per wave
static constexpr int kCompactTileBig = 32 * 1024;
static constexpr int kPiecesPerWaveBig = kCompactTileBig / kPiece / (kCompactThreads / 64);

// blockIdx -> (region, move, tile). XCD == true: the (region, move) pairs are dealt to the 8 XCDs so that one
// XCD copies a whole block (all its tiles) — the same placement idea as zero_fill_pages; speed only.
template <bool XCD>
__device__ __forceinline__ bool compact_index(unsigned n_moves, unsigned n_regions, unsigned tiles_per_block, unsigned &r,
                                              unsigned &m, unsigned &t) {
  if (XCD) {
    const unsigned x = blockIdx.x & 7u, i = blockIdx.x >> 3;
    const unsigned q = i / tiles_per_block;
    t = i - q * tiles_per_block;
    const unsigned pair = q * 8u + x;
    if (pair >= n_moves * n_regions) return false;
    r = pair / n_moves;
    m = pair - r * n_moves;
  } else {
    t = blockIdx.x % tiles_per_block;
    m = (blockIdx.x / tiles_per_block) % n_moves;
    r = blockIdx.x / (tiles_per_block * n_moves);
  }
  return true;
}

// Variant 0 — LDS-staged. Each wave DMAs its four 1 KiB pieces global->LDS
// (global_load_lds_dwordx4: per-lane source address, wave-uniform LDS base + lane*16), waits
// for its own DMAs (vmcnt), reads them back with ds_read_b128 and streams them out with
// global_store_dwordx4. A wave only ever reads LDS bytes it loaded itself, so no barrier is
// needed; 16 KiB of LDS per workgroup lets 8+ workgroups share a CU (>=128 KiB in flight).
template <bool XCD, bool NT>
__global__ __launch_bounds__(kCompactThreads) void compact_blocks_lds_kernel(CompactArgs a, unsigned n_moves,
                                                                              unsigned n_regions,
                                                                              unsigned tiles_per_block,
                                                                              unsigned block_bytes) {
  __shared__ __attribute__((aligned(16))) unsigned char tile[kCompactTile];
  unsigned r, m, t;
  if (!compact_index<XCD>(n_moves, n_regions, tiles_per_block, r, m, t)) return;
  const char *src = static_cast<const char *>(a.base[r]) + a.src[m] * (int64_t)block_bytes + (size_t)t * kCompactTile;
  char *dst = static_cast<char *>(a.base[r]) + a.dst[m] * (int64_t)block_bytes + (size_t)t * kCompactTile;
  const unsigned remain = block_bytes - t * kCompactTile; // bytes of this tile that exist (>= 16)
  const unsigned wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const bool full = remain >= (unsigned)kCompactTile; // wave-uniform: every tile but a ragged last one
  const unsigned limit = full ? (unsigned)kCompactTile : remain;
#pragma unroll
  for (int i = 0; i < kPiecesPerWave; ++i) {
    const unsigned off = (wave * kPiecesPerWave + i) * kPiece + lane * 16;
    if (full || off < limit)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(src + off),
                                       (__attribute__((address_space(3))) void *)(tile +
                                                                                  (wave * kPiecesPerWave + i) * kPiece),
                                       16, 0, NT ? 2 : 0); // aux bit 1 = nt on gfx94x/gfx950
  }
  // The DMA writes LDS behind the compiler's back: wait for this wave's pieces by hand.
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if (full) {
    v4u v[kPiecesPerWave];
#pragma unroll
    for (int i = 0; i < kPiecesPerWave; ++i)
      v[i] = *reinterpret_cast<const v4u *>(tile + (wave * kPiecesPerWave + i) * kPiece + lane * 16);
#pragma unroll
    for (int i = 0; i < kPiecesPerWave; ++i) {
      v4u *o = reinterpret_cast<v4u *>(dst + (wave * kPiecesPerWave + i) * kPiece + lane * 16);
      if (NT)
        __builtin_nontemporal_store(v[i], o);
      else
        *o = v[i];
    }
  } else {
    for (int i = 0; i < kPiecesPerWave; ++i) {
      const unsigned off = (wave * kPiecesPerWave + i) * kPiece + lane * 16;
      if (off < limit) *reinterpret_cast<v4u *>(dst + off) = *reinterpret_cast<const v4u *>(tile + off);
    }
  }
}

// Variant 1 — register-staged: all loads of a lane in flight first, then the stores.
template <bool XCD, bool NT>
__global__ __launch_bounds__(kCompactThreads) void compact_blocks_reg_kernel(CompactArgs a, unsigned n_moves,
                                                                              unsigned n_regions,
                                                                              unsigned tiles_per_block,
                                                                              unsigned block_bytes) {
  unsigned r, m, t;
  if (!compact_index<XCD>(n_moves, n_regions, tiles_per_block, r, m, t)) return;
  const char *src = static_cast<const char *>(a.base[r]) + a.src[m] * (int64_t)block_bytes + (size_t)t * kCompactTile;
  char *dst = static_cast<char *>(a.base[r]) + a.dst[m] * (int64_t)block_bytes + (size_t)t * kCompactTile;
  const unsigned remain = block_bytes - t * kCompactTile;
  const unsigned wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  if (remain >= (unsigned)kCompactTile) { // full tile: every load in flight before the first store
    v4u v[kPiecesPerWave];
#pragma unroll
    for (int i = 0; i < kPiecesPerWave; ++i) {
      const v4u *in = reinterpret_cast<const v4u *>(src + (wave * kPiecesPerWave + i) * kPiece + lane * 16);
      v[i] = NT ? __builtin_nontemporal_load(in) : *in;
    }
#pragma unroll
    for (int i = 0; i < kPiecesPerWave; ++i) {
      v4u *o = reinterpret_cast<v4u *>(dst + (wave * kPiecesPerWave + i) * kPiece + lane * 16);
      if (NT)
        __builtin_nontemporal_store(v[i], o);
      else
        *o = v[i];
    }
  } else {
    for (int i = 0; i < kPiecesPerWave; ++i) {
      const unsigned off = (wave * kPiecesPerWave + i) * kPiece + lane * 16;
      if (off < remain) *reinterpret_cast<v4u *>(dst + off) = *reinterpret_cast<const v4u *>(src + off);
    }
  }
}

// Variants 6 (LDS-staged) and 7 (register-staged) with 32 KiB tiles: every wave has 8 KiB in flight instead of 4.
// BLOCKED (variant 9): an XCD takes a contiguous eighth of the (region, move) pairs instead of every eighth pair - neighbouring
// moves of a planner-made list share their source and destination pages, and an XCD has its own L2 / translation cache.
// ROTATED (variant 11): as BLOCKED, and XCD x starts x * (an eighth of its range + 57) pairs into its range (wrapping round):
// with every XCD walking the SAME move list through its own regions in step, the eight streams would read the same block index of
// regions that lie a power of two apart at the same moment.
template <bool LDS, bool BLOCKED = false, bool ROTATED = false>
__global__ __launch_bounds__(kCompactThreads) void compact_blocks_big_kernel(CompactArgs a, unsigned n_moves, unsigned n_regions,
                                                                              unsigned tiles_per_block, unsigned block_bytes) {
  __shared__ __attribute__((aligned(16))) unsigned char tile[LDS ? kCompactTileBig : 16];
  const unsigned x = blockIdx.x & 7u, i = blockIdx.x >> 3;
  const unsigned t = i % tiles_per_block;
  unsigned q = i / tiles_per_block;
  const unsigned per_xcd = (n_moves * n_regions + 7u) >> 3;
  if (ROTATED) q = (q + x * ((per_xcd >> 3) + 57u)) % per_xcd;
  const unsigned pair = BLOCKED ? x * per_xcd + q : q * 8u + x;
  if (pair >= n_moves * n_regions) return;
  const unsigned r = pair / n_moves, m = pair - r * n_moves;
  const char *src = static_cast<const char *>(a.base[r]) + a.src[m] * (int64_t)block_bytes + (size_t)t * kCompactTileBig;
  char *dst = static_cast<char *>(a.base[r]) + a.dst[m] * (int64_t)block_bytes + (size_t)t * kCompactTileBig;
  const unsigned remain = block_bytes - t * kCompactTileBig;
  const unsigned wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const bool full = remain >= (unsigned)kCompactTileBig;
  const unsigned limit = full ? (unsigned)kCompactTileBig : remain;
  if (LDS) {
#pragma unroll
    for (int k = 0; k < kPiecesPerWaveBig; ++k) {
      const unsigned off = (k * kCompactWaves + wave) * kPiece + lane * 16;
      if (full || off < limit)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(src + off),
                                         (__attribute__((address_space(3))) void *)(tile + (k * kCompactWaves + wave) * kPiece), 16, 0, 2);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (full) {
      v4u v[kPiecesPerWaveBig];
#pragma unroll
      for (int k = 0; k < kPiecesPerWaveBig; ++k) v[k] = *reinterpret_cast<const v4u *>(tile + (k * kCompactWaves + wave) * kPiece + lane * 16);
#pragma unroll
      for (int k = 0; k < kPiecesPerWaveBig; ++k)
        __builtin_nontemporal_store(v[k], reinterpret_cast<v4u *>(dst + (k * kCompactWaves + wave) * kPiece + lane * 16));
    } else {
      for (int k = 0; k < kPiecesPerWaveBig; ++k) {
        const unsigned off = (k * kCompactWaves + wave) * kPiece + lane * 16;
        if (off < limit) *reinterpret_cast<v4u *>(dst + off) = *reinterpret_cast<const v4u *>(tile + off);
      }
    }
  } else if (full) {
    v4u v[kPiecesPerWaveBig];
#pragma unroll
    for (int k = 0; k < kPiecesPerWaveBig; ++k)
      v[k] = __builtin_nontemporal_load(reinterpret_cast<const v4u *>(src + (k * kCompactWaves + wave) * kPiece + lane * 16));
#pragma unroll
    for (int k = 0; k < kPiecesPerWaveBig; ++k)
      __builtin_nontemporal_store(v[k], reinterpret_cast<v4u *>(dst + (k * kCompactWaves + wave) * kPiece + lane * 16));
  } else {
    for (int k = 0; k < kPiecesPerWaveBig; ++k) {
      const unsigned off = (k * kCompactWaves + wave) * kPiece + lane * 16;
      if (off < limit) *reinterpret_cast<v4u *>(dst + off) = *reinterpret_cast<const v4u *>(src + off);
    }
  }
}

hipError_t launch_compact_blocks(void *const *bases, int n_regions, const int64_t *src, const int64_t *dst, int n_moves,
                                 size_t block_bytes, hipStream_t stream, int variant) {
  if (n_regions <= 0 || n_moves <= 0) return hipSuccess;
  if (n_regions > kMaxRegionsPerLaunch || n_moves > kMaxMovesPerLaunch || block_bytes == 0 || block_bytes % 16 != 0 ||
      block_bytes > 0x7fffffffull)
    return hipErrorInvalidValue;
  CompactArgs a;
  for (int i = 0; i < n_regions; ++i) {
    if (reinterpret_cast<uintptr_t>(bases[i]) % 16 != 0) return hipErrorInvalidValue;
    a.base[i] = bases[i];
  }
  for (int i = 0; i < n_moves; ++i) {
    if (src[i] < 0 || dst[i] < 0) return hipErrorInvalidValue;
    a.src[i] = src[i];
    a.dst[i] = dst[i];
  }
  const unsigned tiles = (unsigned)((block_bytes + kCompactTile - 1) / kCompactTile);
  const size_t pairs = (size_t)n_moves * (size_t)n_regions;
  const size_t grid = (size_t)tiles * pairs, grid_xcd = (size_t)tiles * ((pairs + 7) / 8 * 8);
  if (grid_xcd > 0x7fffffffull) return hipErrorInvalidValue;
  const dim3 blk(kCompactThreads);
  const unsigned nm = (unsigned)n_moves, nr = (unsigned)n_regions, bb = (unsigned)block_bytes;
  // Non-temporal loads and stores are the default: every byte is touched exactly once, and keeping it out of the
  // way of the L2/MALL is worth +2..7 % (profiles/r01_compact_bench.jsonl; same on a plain contiguous copy,
  // tools/copy_bench.cpp: 5.9 vs 5.66 TB/s).
  // Default since round 3: LDS-staged, XCD-aware, non-temporal, 32 KiB tiles (8 KiB in flight per wave): the same rate as the
  // 16 KiB tiles on 32 KiB blocks (5.47 vs 5.50 TB/s), +2.6 % on 16 KiB blocks and +12 % on ragged 18 KiB MLA blocks (one tile per
  // block instead of a full one and a 2 KiB tail) - profiles/r03_compact_bench.jsonl. Variant 8 is the 16 KiB-tile form.
  // Since the end of round 3 the default hands every XCD a contiguous eighth of the (region, move) pairs (variant 9) instead of
  // every eighth pair (variant 6): with 64 regions an XCD then works inside 8 of them, and its L2 / translation cache sees an
  // eighth of the pages - +10..18 % on the Llama-3-8B geometry, random or planner-ordered moves alike
  // (profiles/r03_compact_bench_blocked.jsonl).
  if (variant == 0) variant = 9;
  if (variant == 6 || variant == 7 || variant == 9 || variant == 10 || variant == 11) {
    const unsigned tiles_big = (unsigned)((block_bytes + kCompactTileBig - 1) / kCompactTileBig);
    const size_t g = (size_t)tiles_big * ((pairs + 7) / 8 * 8);
    if (variant == 6)
      compact_blocks_big_kernel<true><<<dim3((unsigned)g), blk, 0, stream>>>(a, nm, nr, tiles_big, bb);
    else if (variant == 7)
      compact_blocks_big_kernel<false><<<dim3((unsigned)g), blk, 0, stream>>>(a, nm, nr, tiles_big, bb);
    else if (variant == 9)
      compact_blocks_big_kernel<true, true><<<dim3((unsigned)g), blk, 0, stream>>>(a, nm, nr, tiles_big, bb);
    else if (variant == 10)
      compact_blocks_big_kernel<false, true><<<dim3((unsigned)g), blk, 0, stream>>>(a, nm, nr, tiles_big, bb);
    else
      compact_blocks_big_kernel<true, true, true><<<dim3((unsigned)g), blk, 0, stream>>>(a, nm, nr, tiles_big, bb);
    return hipGetLastError();
  }
  switch (variant) {
  case 1: // register-staged, XCD-aware, non-temporal
    compact_blocks_reg_kernel<true, true><<<dim3((unsigned)grid_xcd), blk, 0, stream>>>(a, nm, nr, tiles, bb);
    break;
  case 2: // LDS-staged, interleaved placement, temporal (the first version; A/B runs)
    compact_blocks_lds_kernel<false, false><<<dim3((unsigned)grid), blk, 0, stream>>>(a, nm, nr, tiles, bb);
    break;
  case 3: // register-staged, interleaved, temporal
    compact_blocks_reg_kernel<false, false><<<dim3((unsigned)grid), blk, 0, stream>>>(a, nm, nr, tiles, bb);
    break;
  case 4: // LDS-staged, XCD-aware, temporal
    compact_blocks_lds_kernel<true, false><<<dim3((unsigned)grid_xcd), blk, 0, stream>>>(a, nm, nr, tiles, bb);
    break;
  case 5: // register-staged, XCD-aware, temporal
    compact_blocks_reg_kernel<true, false><<<dim3((unsigned)grid_xcd), blk, 0, stream>>>(a, nm, nr, tiles, bb);
    break;
  default: // (8) LDS-staged, XCD-aware, non-temporal, 16 KiB tiles: the default until round 3
    compact_blocks_lds_kernel<true, true><<<dim3((unsigned)grid_xcd), blk, 0, stream>>>(a, nm, nr, tiles, bb);
  }
  return hipGetLastError();
}

} // namespace kvc
