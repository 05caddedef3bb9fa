// index_kernels.hip — block ids <-> token slot indices (gfx950), the step right after
// KVCacheManager.alloc()/before KVCacheManager.free() in the SGLang token-pool allocators
// (reference glue: kvcached/integration/sglang/patches.py:186-276; it builds the index tensors with
// torch.tensor(list) + broadcasting arithmetic and SGLang's Triton alloc_extend/alloc_decode kernels,
// and turns freed token indices back into block ids with torch.unique(...).cpu()).
//
// Integer work, launch-latency bound (a 16k-token prefill writes 128 KiB). What is done for it:
//   * the block ids KVCacheManager.alloc() returned (a HOST list) ride in the kernarg segment
//     (<=1024 ids = 8 KiB): no host->device copy, no staging buffer, nothing to synchronise;
//     per-lane indexing reads the kernarg segment as ordinary constant memory (a by-value array
//     indexed with a VGPR would be spilled to scratch by the compiler);
//   * one launch per call, 64-lane coalesced 8-byte stores (512 B per wave-instruction);
//   * the prefix sums over the batch (where request i's output and new pages start) are recomputed
//     per workgroup from L2-resident length arrays: O(bs^2/256) loads, nothing for bs in the hundreds;
//   * free(): a bitmap + single-workgroup ordered sweep replaces the sort behind torch.unique.

#include "kernels.hpp"

namespace kvc {

struct IdTable {
  int64_t id[kMaxIdsPerLaunch];
};
static_assert(sizeof(IdTable) == 8192, "kernarg segment budget");

// ids live in the kernarg segment at byte 0 (IdTable is the first kernel parameter) unless a device
// array is given.
__device__ __forceinline__ const int64_t *id_array(const int64_t *ids_dev) {
  if (ids_dev) return ids_dev;
  return (const int64_t *)(const void *)__builtin_amdgcn_kernarg_segment_ptr();
}

__device__ __forceinline__ int64_t wave_sum(int64_t v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
  return v;
}

// Sum of (a, b) over the 256 threads of the workgroup, result valid in every thread.
__device__ __forceinline__ void block_sum2(int64_t &a, int64_t &b, int64_t *lds /* 8 entries */) {
  a = wave_sum(a);
  b = wave_sum(b);
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  if (lane == 0) {
    lds[wave] = a;
    lds[4 + wave] = b;
  }
  __syncthreads();
  a = lds[0] + lds[1] + lds[2] + lds[3];
  b = lds[4] + lds[5] + lds[6] + lds[7];
  __syncthreads();
}

__device__ __forceinline__ int64_t cdiv(int64_t a, int64_t b) { return (a + b - 1) / b; }

// ---------------------------------------------------------------------------- expand_block_ids
// out[i*tpb + j] = ids[i]*tpb + j        (patches.py:192-196: page_ids[:, None] * page_size + arange)
__global__ __launch_bounds__(256) void expand_block_ids_kernel([[maybe_unused]] IdTable tbl, const int64_t *ids_dev, size_t n_ids,
                                                                unsigned tpb, int64_t *out) {
  const int64_t *ids = id_array(ids_dev);
  const size_t t = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (t >= n_ids * tpb) return;
  const size_t i = t / tpb;
  out[t] = ids[i] * (int64_t)tpb + (int64_t)(t - i * tpb);
}

// ---------------------------------------------------------------------------- alloc_extend_indices
// Request r extends its sequence from pre_len to seq_len tokens. The tokens that still fit into its
// last, partially filled block continue after last_loc; the rest go into this request's share of the
// new blocks, in order. Requests are laid out back to back in `out`; new blocks are consumed in
// request order. (Same result as SGLang's alloc_extend_kernel, restated position by position.)
static constexpr int kExtendTokensPerGroup = 4096;

__global__ __launch_bounds__(256) void alloc_extend_kernel([[maybe_unused]] IdTable tbl, const int64_t *ids_dev, size_t n_ids,
                                                            const int64_t *__restrict__ pre_lens,
                                                            const int64_t *__restrict__ seq_lens,
                                                            const int64_t *__restrict__ last_loc, int64_t tpb,
                                                            int64_t *__restrict__ out, int64_t out_len) {
  __shared__ int64_t lds[8];
  const int64_t *ids = id_array(ids_dev);
  const unsigned r = blockIdx.x;
  const int64_t pre = pre_lens[r], seq = seq_lens[r];
  const int64_t extend = seq - pre;
  const int64_t t0 = (int64_t)blockIdx.y * kExtendTokensPerGroup;
  if (t0 >= extend) return; // uniform for the workgroup
  int64_t out_start = 0, page_start = 0;
  for (unsigned j = threadIdx.x; j < r; j += 256) {
    const int64_t s = seq_lens[j], p = pre_lens[j];
    out_start += s - p;
    page_start += cdiv(s, tpb) - cdiv(p, tpb);
  }
  block_sum2(out_start, page_start, lds);
  const int64_t first_new = cdiv(pre, tpb); // index (within the sequence) of the first new block
  const int64_t loc = last_loc[r];
  const int64_t t1 = min(extend, t0 + kExtendTokensPerGroup);
  for (int64_t t = t0 + threadIdx.x; t < t1; t += 256) {
    const int64_t pos = pre + t, blk = pos / tpb;
    int64_t v;
    if (blk < first_new) {
      v = loc + 1 + t;
    } else {
      const int64_t k = page_start + (blk - first_new);
      if ((size_t)k >= n_ids) continue; // inconsistent inputs: never read past the id table
      v = ids[k] * tpb + (pos - blk * tpb);
    }
    if (out_start + t < out_len) out[out_start + t] = v;
  }
}

// ---------------------------------------------------------------------------- alloc_decode_indices
// One new token per request (seq_len already counts it): it opens a new block iff (seq_len-1) is a
// multiple of tpb, else it follows last_loc. New blocks are consumed in request order.
__global__ __launch_bounds__(256) void alloc_decode_kernel([[maybe_unused]] IdTable tbl, const int64_t *ids_dev, size_t n_ids,
                                                            const int64_t *__restrict__ seq_lens,
                                                            const int64_t *__restrict__ last_loc, unsigned bs, int64_t tpb,
                                                            int64_t *__restrict__ out) {
  __shared__ int64_t lds[8];
  __shared__ unsigned wave_cnt[4];
  const int64_t *ids = id_array(ids_dev);
  const unsigned base = blockIdx.x * 256, r = base + threadIdx.x;
  int64_t before = 0, unused = 0;
  for (unsigned j = threadIdx.x; j < base; j += 256) before += ((seq_lens[j] - 1) % tpb == 0) ? 1 : 0;
  block_sum2(before, unused, lds);
  const bool need = r < bs && ((seq_lens[r] - 1) % tpb == 0);
  const unsigned long long mask = __ballot(need);
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  if (lane == 0) wave_cnt[wave] = (unsigned)__popcll(mask);
  __syncthreads();
  unsigned off = (unsigned)__popcll(mask & ((1ull << lane) - 1ull));
  for (int w = 0; w < wave; ++w) off += wave_cnt[w];
  if (r >= bs) return;
  if (need) {
    const size_t k = (size_t)before + off;
    if (k < n_ids) out[r] = ids[k] * tpb;
  } else {
    out[r] = last_loc[r] + 1;
  }
}

// ---------------------------------------------------------------------------- unique_block_ids
// free(): token indices -> the sorted set of blocks they fall into (torch.unique(idx // tpb)).
// Pass 1 sets one bit per touched block and tracks the touched word range; pass 2 (one workgroup)
// sweeps that range in ascending order, emits the set bits and clears them again.
struct UniqueHeader {
  unsigned min_word, max_word; // touched range (min starts at 0xffffffff, max at 0)
  unsigned bad;                // indices outside [0, n_blocks*tpb)
  unsigned pad;
};

__global__ __launch_bounds__(256) void mark_blocks_kernel(const int64_t *__restrict__ idx, size_t n, int64_t tpb,
                                                           int64_t n_blocks, unsigned *__restrict__ bitmap,
                                                           UniqueHeader *hdr) {
  const size_t t = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (t >= n) return;
  const int64_t v = idx[t];
  if (v < 0 || v >= n_blocks * tpb) {
    atomicAdd(&hdr->bad, 1u);
    return;
  }
  const unsigned b = (unsigned)(v / tpb), w = b >> 5;
  const unsigned bit = 1u << (b & 31u);
  if (!(__hip_atomic_load(&bitmap[w], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & bit)) {
    atomicOr(&bitmap[w], bit);
    atomicMin(&hdr->min_word, w);
    atomicMax(&hdr->max_word, w);
  }
}

// result[0] = count, result[1..] = ascending block ids. One workgroup of 1024 threads.
__global__ __launch_bounds__(1024) void sweep_blocks_kernel(unsigned *__restrict__ bitmap, UniqueHeader *hdr,
                                                             int64_t *__restrict__ result, size_t cap) {
  __shared__ unsigned wave_cnt[16];
  __shared__ unsigned chunk_total;
  const unsigned lo = hdr->min_word, hi = hdr->max_word;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  size_t emitted = 0;
  if (lo <= hi) {
    for (unsigned w0 = lo; w0 <= hi; w0 += 1024) {
      const unsigned w = w0 + threadIdx.x;
      unsigned bits = (w <= hi) ? bitmap[w] : 0u;
      if (bits) bitmap[w] = 0u;
      const unsigned c = (unsigned)__popc(bits);
      // exclusive scan of c over the workgroup
      unsigned incl = c;
#pragma unroll
      for (int off = 1; off < 64; off <<= 1) {
        const unsigned up = __shfl_up(incl, off, 64);
        if (lane >= off) incl += up;
      }
      if (lane == 63) wave_cnt[wave] = incl;
      __syncthreads();
      unsigned pre = incl - c;
      for (int k = 0; k < wave; ++k) pre += wave_cnt[k];
      if (threadIdx.x == 1023) chunk_total = pre + c;
      __syncthreads();
      size_t o = emitted + pre;
      while (bits) {
        const unsigned b = (unsigned)__ffs(bits) - 1u;
        bits &= bits - 1u;
        if (o < cap) result[1 + o] = (int64_t)w * 32 + b;
        ++o;
      }
      emitted += chunk_total;
      __syncthreads();
    }
  }
  if (threadIdx.x == 0) {
    result[0] = hdr->bad ? -(int64_t)hdr->bad : (int64_t)emitted;
    hdr->min_word = 0xffffffffu;
    hdr->max_word = 0u;
    hdr->bad = 0u;
  }
}

// ---------------------------------------------------------------------------- launchers
static void load_table(IdTable &t, const int64_t *ids_host, size_t n) {
  for (size_t i = 0; i < n; ++i) t.id[i] = ids_host[i];
}

hipError_t launch_expand_block_ids(const int64_t *ids_host, const int64_t *ids_dev, size_t n_ids, int64_t tpb,
                                   int64_t *out, hipStream_t stream) {
  if (n_ids == 0) return hipSuccess;
  if (tpb <= 0 || tpb > 0x7fffffff || (!ids_dev && (n_ids > (size_t)kMaxIdsPerLaunch || !ids_host))) return hipErrorInvalidValue;
  IdTable t;
  if (!ids_dev) load_table(t, ids_host, n_ids);
  const size_t total = n_ids * (size_t)tpb, grid = (total + 255) / 256;
  if (grid > 0x7fffffffull) return hipErrorInvalidValue;
  expand_block_ids_kernel<<<dim3((unsigned)grid), dim3(256), 0, stream>>>(t, ids_dev, n_ids, (unsigned)tpb, out);
  return hipGetLastError();
}

hipError_t launch_alloc_extend(const int64_t *ids_host, const int64_t *ids_dev, size_t n_ids, const int64_t *pre_lens,
                               const int64_t *seq_lens, const int64_t *last_loc, size_t bs, int64_t tpb, int64_t *out,
                               size_t out_len, hipStream_t stream) {
  if (bs == 0 || out_len == 0) return hipSuccess;
  if (tpb <= 0 || bs > 0x7fffffffull || (!ids_dev && n_ids > (size_t)kMaxIdsPerLaunch) || (n_ids && !ids_dev && !ids_host))
    return hipErrorInvalidValue;
  IdTable t;
  if (!ids_dev) load_table(t, ids_host, n_ids);
  const size_t groups = (out_len + kExtendTokensPerGroup - 1) / kExtendTokensPerGroup;
  if (groups > 65535) return hipErrorInvalidValue;
  alloc_extend_kernel<<<dim3((unsigned)bs, (unsigned)groups), dim3(256), 0, stream>>>(t, ids_dev, n_ids, pre_lens, seq_lens,
                                                                                        last_loc, tpb, out, (int64_t)out_len);
  return hipGetLastError();
}

hipError_t launch_alloc_decode(const int64_t *ids_host, const int64_t *ids_dev, size_t n_ids, const int64_t *seq_lens,
                               const int64_t *last_loc, size_t bs, int64_t tpb, int64_t *out, hipStream_t stream) {
  if (bs == 0) return hipSuccess;
  if (tpb <= 0 || bs > 0x7fffffffull || (!ids_dev && n_ids > (size_t)kMaxIdsPerLaunch) || (n_ids && !ids_dev && !ids_host))
    return hipErrorInvalidValue;
  IdTable t;
  if (!ids_dev) load_table(t, ids_host, n_ids);
  alloc_decode_kernel<<<dim3((unsigned)((bs + 255) / 256)), dim3(256), 0, stream>>>(t, ids_dev, n_ids, seq_lens, last_loc,
                                                                                     (unsigned)bs, tpb, out);
  return hipGetLastError();
}

hipError_t launch_unique_block_ids(const int64_t *idx, size_t n, int64_t tpb, int64_t n_blocks, unsigned *bitmap,
                                   void *header, int64_t *result, size_t cap, hipStream_t stream) {
  if (tpb <= 0 || n_blocks <= 0 || n_blocks > 0x7fffffffll * 32) return hipErrorInvalidValue;
  const size_t grid = (n + 255) / 256;
  if (grid > 0x7fffffffull) return hipErrorInvalidValue;
  if (n) mark_blocks_kernel<<<dim3((unsigned)grid), dim3(256), 0, stream>>>(idx, n, tpb, n_blocks, bitmap, (UniqueHeader *)header);
  sweep_blocks_kernel<<<dim3(1), dim3(1024), 0, stream>>>(bitmap, (UniqueHeader *)header, result, cap);
  return hipGetLastError();
}

} // namespace kvc
