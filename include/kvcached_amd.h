/* kvcached_amd.h — C ABI of libkvcached_amd.so, the MI355X-native elastic KV-cache VMM.
 *
 * This is the drop-in boundary for the reference's hot path. The reference exposes the
 * path as a pybind11 module `kvcached.vmm_ops` (csrc/torch_bindings.cpp:182-258); every
 * entry point below is what that module's functions/methods bind, one C symbol per
 * Python-visible function, with plain pointers and sizes only (no torch, no C++ types).
 * kvcached_amd/csrc/vmm_ops.cpp is the pybind11 layer that re-creates the exact
 * `vmm_ops` Python surface on top of these symbols; INTEGRATION.md shows the stub.
 *
 * Conventions
 *   - return 0 / non-negative on success, a negative KVC_E_* code on failure;
 *     kvc_last_error() returns the message for the calling thread.
 *   - all functions are thread-safe unless noted; none requires the Python GIL.
 *   - offsets and sizes are bytes; ids are int64_t like the reference's page_id_t.
 *   - device strings: "cuda", "cuda:N", "hip:N" (PyTorch-ROCm spelling) or "cpu".
 *     "cpu" is the reference's own host device (csrc/ftensor.cpp:40-44, csrc/page.cpp:28-37):
 *     anonymous mmap, map/unmap are bookkeeping only. It is never selected implicitly and
 *     the HIP kernels refuse it (KVC_E_NO_GPU).
 */
#ifndef KVCACHED_AMD_H
#define KVCACHED_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define KVC_ABI_VERSION 5

enum {
  KVC_OK = 0,
  KVC_E_INVALID = -1,  /* bad argument / bad state (reference: LOGGER(ERROR)+false or abort) */
  KVC_E_GPU = -2,      /* a HIP runtime call failed (reference: CHECK_GPU -> abort, csrc/inc/gpu_vmm.hpp:37-45) */
  KVC_E_NO_PAGES = -3, /* "No free pages left" (csrc/page_allocator.cpp:201) */
  KVC_E_RUNTIME = -4,  /* any other std::runtime_error of the reference */
  KVC_E_NO_GPU = -5,   /* a GPU-only entry point was called on the "cpu" device / without a GPU */
  KVC_E_CALLBACK = -6, /* a broadcast callback reported failure */
  KVC_E_NOT_CREATED = -7 /* map/unmap before create_kv_tensors: the reference logs and returns false (allocator.cpp:163-166) */
};

const char *kvc_last_error(void);
int kvc_abi_version(void);

/* ------------------------------------------------------------------ allocator lifecycle
 * replaces kvcached.vmm_ops.{init_kvcached, shutdown_kvcached, create_kv_tensors,
 * kv_tensors_created, map_to_kv_tensors, unmap_from_kv_tensors}
 * (csrc/torch_bindings.cpp:20-58,185-198 -> csrc/allocator.cpp:72-257). */

/* page_size 0 = keep default 2 MiB; must be a multiple of 2 MiB (allocator.cpp:81-90). */
int kvc_init(const char *dev_str, size_t page_size, int contiguous_layout);
int kvc_shutdown(void);

/* Reserves VA and (in compat mode) backfills it with the shared zero page.
 * On return out_ptrs[i]/out_nbytes[i] describe tensor i (num_layers tensors, or ONE in
 * contiguous layout); *inout_count is capacity on entry, tensor count on exit.
 * dtype_size must be 1/2/4/8 (csrc/inc/impl/torch_utils.ipp:32-46). */
int kvc_create_kv_tensors(size_t size, size_t dtype_size, const char *dev_str, int64_t num_layers,
                          int64_t num_kv_buffers, int64_t group_id, int unified_pool, void **out_ptrs,
                          size_t *out_nbytes, int64_t *inout_count);
int kvc_kv_tensors_created(int64_t group_id); /* 1 / 0 / <0 */
/* Device chosen by kvc_init: *is_gpu 0 for "cpu"; *index is the resolved HIP device ordinal. */
int kvc_get_device(int *is_gpu, int *index);

/* The batched hot path. One call backs every (layer x K/V) slot of every offset:
 * physical handles come from the per-device pool, hipMemMap + ranged hipMemSetAccess,
 * then the zero_fill_pages kernel on the allocator's stream, overlapped with the
 * remaining driver calls; returns after the fill has completed. */
int kvc_map_to_kv_tensors(const int64_t *offsets, size_t n, int64_t group_id);
int kvc_unmap_from_kv_tensors(const int64_t *offsets, size_t n, int64_t group_id);

/* Runtime knobs (also read once from the environment at kvc_init):
 *   KVC_OPT_ZERO_BACKFILL  1 (default, "compat", as the reference does: csrc/ftensor.cpp:160-176) = unbacked VA reads as
 *                              zeros from the moment an unmap returns: with the drm backend on gfx950 it is a PRT
 *                              mapping (reads 0, writes dropped, no fault, no zero page: DESIGN.md §4.2;
 *                              kvc_get_option(128)), elsewhere or with KVCACHED_PRT=false it aliases zero pages (one
 *                              zero extent on drm, sharded pages through ROCr otherwise); map and unmap calls both
 *                              invalidate the GPU TLBs before they return (a PRT entry or a zero alias that something
 *                              has looked at is cached). 0 ("lazy", KVCACHED_ZERO_BACKFILL=false) = unbacked VA is
 *                              plain unmapped - a stray access FAULTS -, a map call needs no invalidation and the one
 *                              an unmap owes runs behind the call on a thread of the library (KVCACHED_PRT=true puts
 *                              PRT behind a lazy region too: harmless stray accesses, invalidating map calls).
 *   KVC_OPT_ZERO_FILL      1 = zero freshly backed pages on the GPU (default), 0 = skip.
 *   KVC_OPT_POOL_BYTES     max bytes of idle physical handles kept for reuse.
 *   KVC_OPT_PROFILE        1 = time every kernel launch with HIP events (bench.py).
 *   KVC_OPT_TLB_SHOOTDOWN  1 (default) = force the driver to invalidate the GPU TLBs after every batch of unmaps, and
 *                              after a batch of maps when a translation of the affected VA or pages can still be
 *                              cached (zero alias or PRT entry replaced, deferred invalidation outstanding; a
 *                              translation of an UNMAPPED address is never cached on GFX9+, so backing one needs none;
 *                              KVCACHED_MAP_SHOOTDOWN=always invalidates after every map batch regardless). On
 *                              ROCm 7.2 / MI355X the VMM calls alone leave stale translations behind
 *                              (DESIGN.md §4.3); 0 only for measurements. The invalidation an unmap batch owes is
 *                              performed at once by a thread of the library (KVCACHED_ASYNC_SHOOTDOWN, default true;
 *                              compat mode and imported pages invalidate inside the call); kvc_flush_unmaps()
 *                              returns when nothing is owed or in flight; kvc_get_option(111) counts the
 *                              invalidations done off the callers' threads.
 *   KVC_OPT_DEFER_UNMAP_SHOOTDOWN 1 = the invalidation owed by an unmap batch whose pages all go back
 *                              to the library's own handle pool waits for the next map batch (which invalidates
 *                              before it touches anything) or for the moment handles are given back to the
 *                              driver, whichever comes first; 0 (default) = invalidate inside every unmap call
 *                              (the cost only moves from free() to the next alloc()).
 *                              Compat mode (ZERO_BACKFILL) and imported pages always invalidate at once.
 *   (environment only) KVCACHED_VMM_BACKEND = hybrid | drm | hip | hsa: which API backs the slots. hybrid
 *                              registers each slot with HIP once and then maps/unmaps it through ROCr's
 *                              hsa_amd_vmem_* (hipMemUnmap spins ~10 us on a GPU marker per mapping; ROCr's unmap
 *                              takes 3 us) - every hipMemcpy flavour keeps working; checked by a self test at
 *                              kvc_init, which falls back to hip. kvc_get_option(108) reports the backend in
 *                              effect (0 hip, 1 hsa, 2 hybrid, 3 drm). drm = hybrid, and this process's own
 *                              pages are mapped/unmapped with one DRM_AMDGPU_GEM_VA ioctl each through
 *                              libdrm_amdgpu on buffer objects imported once per handle (2.2 + 2.1 us per page
 *                              instead of 5.3 + 2.8); self-tested at kvc_init, falls back to hybrid. With drm, physical
 *                              pages are also allocated straight from KFD (AMDKFD_IOC_ALLOC_MEMORY_OF_GPU on the
 *                              library's own fd of /dev/kfd: 5 us flat instead of the runtime's O(live handles);
 *                              KVCACHED_DRM_KFD_CREATE=false keeps ROCr's creation; kvc_get_option(110) = 1 when active).
 *                              KVCACHED_PHYS_CHUNK_PAGES=k (default 1) allocates that memory in chunks of k pages and
 *                              maps runs of adjacent slots with one ioctl (opt-in).
 *                              DESIGN.md §4.6/§4.7.
 *   KVC_OPT_UNMAP_INVALIDATION_US  compat regions only (env KVCACHED_UNMAP_INVALIDATION_US). 0 (default): kvc_unmap_from_kv_tensors
 *                          invalidates the GPU TLBs before it returns - a freed address reads as zeros from that moment on.
 *                          T > 0: the invalidation may trail the call by at most T microseconds; it is performed by the
 *                          library's own thread, or absorbed by the invalidation of the next map batch if that comes
 *                          first (one invalidation per free+alloc cycle instead of two). Until it has happened the freed
 *                          pages are neither zeroed nor on offer (order: page tables, invalidation, zero fill, pool), so
 *                          no stale translation can ever reach a page somebody else holds; what is relaxed is only that a
 *                          read of a FREED address may still see the old contents for up to T microseconds. Still
 *                          stricter than the reference, which never invalidates (csrc/ftensor.cpp:120-140).
 *   KVC_OPT_ASYNC_UNMAP    1 = kvc_unmap_from_kv_tensors only marks the slots and queues them; a reclaimer thread
 *                              of the library carries out hipMemUnmap + invalidation + handle recycling in small
 *                              chunks, yielding to map calls. A slot that is mapped again before its turn is kept
 *                              as it is (no driver call, zero-filled again). The caller's free() path drops from
 *                              ~15 us per slot to a queue push; queued bytes count as free in kvc_mem_get_info.
 *                              kvc_flush_unmaps() waits for the queue (trim/resize/shutdown do it themselves).
 *                              0 (default) = synchronous, like the reference. Ignored in compat mode.
 * Read-only diagnostics through kvc_get_option (numbers >= 100; kvcached_amd/capi.py names them): 108 backend in effect,
 * 110 pages straight from KFD, 111 invalidations done off the callers' threads, 112-117 page creation / release time
 * split, 118 the KFD ioctl pair is the invalidation in use, 119 pages per extent at most, 120-124 the extent pool's
 * footprint, 125/126 pages zeroed on their way back / handed out without a fill of their own, 127 pages of the zero
 * extent, 128 PRT behind unbacked VA, 129 lanes per buffer (page ids backed as units), 130-153 host nanoseconds of the map /
 * unmap calls by segment and ioctl counts (bench.py), 163/164 peer pages imported straight into KFD + DRM / through the runtime. */
enum { KVC_OPT_ZERO_BACKFILL = 1, KVC_OPT_ZERO_FILL = 2, KVC_OPT_POOL_BYTES = 3, KVC_OPT_PROFILE = 4,
       KVC_OPT_TLB_SHOOTDOWN = 5, KVC_OPT_DEFER_UNMAP_SHOOTDOWN = 6, KVC_OPT_ASYNC_UNMAP = 7,
       KVC_OPT_UNMAP_INVALIDATION_US = 8 };
int kvc_set_option(int opt, int64_t value);
int64_t kvc_get_option(int opt);
int kvc_flush_unmaps(void); /* wait until every queued (async) unmap has been carried out */
/* Hold every page-table update of this process's KV regions (map / unmap calls of any thread, the prealloc thread, the
 * reclaimer) between _begin and _end, from the thread that called _begin. For code that reads or copies WHOLE KV tensors,
 * unbacked parts included (a checkpoint, a debugger's dump, the soak's sweeps): "unbacked VA reads as zeros, never faults"
 * holds for slots at rest, but inside the one ioctl that backs a slot or gives it up the kernel clears the entries before
 * it writes the new ones, and an access that lands on exactly that slot in that window (~2 us) is a GPU fault - there is no
 * sequence of DRM operations without it (tools/engine_ioctl_probe.cpp, part H). The engine's own accesses never meet it:
 * nothing touches a slot that is being backed (not handed out yet) or given up (already freed). No reference counterpart
 * (its FTensor::map unmaps the zero alias and maps the page in two separate calls: csrc/ftensor.cpp:100-118). */
int kvc_quiesce_begin(void);
int kvc_quiesce_end(void);

/* Counters since kvc_init / last reset. */
typedef struct kvc_stats {
  int64_t pages_mapped, pages_unmapped;       /* physical slots */
  int64_t handles_created, handles_released, handles_reused;
  int64_t map_calls, unmap_calls;             /* kvc_(un)map_to_kv_tensors invocations */
  int64_t map_ns, unmap_ns;                   /* host wall time inside them */
  int64_t fill_launches, fill_bytes;          /* zero_fill_pages */
  double fill_ms;                             /* sum of event-timed kernel durations (profile on) */
  int64_t compact_launches, compact_bytes;    /* compact_blocks: bytes read + written */
  double compact_ms;
  int64_t tlb_shootdowns;                     /* explicit GPU TLB invalidations (see KVC_OPT_TLB_SHOOTDOWN) */
  int64_t shootdown_ns;                       /* host wall time spent in them */
  int64_t index_launches;                     /* block id <-> token index kernels (kvc_expand_block_ids ...) */
  int64_t unmaps_queued, unmaps_cancelled;    /* async unmap: slots queued / mapped again before the reclaimer's turn */
} kvc_stats_t;
int kvc_get_stats(kvc_stats_t *out);
int kvc_reset_stats(void);
/* Diagnostics: host nanoseconds spent inside each driver call class since the last reset, out[8] =
 * {unmap zero alias, handle acquire (pool / hipMemCreate), hipMemMap, hipMemSetAccess, hipMemUnmap,
 *  handle release, re-alias, wait for the fill kernel}. */
int kvc_get_driver_breakdown(int64_t *out8);

/* hipMemGetInfo of the allocator's device; kvc_set_mem_info_override(free,total) replaces
 * the reading (tests and the "cpu" device), (0,0) removes the override. */
int kvc_mem_get_info(size_t *free_bytes, size_t *total_bytes);
int kvc_set_mem_info_override(size_t free_bytes, size_t total_bytes);

/* ------------------------------------------------------------------ InternalPage
 * replaces vmm_ops.InternalPage (torch_bindings.cpp:242-257 -> page_allocator.cpp:40-100). */
typedef struct kvc_page kvc_page_t;
kvc_page_t *kvc_page_new(int64_t page_id, int64_t page_size);
void kvc_page_delete(kvc_page_t *p);
int64_t kvc_page_id(const kvc_page_t *p);
int64_t kvc_page_size(const kvc_page_t *p);
void kvc_page_init(kvc_page_t *p, int64_t block_mem_size);
int64_t kvc_page_alloc(kvc_page_t *p, int64_t num_blocks, int64_t *out); /* count, or KVC_E_RUNTIME "Not enough free blocks in page" */
void kvc_page_free(kvc_page_t *p, int64_t block_id);
void kvc_page_free_batch(kvc_page_t *p, const int64_t *block_ids, size_t n);
int kvc_page_empty(const kvc_page_t *p);
int kvc_page_full(const kvc_page_t *p);
int64_t kvc_page_num_free_blocks(const kvc_page_t *p);
int64_t kvc_page_get_free_blocks(const kvc_page_t *p, int64_t *out, int64_t cap); /* returns count; copies if cap suffices */
void kvc_page_get_block_range(int64_t page_id, int64_t page_size, int64_t block_mem_size, int64_t *start, int64_t *end);
int64_t kvc_page_get_num_blocks(int64_t page_size, int64_t block_mem_size);

/* ------------------------------------------------------------------ PageAllocator
 * replaces vmm_ops.PageAllocator (torch_bindings.cpp:201-239 -> page_allocator.cpp:103-782). */
typedef struct kvc_page_allocator kvc_page_allocator_t;
/* returns nonzero to report failure (raised as "Failed to map page N: ..." by alloc_page) */
typedef int (*kvc_broadcast_cb)(void *user, int64_t world_size, const int64_t *offsets, size_t n);
typedef int (*kvc_bool_cb)(void *user);

kvc_page_allocator_t *kvc_pa_new(int64_t num_layers, int64_t mem_size_per_layer, int64_t page_size,
                                 int64_t world_size, int64_t pp_rank, int async_sched, int contiguous_layout,
                                 int enable_page_prealloc, int64_t num_kv_buffers, int64_t group_id,
                                 const char *ipc_name);
void kvc_pa_delete(kvc_page_allocator_t *pa);
int kvc_pa_start_prealloc_thread(kvc_page_allocator_t *pa);
int kvc_pa_stop_prealloc_thread(kvc_page_allocator_t *pa);
int64_t kvc_pa_alloc_page(kvc_page_allocator_t *pa); /* page id, or KVC_E_* */
/* Addition: the ids n alloc_page() calls would return, with ONE map call for those that need backing;
 * all-or-nothing. Returns n (ids in out_ids[n]) or KVC_E_*. */
int64_t kvc_pa_alloc_pages(kvc_page_allocator_t *pa, int64_t n, int64_t *out_ids);
int kvc_pa_free_page(kvc_page_allocator_t *pa, int64_t page_id);
int kvc_pa_free_pages(kvc_page_allocator_t *pa, const int64_t *page_ids, size_t n);
int kvc_pa_resize(kvc_page_allocator_t *pa, int64_t new_mem_size); /* 1 / 0 / <0 */
int kvc_pa_trim(kvc_page_allocator_t *pa);
int kvc_pa_reset_free_page_order(kvc_page_allocator_t *pa);
int64_t kvc_pa_get_num_free_pages(const kvc_page_allocator_t *pa);
int64_t kvc_pa_get_num_inuse_pages(const kvc_page_allocator_t *pa);
int64_t kvc_pa_get_num_total_pages(const kvc_page_allocator_t *pa);
int64_t kvc_pa_get_num_reserved_pages(const kvc_page_allocator_t *pa);
int64_t kvc_pa_get_avail_physical_pages(const kvc_page_allocator_t *pa);
int64_t kvc_pa_check_and_get_resize_target(const kvc_page_allocator_t *pa, int64_t current_mem_size);
int64_t kvc_pa_get_resize_target(const kvc_page_allocator_t *pa);
int64_t kvc_pa_get_page_id(const kvc_page_allocator_t *pa, int64_t block_id, int64_t block_mem_size);
/* group_indices_by_page, flattened in the ITERATION ORDER of the reference's
 * std::unordered_map (page_allocator.cpp:471-498): keys[k], counts[k], values concatenated.
 * keys/counts/values need room for n entries. Returns the number of groups. */
int64_t kvc_pa_group_indices_by_page(const kvc_page_allocator_t *pa, const int64_t *indices, size_t n,
                                     int64_t block_mem_size, int64_t *keys, int64_t *counts, int64_t *values);
int kvc_pa_set_broadcast_map_callback(kvc_page_allocator_t *pa, kvc_broadcast_cb cb, void *user);
int kvc_pa_set_broadcast_unmap_callback(kvc_page_allocator_t *pa, kvc_broadcast_cb cb, void *user);
int kvc_pa_set_should_use_worker_ipc_callback(kvc_page_allocator_t *pa, kvc_bool_cb cb, void *user);
/* Introspection used by tests and the compaction planner: which = 0 free, 1 reserved, 2 reclaimed. */
int64_t kvc_pa_get_page_list(const kvc_page_allocator_t *pa, int which, int64_t *out, int64_t cap);
const char *kvc_pa_ipc_name(const kvc_page_allocator_t *pa);

/* ------------------------------------------------------------------ HIP kernels (north-star additions;
 * no reference symbol — SURVEY.md §8 row a-N). `stream` is a hipStream_t or NULL for the
 * library's own stream; launches are asynchronous unless `sync` is nonzero. */

/* zero_fill_pages: each of the n page base pointers (device VAs, host array) gets page_bytes
 * of zeros. page_bytes must be a multiple of 64 KiB; pointers 64 KiB-aligned. */
int kvc_zero_fill_pages(void *const *page_ptrs, size_t n, size_t page_bytes, void *stream, int sync);

/* compact_blocks: for every region base (one per layer x K/V buffer; device VAs, host array) and
 * every move m: copy block_bytes from base + src_block[m]*block_bytes to base + dst_block[m]*block_bytes.
 * block_bytes must be a multiple of 16; src and dst block sets must be disjoint. */
int kvc_compact_blocks(void *const *region_bases, size_t n_regions, const int64_t *src_blocks,
                       const int64_t *dst_blocks, size_t n_moves, size_t block_bytes, void *stream, int sync);

/* Region bases of a group's KV tensors in map order (layer-major, K then V), for compact_blocks.
 * Returns the count; copies if cap suffices. */
int64_t kvc_get_region_bases(int64_t group_id, void **out, int64_t cap);

/* ------------------------------------------------------------------ block ids <-> token slot indices
 * The step right after KVCacheManager.alloc() / before KVCacheManager.free() in the SGLang token-pool
 * allocators; replaces the torch/Triton glue of the reference's ElasticTokenToKVPoolAllocator and
 * ElasticPagedTokenToKVPoolAllocator (kvcached/integration/sglang/patches.py:100-117,186-276).
 * `block_ids` / `new_block_ids` are HOST arrays (what alloc() returned); every other pointer is device
 * memory (int64). Asynchronous on `stream` unless stated otherwise; NULL = the device's default (null) stream, i.e. what
 * torch.cuda.current_stream() is unless the caller switched streams - these kernels consume and produce the caller's tensors.
 *
 * kvc_expand_block_ids      out[i*tpb + j] = block_ids[i]*tpb + j             (patches.py:188-197; tpb = 1: :100-102)
 * kvc_alloc_extend_indices  request r grows from prefix_lens[r] to seq_lens[r] tokens: tokens that still fit its
 *                           last block continue after last_loc[r], the others fill its share of new_block_ids in
 *                           order; requests back to back in out[extend_num_tokens]   (patches.py:199-243 +
 *                           SGLang's alloc_extend_kernel)
 * kvc_alloc_decode_indices  one new token per request: a new block iff (seq_lens[r]-1) % tpb == 0, else
 *                           last_loc[r]+1; out[bs]                              (patches.py:245-276 + alloc_decode_kernel)
 * kvc_unique_block_ids      BLOCKING. Distinct blocks of n token indices, ascending = torch.unique(idx // tpb)
 *                           (patches.py:278-288), into the host array out_host[cap]; returns the count (copied only
 *                           if it fits) or KVC_E_INVALID if an index is outside [0, num_blocks*tpb). */
int kvc_expand_block_ids(const int64_t *block_ids, size_t n, int64_t tokens_per_block, int64_t *out_dev, void *stream);
int kvc_alloc_extend_indices(const int64_t *prefix_lens_dev, const int64_t *seq_lens_dev, const int64_t *last_loc_dev,
                             size_t bs, const int64_t *new_block_ids, size_t n_new, int64_t tokens_per_block,
                             int64_t *out_dev, size_t extend_num_tokens, void *stream);
int kvc_alloc_decode_indices(const int64_t *seq_lens_dev, const int64_t *last_loc_dev, size_t bs,
                             const int64_t *new_block_ids, size_t n_new, int64_t tokens_per_block, int64_t *out_dev,
                             void *stream);
int64_t kvc_unique_block_ids(const int64_t *token_indices_dev, size_t n, int64_t tokens_per_block, int64_t num_blocks,
                             int64_t *out_host, size_t cap, void *stream);

/* ------------------------------------------------------------------ TP shared pool (north-star addition;
 * the reference has no memory sharing — kvcached/tp_ipc_util.py only broadcasts offsets).
 * Rank 0 backs slots with exportable handles and exports one POSIX fd per slot; peers import
 * the fds (received over SCM_RIGHTS) and map them at the same offsets. */
int kvc_export_mapped_slots(const int64_t *offsets, size_t n, int64_t group_id, int *out_fds, int64_t cap); /* returns fd count */
int kvc_map_imported_slots(const int64_t *offsets, size_t n, int64_t group_id, const int *fds, size_t n_fds);
/* The same with page ids as units (multi-row geometries whose page ids are backed by lanes: kvc_get_option(129) > 0). ONE dmabuf fd
 * per BUFFER - up to 8 page ids live in one, their lanes side by side - instead of one per 2 MiB slot (64 per page id for
 * Llama-3-8B), and three numbers per page id that tell the peer where its pages are: meta[3i] = which of the exported fds,
 * meta[3i+1] = lanes that buffer holds (k), meta[3i+2] = the lane that is page id i's (j); the page behind row r is at byte
 * (r * k + j) * page_size of the buffer. kvc_export_page_ids returns the number of fds written (<= n; with cap < n: n, nothing
 * exported); KVC_E_INVALID if one of the page ids is not backed by a lane of this process (then export slot by slot).
 * kvc_map_imported_page_ids imports every fd once and maps the page ids at the same offsets - every row of every page id must
 * be unbacked; page ids that are neighbours in one buffer are one ioctl per row - and owns the imported buffers from then on (each
 * is released when the last page id it backs is unmapped, after the TLB invalidation that imports always get inside the call). */
int kvc_export_page_ids(const int64_t *offsets, size_t n, int64_t group_id, int *out_fds, int64_t *out_meta, int64_t cap);
int kvc_map_imported_page_ids(const int64_t *offsets, size_t n, int64_t group_id, const int *fds, size_t n_fds, const int64_t *meta);

#ifdef __cplusplus
}
#endif
#endif /* KVCACHED_AMD_H */
