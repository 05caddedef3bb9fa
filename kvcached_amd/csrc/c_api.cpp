// c_api.cpp — the extern "C" surface declared in include/kvcached_amd.h.
#include "../../include/kvcached_amd.h"

#include <algorithm>
#include <cstring>

#include "kernels.hpp"
#include "kv_allocator.hpp"
#include "page_allocator.hpp"

using namespace kvc;

namespace {
thread_local std::string g_last_error;

int fail(int code, const char *what) {
  g_last_error = what ? what : "";
  return code;
}

// Maps the exception zoo onto KVC_E_* and records the message.
template <class F> auto guarded(F &&f) -> decltype(f()) {
  using R = decltype(f());
  try {
    return f();
  } catch (const NoPagesError &e) {
    return (R)fail(KVC_E_NO_PAGES, e.what());
  } catch (const GpuError &e) {
    return (R)fail(KVC_E_GPU, e.what());
  } catch (const NoGpuError &e) {
    return (R)fail(KVC_E_NO_GPU, e.what());
  } catch (const InvalidError &e) {
    return (R)fail(KVC_E_INVALID, e.what());
  } catch (const CallbackError &e) {
    return (R)fail(KVC_E_CALLBACK, e.what());
  } catch (const std::exception &e) {
    return (R)fail(KVC_E_RUNTIME, e.what());
  } catch (...) {
    return (R)fail(KVC_E_RUNTIME, "unknown error");
  }
}

template <class C> int64_t copy_out(const C &v, int64_t *out, int64_t cap) {
  const int64_t n = (int64_t)v.size();
  if (out && cap >= n) std::copy(v.begin(), v.end(), out);
  return n;
}

inline InternalPage *P(kvc_page_t *p) { return reinterpret_cast<InternalPage *>(p); }
inline const InternalPage *P(const kvc_page_t *p) { return reinterpret_cast<const InternalPage *>(p); }
inline PageAllocator *A(kvc_page_allocator_t *p) { return reinterpret_cast<PageAllocator *>(p); }
inline const PageAllocator *A(const kvc_page_allocator_t *p) { return reinterpret_cast<const PageAllocator *>(p); }
} // namespace

extern "C" {

const char *kvc_last_error(void) { return g_last_error.c_str(); }
int kvc_abi_version(void) { return KVC_ABI_VERSION; }

// ---------------------------------------------------------------- allocator lifecycle
int kvc_init(const char *dev_str, size_t page_size, int contiguous_layout) {
  return guarded([&]() -> int {
    if (!dev_str) throw InvalidError("dev_str is NULL");
    KvAllocator::init(dev_str, page_size, contiguous_layout != 0);
    return KVC_OK;
  });
}
int kvc_shutdown(void) {
  return guarded([&]() -> int {
    KvAllocator::shutdown();
    return KVC_OK;
  });
}
int kvc_create_kv_tensors(size_t size, size_t dtype_size, const char *dev_str, int64_t num_layers,
                          int64_t num_kv_buffers, int64_t group_id, int unified_pool, void **out_ptrs,
                          size_t *out_nbytes, int64_t *inout_count) {
  return guarded([&]() -> int {
    if (!out_ptrs || !out_nbytes || !inout_count) throw InvalidError("output arrays are NULL");
    auto descs = KvAllocator::global(group_id)->create_kv_tensors(size, dtype_size, dev_str ? dev_str : "", num_layers,
                                                                  num_kv_buffers, unified_pool != 0);
    if ((int64_t)descs.size() > *inout_count) throw InvalidError("output arrays too small for the tensor list");
    for (size_t i = 0; i < descs.size(); ++i) {
      out_ptrs[i] = descs[i].ptr;
      out_nbytes[i] = descs[i].nbytes;
    }
    *inout_count = (int64_t)descs.size();
    return KVC_OK;
  });
}
int kvc_kv_tensors_created(int64_t group_id) {
  return guarded([&]() -> int { return KvAllocator::global(group_id)->kv_tensors_created() ? 1 : 0; });
}
int kvc_get_device(int *is_gpu, int *index) {
  if (!KvAllocator::initialized()) return fail(KVC_E_INVALID, "init_kvcached has not been called");
  DeviceSpec d = KvAllocator::device();
  if (is_gpu) *is_gpu = d.is_gpu ? 1 : 0;
  if (index) *index = d.index;
  return KVC_OK;
}
int kvc_map_to_kv_tensors(const int64_t *offsets, size_t n, int64_t group_id) {
  return guarded([&]() -> int {
    return KvAllocator::global(group_id)->map_to_kv_tensors(offsets, n) ? KVC_OK
                                                                         : fail(KVC_E_NOT_CREATED, "KV tensors are not created");
  });
}
int kvc_unmap_from_kv_tensors(const int64_t *offsets, size_t n, int64_t group_id) {
  return guarded([&]() -> int {
    return KvAllocator::global(group_id)->unmap_from_kv_tensors(offsets, n)
               ? KVC_OK
               : fail(KVC_E_NOT_CREATED, "KV tensors are not created");
  });
}

int kvc_set_option(int opt, int64_t value) {
  switch (opt) {
  case KVC_OPT_ZERO_BACKFILL: options().zero_backfill = value; break;
  case KVC_OPT_ZERO_FILL: options().zero_fill = value; break;
  case KVC_OPT_POOL_BYTES: options().pool_bytes = value; break;
  case KVC_OPT_PROFILE: options().profile = value; break;
  case KVC_OPT_TLB_SHOOTDOWN: options().tlb_shootdown = value; break;
  case KVC_OPT_DEFER_UNMAP_SHOOTDOWN: options().defer_unmap_shootdown = value; break;
  case KVC_OPT_ASYNC_UNMAP:
    if (!value) KvAllocator::flush_all_unmaps(); // switching off: nothing may stay queued
    options().async_unmap = value;
    break;
  case KVC_OPT_UNMAP_INVALIDATION_US:
    if (value <= 0) { // switching off: nothing may stay parked behind an invalidation nobody is in a hurry for
      if (GpuContext *ctx = KvAllocator::gpu()) ctx->flush_limbo();
    }
    options().deferred_unmap_flush_us = value < 0 ? 0 : value;
    break;
  case 102: options().access_run_slots = value; break; // tuning only
  case 103: options().zero_alias_fanout = value; break; // takes effect at the next create_kv_tensors
  case 105: options().fill_chunk_slots = value < 1 ? 1 : value; break; // tuning only
  case 107: options().pool_idle_ms = value < 0 ? 0 : value; break;
  case 109: options().async_shootdown = value; break;
  case 104: fail_after_creates() = value; break;       // fault injection: the (value+1)-th hipMemCreate fails
  case 100: options().fill_variant = value; break;    // tuning only
  case 101: options().compact_variant = value; break; // tuning only
  default: return fail(KVC_E_INVALID, "unknown option");
  }
  return KVC_OK;
}
int64_t kvc_get_option(int opt) {
  switch (opt) {
  case KVC_OPT_ZERO_BACKFILL: return options().zero_backfill;
  case KVC_OPT_ZERO_FILL: return options().zero_fill;
  case KVC_OPT_POOL_BYTES: return options().pool_bytes;
  case KVC_OPT_PROFILE: return options().profile;
  case KVC_OPT_TLB_SHOOTDOWN: return options().tlb_shootdown;
  case KVC_OPT_DEFER_UNMAP_SHOOTDOWN: return options().defer_unmap_shootdown;
  case KVC_OPT_ASYNC_UNMAP: return options().async_unmap;
  case KVC_OPT_UNMAP_INVALIDATION_US: return options().deferred_unmap_flush_us;
  case 102: return options().access_run_slots;
  case 103: return options().zero_alias_fanout;
  case 105: return options().fill_chunk_slots;
  case 107: return options().pool_idle_ms;
  case 109: return options().async_shootdown;
  case 108: return vmm_backend().load(); // effective VMM backend: 0 hip, 2 hybrid, 3 drm (read-only)
  case 111: return background_shootdowns().load(); // TLB invalidations done by the library's own threads (read-only)
  case 110: return vmm_backend().load() == kVmmDrm && DrmVm::instance().kfd_ready() ? 1 : 0; // physical pages straight from KFD (read-only)
  case 112: return DrmVm::instance().create_times().alloc_ns.load();  // KFD allocation ioctls, ns (read-only, 112-117)
  case 113: return DrmVm::instance().create_times().export_ns.load(); // dmabuf exports
  case 114: return DrmVm::instance().create_times().import_ns.load(); // imports into DRM
  case 115: return DrmVm::instance().create_times().count.load();
  case 116: return DrmVm::instance().create_times().free_ns.load();   // DRM + KFD frees
  case 117: return DrmVm::instance().create_times().frees.load();
  // 130..149: host nanoseconds of the map / unmap calls by segment since the last reset (read-only diagnostics).
  // map: 130 offsets -> slots, 131 classify, 132 sort into runs, 133 pool, 134 page-table ioctls, 135 per-run bookkeeping,
  // 136 invalidation owed, 137 wait for own fill, 138 wait for the scrub of these pages, 139 (a count: scrubs behind the newest); unmap: 140 offsets -> slots, 141 runs, 142 page-table ioctls, 143 per-slot bookkeeping,
  // 144 remainders, 145 epochs, 146 invalidation, 147 scrub launch, 148 pool
  case 130: case 131: case 132: case 133: case 134: case 135: case 136: case 137: case 138: case 139:
  case 140: case 141: case 142: case 143: case 144: case 145: case 146: case 147: case 148: case 149:
    return stats().seg[opt - 130];
  // 150..161: more of the same. 150 map: rewrite of split PRT / zero-extent remainders (a share of 136), 151 map: page-table
  // ioctls issued (a count), 152 unmap: page-table ioctls issued (a count), 153 map: runs of adjacent slots (a count)
  case 150: case 151: case 152: case 153: case 154: case 155: case 156: case 157: case 158: case 159: case 160: case 161:
    return stats().seg[opt - 130];
  case 118: { // the direct KFD TLB flush is what tlb_shootdown() uses (read-only)
    GpuContext *ctx = KvAllocator::gpu();
    return ctx && ctx->kfd_flush_active() ? 1 : 0;
  }
  case 119: return options().phys_chunk_pages; // pages per extent at most (read-only; KVCACHED_PHYS_CHUNK_PAGES at init)
  case 120: case 121: case 122: case 123: { // footprint of the pool the engine's pages come from, in pages: held from the driver,
    // handed out, free inside partly used buffers (the waste), and the size new buffers get right now (read-only)
    GpuContext *ctx = KvAllocator::gpu();
    if (!ctx) return 0;
    ExtentPool *p = ctx->primary_pool();
    if (!p) return opt == 123 ? (int64_t)options().phys_chunk_pages.load() : 0; // (nothing mapped yet: no pool is made for a diagnostic)
    const auto f = p->footprint();
    const int64_t scale = (int64_t)p->counter_scale();
    return scale * (int64_t)(opt == 120 ? f.held_pages : opt == 121 ? f.out_pages : opt == 122 ? f.free_pieces : f.extent_pages_now);
  }
  case 127: { // pages of the zero extent behind compat-mode regions (0: none - PRT, or sharded zero pages through ROCr; read-only)
    GpuContext *ctx = KvAllocator::gpu();
    return ctx ? (int64_t)ctx->zero_extent_pages(KvAllocator::page_size()) : 0;
  }
  case 128: { // unbacked slots of group 0 are PRT mappings: reads return 0, writes are dropped, nothing faults (read-only)
    try {
      return KvAllocator::initialized() && KvAllocator::global(0)->uses_prt() ? 1 : 0;
    } catch (...) {
      return 0;
    }
  }
  case 125: return stats().pages_scrubbed;    // pages zeroed on their way back (read-only; reset with the stats)
  case 126: return stats().pages_prescrubbed; // pages a map call handed out without launching a fill for them
  case 124: { // releases of pieces the pool did not know (must stay 0; read-only)
    GpuContext *ctx = KvAllocator::gpu();
    ExtentPool *p = ctx ? ctx->primary_pool() : nullptr;
    return p ? (int64_t)p->footprint().bad_releases : 0;
  }
  case 163: return import_count(true);  // shared pool: pages of a peer imported straight into KFD + DRM (read-only)
  case 164: return import_count(false); // ... and through the runtime's import (ROCr / HIP): the fallback, e.g. for a buffer on another GPU
  case 129: { // page ids of group 0 are backed as units (lanes: DESIGN.md §4.11), and how many lanes a buffer holds at most (0: no; read-only)
    try {
      return KvAllocator::initialized() ? (int64_t)KvAllocator::global(0)->lanes_per_extent() : 0;
    } catch (...) {
      return 0;
    }
  }
  case 104: return fail_after_creates();
  case 100: return options().fill_variant;
  case 101: return options().compact_variant;
  default: return fail(KVC_E_INVALID, "unknown option");
  }
}

int kvc_get_stats(kvc_stats_t *o) {
  if (!o) return fail(KVC_E_INVALID, "NULL stats");
  Stats &s = stats();
  o->pages_mapped = s.pages_mapped;
  o->pages_unmapped = s.pages_unmapped;
  o->handles_created = s.vmm.created;
  o->handles_released = s.vmm.released;
  o->handles_reused = s.vmm.reused;
  o->map_calls = s.map_calls;
  o->unmap_calls = s.unmap_calls;
  o->map_ns = s.map_ns;
  o->unmap_ns = s.unmap_ns;
  o->fill_launches = s.fill_launches;
  o->fill_bytes = s.fill_bytes;
  o->compact_launches = s.compact_launches;
  o->compact_bytes = s.compact_bytes;
  o->tlb_shootdowns = s.tlb_shootdowns;
  o->shootdown_ns = s.shootdown_ns;
  o->index_launches = s.index_launches;
  o->unmaps_queued = s.unmaps_queued;
  o->unmaps_cancelled = s.unmaps_cancelled;
  std::lock_guard<std::mutex> g(s.mu);
  o->fill_ms = s.fill_ms;
  o->compact_ms = s.compact_ms;
  return KVC_OK;
}
int kvc_quiesce_begin(void) {
  return guarded([&]() -> int {
    KvAllocator::quiesce_all(true);
    return KVC_OK;
  });
}
int kvc_quiesce_end(void) {
  return guarded([&]() -> int {
    KvAllocator::quiesce_all(false);
    return KVC_OK;
  });
}
int kvc_flush_unmaps(void) {
  return guarded([&]() -> int {
    KvAllocator::flush_all_unmaps();
    return KVC_OK;
  });
}
int kvc_get_driver_breakdown(int64_t *o) {
  if (!o) return fail(KVC_E_INVALID, "NULL output");
  Stats &s = stats();
  o[0] = s.t_unmap_alias; o[1] = s.t_acquire; o[2] = s.t_map; o[3] = s.t_access;
  o[4] = s.t_unmap; o[5] = s.t_release; o[6] = s.t_realias; o[7] = s.t_sync;
  return KVC_OK;
}
int kvc_reset_stats(void) {
  stats().reset();
  return KVC_OK;
}

int kvc_mem_get_info(size_t *free_bytes, size_t *total_bytes) {
  return guarded([&]() -> int {
    mem_get_info(free_bytes, total_bytes);
    return KVC_OK;
  });
}
int kvc_set_mem_info_override(size_t free_bytes, size_t total_bytes) {
  set_mem_info_override(free_bytes, total_bytes);
  return KVC_OK;
}

// ---------------------------------------------------------------- InternalPage
kvc_page_t *kvc_page_new(int64_t page_id, int64_t page_size) {
  return reinterpret_cast<kvc_page_t *>(new InternalPage(page_id, page_size));
}
void kvc_page_delete(kvc_page_t *p) { delete P(p); }
int64_t kvc_page_id(const kvc_page_t *p) { return P(p)->page_id; }
int64_t kvc_page_size(const kvc_page_t *p) { return P(p)->page_size; }
void kvc_page_init(kvc_page_t *p, int64_t block_mem_size) { P(p)->init(block_mem_size); }
int64_t kvc_page_alloc(kvc_page_t *p, int64_t num_blocks, int64_t *out) {
  return guarded([&]() -> int64_t {
    auto v = P(p)->alloc(num_blocks);
    return copy_out(v, out, num_blocks);
  });
}
void kvc_page_free(kvc_page_t *p, int64_t block_id) { P(p)->free(block_id); }
void kvc_page_free_batch(kvc_page_t *p, const int64_t *ids, size_t n) { P(p)->free_batch(ids, n); }
int kvc_page_empty(const kvc_page_t *p) { return P(p)->empty(); }
int kvc_page_full(const kvc_page_t *p) { return P(p)->full(); }
int64_t kvc_page_num_free_blocks(const kvc_page_t *p) { return P(p)->num_free_blocks(); }
int64_t kvc_page_get_free_blocks(const kvc_page_t *p, int64_t *out, int64_t cap) {
  return copy_out(P(p)->free_blocks(), out, cap);
}
void kvc_page_get_block_range(int64_t page_id, int64_t page_size, int64_t block_mem_size, int64_t *start, int64_t *end) {
  auto r = InternalPage::get_block_range(page_id, page_size, block_mem_size);
  *start = r.first;
  *end = r.second;
}
int64_t kvc_page_get_num_blocks(int64_t page_size, int64_t block_mem_size) {
  return InternalPage::get_num_blocks(page_size, block_mem_size);
}

// ---------------------------------------------------------------- PageAllocator
kvc_page_allocator_t *kvc_pa_new(int64_t num_layers, int64_t mem_size_per_layer, int64_t page_size, int64_t world_size,
                                 int64_t pp_rank, int async_sched, int contiguous_layout, int enable_page_prealloc,
                                 int64_t num_kv_buffers, int64_t group_id, const char *ipc_name) {
  try {
    return reinterpret_cast<kvc_page_allocator_t *>(
        new PageAllocator(num_layers, mem_size_per_layer, page_size, world_size, pp_rank, async_sched != 0,
                          contiguous_layout != 0, enable_page_prealloc != 0, num_kv_buffers, group_id,
                          ipc_name ? ipc_name : ""));
  } catch (const std::exception &e) {
    g_last_error = e.what();
    return nullptr;
  }
}
void kvc_pa_delete(kvc_page_allocator_t *pa) { delete A(pa); }
int kvc_pa_start_prealloc_thread(kvc_page_allocator_t *pa) {
  return guarded([&]() -> int {
    A(pa)->start_prealloc_thread();
    return KVC_OK;
  });
}
int kvc_pa_stop_prealloc_thread(kvc_page_allocator_t *pa) {
  return guarded([&]() -> int {
    A(pa)->stop_prealloc_thread();
    return KVC_OK;
  });
}
int64_t kvc_pa_alloc_page(kvc_page_allocator_t *pa) {
  return guarded([&]() -> int64_t { return A(pa)->alloc_page(); });
}
int64_t kvc_pa_alloc_pages(kvc_page_allocator_t *pa, int64_t n, int64_t *out_ids) {
  return guarded([&]() -> int64_t {
    if (n > 0 && !out_ids) throw InvalidError("out_ids is NULL");
    auto ids = A(pa)->alloc_pages(n);
    std::copy(ids.begin(), ids.end(), out_ids);
    return (int64_t)ids.size();
  });
}
int kvc_pa_free_page(kvc_page_allocator_t *pa, int64_t page_id) {
  return guarded([&]() -> int {
    A(pa)->free_page(page_id);
    return KVC_OK;
  });
}
int kvc_pa_free_pages(kvc_page_allocator_t *pa, const int64_t *page_ids, size_t n) {
  return guarded([&]() -> int {
    A(pa)->free_pages(page_ids, n);
    return KVC_OK;
  });
}
int kvc_pa_resize(kvc_page_allocator_t *pa, int64_t new_mem_size) {
  return guarded([&]() -> int { return A(pa)->resize(new_mem_size) ? 1 : 0; });
}
int kvc_pa_trim(kvc_page_allocator_t *pa) {
  return guarded([&]() -> int {
    A(pa)->trim();
    return KVC_OK;
  });
}
int kvc_pa_reset_free_page_order(kvc_page_allocator_t *pa) {
  return guarded([&]() -> int {
    A(pa)->reset_free_page_order();
    return KVC_OK;
  });
}
int64_t kvc_pa_get_num_free_pages(const kvc_page_allocator_t *pa) { return A(pa)->get_num_free_pages(); }
int64_t kvc_pa_get_num_inuse_pages(const kvc_page_allocator_t *pa) { return A(pa)->get_num_inuse_pages(); }
int64_t kvc_pa_get_num_total_pages(const kvc_page_allocator_t *pa) { return A(pa)->get_num_total_pages(); }
int64_t kvc_pa_get_num_reserved_pages(const kvc_page_allocator_t *pa) { return A(pa)->get_num_reserved_pages(); }
int64_t kvc_pa_get_avail_physical_pages(const kvc_page_allocator_t *pa) {
  return guarded([&]() -> int64_t { return A(pa)->get_avail_physical_pages(); });
}
int64_t kvc_pa_check_and_get_resize_target(const kvc_page_allocator_t *pa, int64_t current_mem_size) {
  return A(pa)->check_and_get_resize_target(current_mem_size);
}
int64_t kvc_pa_get_resize_target(const kvc_page_allocator_t *pa) { return A(pa)->get_resize_target(); }
int64_t kvc_pa_get_page_id(const kvc_page_allocator_t *pa, int64_t block_id, int64_t block_mem_size) {
  return A(pa)->get_page_id(block_id, block_mem_size);
}
int64_t kvc_pa_group_indices_by_page(const kvc_page_allocator_t *pa, const int64_t *indices, size_t n,
                                     int64_t block_mem_size, int64_t *keys, int64_t *counts, int64_t *values) {
  return guarded([&]() -> int64_t {
    auto groups = A(pa)->group_indices_by_page(indices, n, block_mem_size);
    int64_t k = 0, w = 0;
    for (auto &kv : groups) { // iteration order is the contract
      keys[k] = kv.first;
      counts[k] = (int64_t)kv.second.size();
      std::memcpy(values + w, kv.second.data(), kv.second.size() * sizeof(int64_t));
      w += (int64_t)kv.second.size();
      ++k;
    }
    return k;
  });
}
int kvc_pa_set_broadcast_map_callback(kvc_page_allocator_t *pa, kvc_broadcast_cb cb, void *user) {
  A(pa)->set_broadcast_map_callback(
      cb ? BroadcastFn([cb, user](int64_t ws, const offset_t *o, size_t n) { return cb(user, ws, o, n); }) : BroadcastFn());
  return KVC_OK;
}
int kvc_pa_set_broadcast_unmap_callback(kvc_page_allocator_t *pa, kvc_broadcast_cb cb, void *user) {
  A(pa)->set_broadcast_unmap_callback(
      cb ? BroadcastFn([cb, user](int64_t ws, const offset_t *o, size_t n) { return cb(user, ws, o, n); }) : BroadcastFn());
  return KVC_OK;
}
int kvc_pa_set_should_use_worker_ipc_callback(kvc_page_allocator_t *pa, kvc_bool_cb cb, void *user) {
  A(pa)->set_should_use_worker_ipc_callback(cb ? BoolFn([cb, user]() { return cb(user) != 0; }) : BoolFn());
  return KVC_OK;
}
int64_t kvc_pa_get_page_list(const kvc_page_allocator_t *pa, int which, int64_t *out, int64_t cap) {
  return copy_out(A(pa)->page_list(which), out, cap);
}
const char *kvc_pa_ipc_name(const kvc_page_allocator_t *pa) { return A(pa)->ipc_name().c_str(); }

// ---------------------------------------------------------------- kernels
int kvc_zero_fill_pages(void *const *page_ptrs, size_t n, size_t page_bytes, void *stream, int sync) {
  return guarded([&]() -> int {
    GpuContext *ctx = KvAllocator::gpu();
    if (!ctx) throw NoGpuError("zero_fill_pages needs init_kvcached on a GPU device");
    if (n && !page_ptrs) throw InvalidError("page_ptrs is NULL");
    if (page_bytes == 0 || page_bytes % kFillSlabBytes != 0) throw InvalidError("page_bytes must be a multiple of 64 KiB");
    for (size_t i = 0; i < n; ++i)
      if (reinterpret_cast<uintptr_t>(page_ptrs[i]) % 16 != 0) throw InvalidError("page pointer is not 16-byte aligned");
    ctx->bind();
    ctx->zero_fill(page_ptrs, n, page_bytes, static_cast<hipStream_t>(stream));
    if (sync) ctx->sync(static_cast<hipStream_t>(stream));
    return KVC_OK;
  });
}

int kvc_compact_blocks(void *const *region_bases, size_t n_regions, const int64_t *src_blocks, const int64_t *dst_blocks,
                       size_t n_moves, size_t block_bytes, void *stream, int sync) {
  return guarded([&]() -> int {
    GpuContext *ctx = KvAllocator::gpu();
    if (!ctx) throw NoGpuError("compact_blocks needs init_kvcached on a GPU device");
    if ((n_regions && !region_bases) || (n_moves && (!src_blocks || !dst_blocks))) throw InvalidError("NULL argument");
    if (block_bytes == 0 || block_bytes % 16 != 0) throw InvalidError("block_bytes must be a positive multiple of 16");
    for (size_t i = 0; i < n_moves; ++i)
      if (src_blocks[i] < 0 || dst_blocks[i] < 0) throw InvalidError("negative block id");
    ctx->bind();
    ctx->compact(region_bases, n_regions, src_blocks, dst_blocks, n_moves, block_bytes, static_cast<hipStream_t>(stream));
    if (sync) ctx->sync(static_cast<hipStream_t>(stream));
    return KVC_OK;
  });
}

int64_t kvc_get_region_bases(int64_t group_id, void **out, int64_t cap) {
  return guarded([&]() -> int64_t {
    auto b = KvAllocator::global(group_id)->region_bases();
    if (out && cap >= (int64_t)b.size()) std::copy(b.begin(), b.end(), out);
    return (int64_t)b.size();
  });
}

// ---------------------------------------------------------------- block ids <-> token indices
namespace {
GpuContext *index_ctx(const char *what) {
  GpuContext *ctx = KvAllocator::gpu();
  if (!ctx) throw NoGpuError(std::string(what) + " needs init_kvcached on a GPU device");
  ctx->bind();
  return ctx;
}
} // namespace

int kvc_expand_block_ids(const int64_t *block_ids, size_t n, int64_t tpb, int64_t *out_dev, void *stream) {
  return guarded([&]() -> int {
    GpuContext *ctx = index_ctx("expand_block_ids");
    if (tpb <= 0 || tpb > 0x7fffffff) throw InvalidError("tokens_per_block must be in [1, 2^31)");
    if (n && (!block_ids || !out_dev)) throw InvalidError("NULL argument");
    ctx->expand_block_ids(block_ids, n, tpb, out_dev, static_cast<hipStream_t>(stream));
    return KVC_OK;
  });
}

int kvc_alloc_extend_indices(const int64_t *prefix_lens_dev, const int64_t *seq_lens_dev, const int64_t *last_loc_dev,
                             size_t bs, const int64_t *new_block_ids, size_t n_new, int64_t tpb, int64_t *out_dev,
                             size_t extend_num_tokens, void *stream) {
  return guarded([&]() -> int {
    GpuContext *ctx = index_ctx("alloc_extend_indices");
    if (tpb <= 0 || tpb > 0x7fffffff) throw InvalidError("tokens_per_block must be in [1, 2^31)");
    if (bs && (!prefix_lens_dev || !seq_lens_dev || !last_loc_dev)) throw InvalidError("NULL argument");
    if ((n_new && !new_block_ids) || (extend_num_tokens && !out_dev)) throw InvalidError("NULL argument");
    if (extend_num_tokens > (size_t)65535 * 4096) throw InvalidError("extend_num_tokens too large for one call");
    ctx->alloc_extend_indices(prefix_lens_dev, seq_lens_dev, last_loc_dev, bs, new_block_ids, n_new, tpb, out_dev,
                              extend_num_tokens, static_cast<hipStream_t>(stream));
    return KVC_OK;
  });
}

int kvc_alloc_decode_indices(const int64_t *seq_lens_dev, const int64_t *last_loc_dev, size_t bs,
                             const int64_t *new_block_ids, size_t n_new, int64_t tpb, int64_t *out_dev, void *stream) {
  return guarded([&]() -> int {
    GpuContext *ctx = index_ctx("alloc_decode_indices");
    if (tpb <= 0 || tpb > 0x7fffffff) throw InvalidError("tokens_per_block must be in [1, 2^31)");
    if (bs && (!seq_lens_dev || !last_loc_dev || !out_dev)) throw InvalidError("NULL argument");
    if (n_new && !new_block_ids) throw InvalidError("NULL argument");
    ctx->alloc_decode_indices(seq_lens_dev, last_loc_dev, bs, new_block_ids, n_new, tpb, out_dev,
                              static_cast<hipStream_t>(stream));
    return KVC_OK;
  });
}

int64_t kvc_unique_block_ids(const int64_t *token_indices_dev, size_t n, int64_t tpb, int64_t num_blocks, int64_t *out_host,
                             size_t cap, void *stream) {
  return guarded([&]() -> int64_t {
    GpuContext *ctx = index_ctx("unique_block_ids");
    if (n && !token_indices_dev) throw InvalidError("NULL argument");
    return ctx->unique_block_ids(token_indices_dev, n, tpb, num_blocks, out_host, cap, static_cast<hipStream_t>(stream));
  });
}

// ---------------------------------------------------------------- TP shared pool
int kvc_export_mapped_slots(const int64_t *offsets, size_t n, int64_t group_id, int *out_fds, int64_t cap) {
  return guarded([&]() -> int { return KvAllocator::global(group_id)->export_mapped_slots(offsets, n, out_fds, cap); });
}
int kvc_map_imported_slots(const int64_t *offsets, size_t n, int64_t group_id, const int *fds, size_t n_fds) {
  return guarded([&]() -> int {
    return KvAllocator::global(group_id)->map_imported_slots(offsets, n, fds, n_fds) ? KVC_OK
                                                                                      : fail(KVC_E_NOT_CREATED, "KV tensors are not created");
  });
}
int kvc_export_page_ids(const int64_t *offsets, size_t n, int64_t group_id, int *out_fds, int64_t *out_meta, int64_t cap) {
  return guarded([&]() -> int { return KvAllocator::global(group_id)->export_page_ids(offsets, n, out_fds, out_meta, cap); });
}
int kvc_map_imported_page_ids(const int64_t *offsets, size_t n, int64_t group_id, const int *fds, size_t n_fds, const int64_t *meta) {
  return guarded([&]() -> int {
    return KvAllocator::global(group_id)->map_imported_page_ids(offsets, n, fds, n_fds, meta) ? KVC_OK
                                                                                          : fail(KVC_E_NOT_CREATED, "KV tensors are not created");
  });
}

} // extern "C"
