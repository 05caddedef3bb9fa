// kv_allocator.hpp — VA regions + the batched map/unmap hot path.
//
// Counterpart of the reference's FTensor (csrc/ftensor.cpp), GPUPage/CPUPage (csrc/page.cpp)
// and FTensorAllocator (csrc/allocator.cpp), re-designed around what the MI355X driver
// charges for (hip_vmm.hpp): slots are tracked in flat vectors instead of one heap object
// + hash-map node per page, a whole map_to_kv_tensors() call is one batch (pooled handles,
// ranged hipMemSetAccess, one zero_fill_pages launch per <=256 pages overlapped with the
// remaining driver calls, a single stream sync at the end), and failures roll the batch back
// instead of aborting the process.
#pragma once

#include <condition_variable>
#include <deque>
#include <functional>
#include <map>
#include <memory>
#include <mutex>
#include <thread>
#include <string>
#include <unordered_map>
#include <vector>

#include "common.hpp"
#include "hip_vmm.hpp"

namespace kvc {

struct Options {
  std::atomic<int64_t> zero_backfill{0};
  std::atomic<int64_t> zero_fill{1};
  std::atomic<int64_t> pool_bytes{16384ll << 20};
  std::atomic<int64_t> profile{0};
  std::atomic<int64_t> tlb_shootdown{1};
  std::atomic<int64_t> pool_idle_ms{1000};       // idle handles older than this are released (0 = keep until pressure)
  std::atomic<int64_t> async_unmap{0};           // unmap_from_kv_tensors only queues; a reclaimer thread does the driver calls
  std::atomic<int64_t> async_shootdown{1};       // with a housekeeping thread around, unmap leaves its TLB invalidation to it
  std::atomic<int64_t> hip_reg_group_mb{64};     // hybrid/drm: VA introduced to HIP per hipMemMap (0 = slot by slot)
  std::atomic<int64_t> clear_run_slots{16};      // drm backend: unmap runs of adjacent slots with one CLEAR ioctl (0 = off)
  std::atomic<int64_t> phys_chunk_pages{64};     // drm backend: a run of adjacent slots is backed by ONE buffer of up to this many pages (1 = off)
  std::atomic<int64_t> extent_waste_pct{5};      // ... and new extents shrink while free pieces of partly used ones exceed this share of the pages in use
  std::atomic<int64_t> phys_reserve_bytes{0};    // idle physical memory the housekeeping thread keeps ready (pre-created, never below it)
  std::atomic<int64_t> map_waits_for_all_flushes{0}; // 1 = a map batch waits for every invalidation owed, not only those of its own slots
  std::atomic<int64_t> scrub_on_release{1};      // drm backend: pages are zeroed (through an alias mapping) when they come back, not when they go out
  std::atomic<int64_t> map_shootdown_always{0};  // 1 = invalidate after every map batch even when no stale translation can exist
  std::atomic<int64_t> defer_unmap_shootdown{0}; // unmap's invalidation may wait for the next map batch / driver release
  // compat regions: the invalidation an unmap owes may trail the call by at most `deferred_unmap_flush_us` (0 = inside the call,
  // the default): the pages wait un-scrubbed and un-offered until it has happened, and the next map batch's own invalidation
  // absorbs it if it comes first (KVCACHED_UNMAP_INVALIDATION_US)
  std::atomic<int64_t> deferred_unmap_flush_us{0};
  std::atomic<int64_t> access_run_slots{1}; // max mappings one hipMemSetAccess call may span
  std::atomic<int64_t> zero_alias_fanout{256}; // unbacked slots that share one physical zero page
  std::atomic<int64_t> fill_chunk_slots{1024}; // slots made usable (one TLB shootdown + fill launches) at a time
  std::atomic<int64_t> fill_variant{0};
  std::atomic<int64_t> compact_variant{0};
};
Options &options();

struct Stats {
  std::atomic<int64_t> pages_mapped{0}, pages_unmapped{0};
  std::atomic<int64_t> map_calls{0}, unmap_calls{0}, map_ns{0}, unmap_ns{0};
  std::atomic<int64_t> fill_launches{0}, fill_bytes{0};
  std::atomic<int64_t> compact_launches{0}, compact_bytes{0};
  std::atomic<int64_t> tlb_shootdowns{0}, shootdown_ns{0};
  std::atomic<int64_t> index_launches{0};
  std::atomic<int64_t> pages_scrubbed{0}, pages_prescrubbed{0}; // zeroed on their way back / handed out without a fill of their own (options 125, 126)
  std::atomic<int64_t> unmaps_queued{0}, unmaps_cancelled{0}; // async unmap: slots queued / re-backed before the reclaimer got to them
  // host time inside each driver call of the map/unmap paths (diagnostics; kvc_get_driver_breakdown)
  std::atomic<int64_t> t_unmap_alias{0}, t_acquire{0}, t_map{0}, t_access{0}, t_unmap{0}, t_release{0}, t_realias{0}, t_sync{0};
  // host time of the map / unmap calls by segment (diagnostics, read-only options 130 + i; names in c_api.cpp)
  std::atomic<int64_t> seg[32] = {};
  std::mutex mu;
  double fill_ms = 0, compact_ms = 0;
  VmmCounters vmm;
  void reset();
};
Stats &stats();
struct SegTimer { // accumulates the time since the previous mark into stats().seg[i]
  int64_t t = now_ns();
  void mark(int i) {
    const int64_t n = now_ns();
    stats().seg[i] += n - t;
    t = n;
  }
};
std::atomic<int64_t> &background_shootdowns(); // TLB invalidations performed off the callers' threads (option 111)

// Per-device state: stream for our kernels, event pool for per-launch timing, handle pools.
class GpuContext {
public:
  explicit GpuContext(int dev);
  ~GpuContext();
  int dev() const { return dev_; }
  void bind() const; // hipSetDevice for the calling thread (HIP's current device is per thread)
  hipStream_t stream() const { return stream_; }
  // What map/unmap take pages from (extent_pool.hpp): run-sized extents of up to KVCACHED_PHYS_CHUNK_PAGES pages with
  // the drm backend and pages straight from KFD (non-exportable pools), single pages everywhere else.
  ExtentPool *extents(size_t page_bytes, bool exportable);
  // The pool of an allocator whose page ids are `rows` slots in `rows` places of the address space (one per layer and K/V
  // half: the non-contiguous layout engines use on ROCm): its unit is a LANE - the `rows` pages behind one page id - and an
  // extent is one buffer of up to KVCACHED_LANE_EXTENT_MB holding k lanes row-major (page (row, lane) at (row * k + lane)),
  // so that k consecutive page ids are ONE map ioctl per row and a page id is acquired, scrubbed and released as a unit
  // (DESIGN.md §4.11). drm backend with pages straight from KFD only (nullptr otherwise).
  ExtentPool *lane_extents(size_t rows, size_t page_bytes);
  // the pool the engine's own pages come from (the reserve and the footprint diagnostics belong to it)
  void set_primary_pool(ExtentPool *p) { primary_pool_.store(p); }
  ExtentPool *primary_pool() const { return primary_pool_.load(); }
  void drain_pools();
  size_t idle_pool_bytes(); // physical memory parked in the handle pools: ours to reuse, invisible to hipMemGetInfo
  // 10 Hz from the allocator's watcher thread: drain idle handles if the device is short of free memory, and
  // let handles that sat idle for KVCACHED_POOL_IDLE_MS go back to the driver (ExtentPool::decay), keep the reserve filled
  void housekeeping();
  // a thread that calls housekeeping() periodically exists / is gone (PageAllocator's watcher): while one does,
  // pools hand over-cap handles to it instead of releasing them on the caller's free() path
  void add_housekeeper(int delta);
  bool has_housekeeper() const { return housekeepers_.load() > 0; }

  // kernel launches on `s` (NULL = own stream), timed with events when profiling is on
  void zero_fill(void *const *pages, size_t n, size_t page_bytes, hipStream_t s);
  void compact(void *const *bases, size_t n_regions, const int64_t *src, const int64_t *dst, size_t n_moves,
               size_t block_bytes, hipStream_t s);
  void sync(hipStream_t s); // hipStreamSynchronize + harvest event timings
  // ---- zeroing pages on their way BACK (drm backend, extents with an alias mapping).
  // A page handed out by map_slots must read zero. Filling it there puts 0.3 us per page of GPU time between the
  // driver calls and the caller; filling it when it is given back - through a second, permanent mapping of its buffer
  // that only the library knows, on a stream of its own - lets that run while the host does whatever comes next.
  // scrub(): launches the fill on the given alias addresses (0 = skip) and returns its ticket; wait_scrub(t): returns
  // once scrub t has finished (no-op if it has). Tickets are issued in launch order on one stream.
  uint64_t scrub(const uint64_t *alias_addrs, size_t n, size_t page_bytes);
  void wait_scrub(uint64_t ticket);
  void wait_all_scrubs() { wait_scrub(scrub_issued_.load()); }
  uint64_t scrubs_issued() const { return scrub_issued_.load(); }
  // VA for alias mappings (never handed to anybody): carved from arenas reserved 64 GiB at a time
  uint64_t alias_alloc(size_t bytes);
  void alias_free(uint64_t va, size_t bytes);
  // The buffer of zeros behind the unbacked slots of compat-mode regions (drm backend): created and filled on first use,
  // `*pages` pages of `page_bytes`; 0 if it cannot be had (the caller aliases sharded zero pages through ROCr instead).
  phys_handle_t zero_extent(size_t page_bytes, size_t *pages);
  size_t zero_extent_pages(size_t page_bytes); // of the one that exists (0: none was needed)
  // block id <-> token index glue (index_kernels.hip); ids are HOST arrays, everything else device memory
  void expand_block_ids(const int64_t *ids, size_t n, int64_t tpb, int64_t *out, hipStream_t s);
  void alloc_extend_indices(const int64_t *pre_lens, const int64_t *seq_lens, const int64_t *last_loc, size_t bs,
                            const int64_t *ids, size_t n_ids, int64_t tpb, int64_t *out, size_t out_len, hipStream_t s);
  void alloc_decode_indices(const int64_t *seq_lens, const int64_t *last_loc, size_t bs, const int64_t *ids, size_t n_ids,
                            int64_t tpb, int64_t *out, hipStream_t s);
  // blocking: distinct blocks of `n` device token indices, ascending, into out_host; returns the count
  int64_t unique_block_ids(const int64_t *idx, size_t n, int64_t tpb, int64_t n_blocks, int64_t *out_host, size_t cap,
                           hipStream_t s);
  // Make the driver invalidate this GPU's TLBs. hipMemMap/hipMemUnmap/hipMemSetAccess do not do it
  // on ROCm 7.2 (stale translations survive a remap: kvcached_amd/csrc/tools/remap_diag.cpp); KFD's unmap ioctl does:
  // issued directly on a buffer of our own (KfdTlbFlush, drm_vm.hpp), or - where /dev/kfd cannot be used that way -
  // through an ordinary >= 2 MiB hipMalloc + hipFree. KvAllocator::init proves on a scratch slot that whichever is
  // in effect really invalidates, and refuses to start otherwise.
  void tlb_shootdown();
  bool kfd_flush_active() const { return kfd_flush_.ready(); }
  // unmap path: the invalidation is owed but nothing needs it yet (see KvAllocator::unmap_slots)
  void defer_tlb_shootdown() { tlb_stale().store(true); }
  bool tlb_owed() const { return tlb_stale().load(); } // some unmap since the last invalidation (hip_vmm.hpp)
  // Invalidate if (and only if) an unmap has happened since the last invalidation; a caller that arrives while
  // another thread's invalidation is in flight waits for it instead of issuing a second one.
  void ensure_flushed();
  void flush_deferred_shootdown() { ensure_flushed(); }
  // Invalidations are numbered. A translation removed NOW is covered by the first invalidation that starts from now on:
  // next_flush_epoch() (read after the driver's unmap call has returned) is its number, and flushed_through(e) says
  // whether that one has finished. A map batch only has to wait for the invalidations that cover ITS slots
  // (KvRegion::stale_epoch) - a stale translation of some other address cannot shadow a mapping made here.
  uint64_t next_flush_epoch() const { return flush_started_.load() + 1; }
  bool flushed_through(uint64_t epoch) const { return flush_done_.load() >= epoch; }
  void ensure_flushed_through(uint64_t epoch);
  // Have this context's own thread do ensure_flushed() right away (started on first use): the unmap path's 0.3-0.5 ms
  // KFD round trip leaves the caller's free(); whoever needs the invalidation earlier (the next map batch before
  // its first fill, a handle leaving for the driver) calls ensure_flushed() and waits for it or performs it.
  void request_async_flush(int64_t within_us = 0); // within_us > 0: at the latest that many microseconds from now, foreground calls or not
  // ---- pages in limbo: unmapped, their invalidation still owed (KVCACHED_UNMAP_INVALIDATION_US). `finish` scrubs them and
  // gives them back to their pool; it runs - on whatever thread completed an invalidation numbered >= epoch - strictly after
  // that invalidation: a page is never zeroed nor on offer while a stale translation of it can exist.
  void park(uint64_t epoch, size_t bytes, std::function<void()> finish);
  void drain_limbo();                       // runs the finishers whose invalidation is over
  void flush_limbo();                       // invalidates if anything is parked, then drains
  size_t limbo_bytes() const { return limbo_bytes_.load(); }
  // The idle memory the primary pool is kept at: KVCACHED_PHYS_RESERVE_MB (never above the pool's cap) plus - where unmaps park
  // their pages - the most that was parked at once during the last second (at most the base again): parked pages are in transit,
  // and a reserve of exactly one batch next to a parked batch leaves the next map call on the edge of having to invalidate for
  // them itself.
  size_t reserve_target_bytes() const;
  // a map / unmap call of an allocator is in progress (or was a moment ago): the background invalidation waits its turn
  struct Foreground {
    GpuContext *c;
    explicit Foreground(GpuContext *ctx) : c(ctx) {
      if (c) c->fg_active_.fetch_add(1);
    }
    ~Foreground() {
      if (c) {
        c->fg_last_ns_.store(now_ns());
        c->fg_active_.fetch_sub(1);
      }
    }
  };
  bool foreground_busy() const { return fg_active_.load() > 0 || now_ns() - fg_last_ns_.load() < 150000; }

private:
  struct Timed {
    hipEvent_t a, b;
    int kind; // 0 fill, 1 compact
  };
  void begin_timed(hipStream_t s, int kind);
  void end_timed(hipStream_t s);
  void harvest();
  void flusher_loop();
  void do_shootdown(); // needs flush_mu_
  int dev_;
  hipStream_t stream_ = nullptr;
  std::mutex flush_mu_; // serialises TLB invalidations
  std::atomic<uint64_t> flush_started_{0}, flush_done_{0};
  std::atomic<int> fg_active_{0};
  std::atomic<int64_t> fg_last_ns_{0};
  std::thread flusher_;
  std::mutex fl_mu_;
  std::condition_variable fl_cv_;
  bool fl_stop_ = false, fl_kick_ = false;
  int64_t fl_deadline_ns_ = 0; // (under fl_mu_) the async flush must have started by then (0: when the foreground is quiet)
  struct Parked {
    uint64_t epoch;
    size_t bytes;
    std::function<void()> finish;
  };
  std::mutex limbo_mu_;
  std::deque<Parked> limbo_;
  std::atomic<size_t> limbo_bytes_{0};
  static constexpr int kLimboTicks = 10;
  std::atomic<size_t> limbo_peak_[kLimboTicks] = {}; // per housekeeping tick (one slot for ever where no thread ticks)
  std::atomic<unsigned> limbo_tick_{0};
  std::atomic<int> housekeepers_{0};
  std::mutex mu_;
  std::unordered_map<size_t, std::unique_ptr<ExtentPool>> extent_pools_[2]; // [exportable], key: page bytes
  std::map<std::pair<size_t, size_t>, std::unique_ptr<ExtentPool>> lane_pools_; // key: (rows, page bytes)
  std::atomic<ExtentPool *> primary_pool_{nullptr};
  // the reserve follows demand (housekeeping(), one thread): units created on callers' paths per tick over the last second
  std::mutex hk_mu_;
  static constexpr size_t kDemandTicks = 10;
  ExtentPool *demand_pool_ = nullptr;
  size_t demand_seen_ = 0, demand_window_[kDemandTicks] = {}, demand_tick_ = 0, demand_quiet_ticks_ = 0, reserve_boost_units_ = 0;
  ExtentDriver make_driver(size_t unit_bytes, bool exportable, bool aliases);
  std::vector<ExtentPool *> all_pools();
  KfdTlbFlush kfd_flush_;
  void *fallback_block_ = nullptr; // the 2 MiB allocation whose hipFree is the next invalidation of the hipMalloc/hipFree fallback
  hipStream_t scrub_stream_ = nullptr;
  std::mutex scrub_mu_; // launch order = ticket order
  std::atomic<uint64_t> scrub_issued_{0}, scrub_done_{0};
  std::deque<std::pair<uint64_t, hipEvent_t>> scrub_events_; // (ticket, recorded behind its launch), ascending; under scrub_mu_
  std::vector<hipEvent_t> scrub_free_events_;
  void retire_scrubs_locked(uint64_t through); // entries <= through are over
  struct Arena {
    char *base;
    size_t size, used;
  };
  std::mutex arena_mu_;
  std::vector<Arena> arenas_;
  std::map<size_t, std::vector<uint64_t>> alias_free_; // by size: extents come in a handful of sizes
  std::vector<std::pair<uint64_t, size_t>> alias_limbo_; // alias addresses whose mapping is gone but not yet invalidated (arena_mu_)
  struct ZeroExtent {
    phys_handle_t h;
    size_t pages;
    uint64_t alias;
  };
  std::mutex zero_mu_;
  std::map<size_t, ZeroExtent> zero_extents_; // by page size
  std::vector<Timed> inflight_;
  std::vector<std::pair<hipEvent_t, hipEvent_t>> free_events_;
  // unique_block_ids scratch (grow-only; bitmap all-zero and header reset between calls)
  std::mutex uniq_mu_;
  unsigned *uniq_bitmap_ = nullptr;
  size_t uniq_words_ = 0;
  void *uniq_header_ = nullptr;
  int64_t *uniq_result_ = nullptr; // pinned host memory the sweep kernel writes into
  size_t uniq_result_cap_ = 0;     // entries, including the count
  void reset_unique_scratch();
};

// One contiguous VA reservation cut into fixed-size slots (the reference's FTensor).
struct KvRegion {
  std::string name;
  char *base = nullptr;
  size_t size = 0;
  size_t page_size = 0;
  bool on_gpu = false;
  bool backfilled = false;             // every unbacked slot currently aliases a zero page
  std::vector<phys_handle_t> zero;     // zero pages of this region; slot i aliases zero[i / fanout]
  size_t fanout = 1;
  phys_handle_t zero_of(size_t slot) const { return zero[slot / fanout]; }
  // drm backend ("zero extent"): unbacked slots alias the pages of ONE multi-page buffer of zeros that the GPU context
  // owns - slot i shows its page i % zx_pages - mapped through DRM a whole group of zx_pages slots per ioctl; a run of
  // slots goes alias -> pages and pages -> alias with one REPLACE each
  bool zx = false;
  phys_handle_t zx_handle = 0;
  size_t zx_pages = 0;
  // drm backend, the rest state of unbacked slots of a compat region (a lazy one only with KVCACHED_PRT=true): a PRT
  // mapping with no buffer (DrmVm::map_prt) - reads return 0, writes are dropped, nothing faults. The TLBs cache such
  // an entry once it has been looked at, so a map batch that replaces one invalidates before its pages are used (§4.2).
  // `backfilled` says whether an unmap invalidates inside the call (compat) or behind it (lazy).
  bool prt = false;
  bool rest_direct() const { return zx || prt; }                       // unbacked slots carry a DRM mapping of ours
  // a REPLACE back to the rest state must not cross a multiple of this: the zero extent's pages repeat with that period,
  // and PRT mappings are kept to groups so that what a split leaves to be rewritten (DrmVm::refresh_prt_remainders) is small
  static constexpr size_t kPrtGroupSlots = 64;
  size_t rest_group() const { return zx ? zx_pages : (prt ? kPrtGroupSlots : (size_t)-1); }
  std::vector<phys_handle_t> handle;   // per slot, valid when mapped[slot]
  std::vector<uint64_t> seq;           // per slot: creation order of that handle (release oldest first)
  std::vector<uint64_t> stale_epoch;   // per slot: the TLB invalidation (GpuContext::next_flush_epoch) that covers its last unmap
  // hybrid/drm backends: slots HIP has been told about (placeholder mapping made and removed again). Registration
  // happens in units of `reg_group` consecutive slots (64 MiB of VA per hipMemMap; the slots behind the last full
  // group one by one), and the placeholder handles are HIP handles of the unit's size, one per 4096 units so that
  // HIP's per-handle bookkeeping stays small.
  std::vector<uint8_t> registered;
  size_t reg_group = 1;
  std::vector<hipMemGenericAllocationHandle_t> shell, shell_group;
  bool in_full_group(size_t slot) const { return reg_group > 1 && slot / reg_group < num_slots() / reg_group; }
  std::vector<uint64_t> mark;          // one bit per slot, all clear at rest: scratch of RunScan (kv_allocator.cpp), under the allocator's lock
  std::vector<uint8_t> mapped;         // per slot: 0 = unbacked, 1 = backed by its own page, 2 = by an imported page,
                                       // 3 = released by the caller, physical unmap still queued (async unmap),
                                       // 4 = backed by a page of a LANE (handle = the lane's: one per page id, shared by all its rows),
                                       // 5 = by a page of a peer's lane (handle = the imported buffer's, shared by all rows of the page id)
  size_t num_slots() const { return size / page_size; }
};

class KvAllocator {
public:
  // process-wide registry keyed by group id (reference: FTensorAllocator multiton, allocator.cpp:18-22,72-119)
  static void init(const std::string &dev_str, size_t page_size, bool contiguous_layout);
  static void shutdown();
  static KvAllocator *global(int64_t group_id);
  static bool initialized();
  static DeviceSpec device();
  static GpuContext *gpu(); // context of the init device (nullptr on "cpu")
  static size_t page_size();

  KvAllocator(DeviceSpec dev, bool contiguous_layout, GpuContext *ctx);
  ~KvAllocator();

  struct TensorDesc {
    void *ptr;
    size_t nbytes;
  };
  std::vector<TensorDesc> create_kv_tensors(size_t size, size_t dtype_size, const std::string &dev_str,
                                            int64_t num_layers, int64_t num_kv_buffers, bool unified_pool);
  bool kv_tensors_created();
  bool map_to_kv_tensors(const offset_t *offsets, size_t n);
  bool unmap_from_kv_tensors(const offset_t *offsets, size_t n);
  std::vector<void *> region_bases(); // layer-major, K then V (compact_blocks' region table)
  bool uses_prt();                    // unbacked slots of this group's regions are PRT mappings
  size_t lanes_per_extent();          // page ids are backed by lanes: lanes a buffer holds at most (0: per-slot pages)

  // async unmap (KVC_OPT_ASYNC_UNMAP): wait until every queued unmap of this allocator / of all allocators has
  // been carried out; bytes still queued (they count as free for this process)
  void flush_unmaps();
  static void flush_all_unmaps();
  // every allocator's lock, taken (true) / released (false) by the calling thread: no page-table update in between
  static void quiesce_all(bool on);
  static size_t pending_unmap_bytes();

  // TP shared pool
  int export_mapped_slots(const offset_t *offsets, size_t n, int *out_fds, int64_t cap);
  bool map_imported_slots(const offset_t *offsets, size_t n, const int *fds, size_t n_fds);
  // ... with page ids as units (lanes): one fd per page id + (lanes in the buffer, lane index) per page id
  int export_page_ids(const offset_t *offsets, size_t n, int *out_fds, int64_t *out_meta, int64_t cap);
  bool map_imported_page_ids(const offset_t *offsets, size_t n, const int *fds, size_t n_fds, const int64_t *meta);

private:
  struct Slot {
    KvRegion *region;
    size_t index;
  };
  std::vector<Slot> slots_for(const offset_t *offsets, size_t n);
  std::unique_ptr<KvRegion> make_region(const std::string &name, size_t size, size_t page_size);
  void destroy_region(KvRegion &r);
  void backfill_all(KvRegion &r);
  bool prt_all(KvRegion &r, bool by_default);      // PRT behind every slot of the region (false: not available / not wanted)
  int rest_replace(KvRegion &r, size_t first, size_t n); // slots [first, first+n) back to their rest state in one ioctl, whatever is mapped there
  int rest_map(KvRegion &r, size_t first, size_t n);     // the same over slots that hold nothing
  void register_slot(KvRegion &r, size_t slot);   // hybrid backend: make HIP aware of the slot's VA (once per slot)
  void unregister_slots(KvRegion &r);             // ... and take that back before the VA range is freed
  // `imported`: a peer's pages, one per slot, instead of pages from the pool; imported_consumed[i] is set for every
  // handle a slot took ownership of (the caller releases the others)
  void map_slots(const std::vector<Slot> &slots, const std::vector<phys_handle_t> *imported,
                 std::vector<uint8_t> *imported_consumed = nullptr);
  void unmap_slots(const std::vector<Slot> &slots);
  // the two halves of unmap_slots: driver unmaps under mu_ (handles collected), then invalidate + give handles back
  struct Unmapped {
    std::vector<Phys> own;
    std::vector<phys_handle_t> imported;
    bool any_backfilled = false;
    uint64_t epoch = 0; // the invalidation that covers these unmaps (GpuContext::next_flush_epoch after the driver calls)
    int64_t n = 0;
    size_t page_size = 0;
  };
  void unmap_collect(const std::vector<Slot> &slots, Unmapped &out);
  void unmap_finish(Unmapped &u, bool may_defer_shootdown);
  void reclaimer_loop();
  // ---- lanes (kv_allocator.cpp, "lanes"): page ids of a multi-row geometry backed, scrubbed and released as units
  struct Row {
    KvRegion *r;
    size_t first; // slot of page id p in this row = first + p
  };
  void setup_lanes();                                      // after the regions exist: decide, build the row table
  bool try_map_lanes(const offset_t *offsets, size_t n);   // false: nothing done, the generic path takes the call
  size_t unmap_lanes(const offset_t *offsets, size_t n, std::vector<offset_t> *others); // lanes unmapped; offsets that are not lane-backed -> others
  void cold_start_reserve(ExtentPool *pool);
  bool steal_pending(size_t page_size, Phys *out);
  void lock_foreground(std::unique_lock<std::mutex> &lk); // mu_ with priority over the reclaimer

  DeviceSpec dev_;
  bool contiguous_;
  GpuContext *ctx_; // owned by the registry, outlives every allocator; nullptr on "cpu"
  bool unified_pool_ = false;
  bool exportable_ = false;
  int64_t num_layers_ = 0;
  int64_t num_kv_buffers_ = 2;
  size_t tensor_bytes_per_layer_ = 0;
  std::mutex mu_;
  std::vector<std::unique_ptr<KvRegion>> layers_; // per-layer regions, or ONE region in contiguous layout
  bool lanes_ = false;           // page ids are backed by lanes
  std::vector<Row> rows_;        // layer-major, K then V: the order slots_for() lists the slots of one offset in
  size_t ids_per_row_ = 0;       // page ids a row has room for
  ExtentPool *lane_pool_ = nullptr;
  std::unordered_map<phys_handle_t, uint32_t> peer_refs_; // imported buffers -> page ids of ours they back (released at 0)
  bool reserve_built_ = false;   // the cold-start instalments of the reserve are over (cold_start_reserve)
  // async unmap state (guarded by mu_)
  std::deque<Slot> pending_;
  std::condition_variable pending_cv_, drained_cv_;
  std::thread reclaimer_;
  bool reclaimer_stop_ = false, reclaimer_busy_ = false;
  std::atomic<int> foreground_waiting_{0};
};

// hipMemGetInfo of the init device, or the test override.
void mem_get_info(size_t *free_b, size_t *total_b);
void set_mem_info_override(size_t free_b, size_t total_b);
void device_synchronize();
int64_t import_count(bool direct); // shared pool: a peer's pages imported straight into KFD + DRM / through the runtime (ROCr, HIP)

} // namespace kvc
