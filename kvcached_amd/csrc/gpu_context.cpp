// gpu_context.cpp — per-device state of the allocator: streams, kernel launches and their timing, the physical pools
// (ExtentPool) with their housekeeping, the alias arena and the scrub queue (pages zeroed on their way back),
// the zero extent of compat mode (§4.2), and the TLB invalidation with its epochs (§4.3). Regions, the map/unmap hot path
// and the registry live in kv_allocator.cpp.
#include "kv_allocator.hpp"

#include <algorithm>

#include "kernels.hpp"

namespace kvc {

// ------------------------------------------------------------------ globals
Options &options() {
  static Options o;
  return o;
}
Stats &stats() {
  static Stats s;
  return s;
}
void Stats::reset() {
  pages_mapped = pages_unmapped = 0;
  map_calls = unmap_calls = map_ns = unmap_ns = 0;
  fill_launches = fill_bytes = compact_launches = compact_bytes = 0;
  tlb_shootdowns = shootdown_ns = 0;
  index_launches = 0;
  unmaps_queued = unmaps_cancelled = 0;
  pages_scrubbed = pages_prescrubbed = 0;
  t_unmap_alias = t_acquire = t_map = t_access = t_unmap = t_release = t_realias = t_sync = 0;
  vmm.created = vmm.released = vmm.reused = 0;
  for (auto &x : seg) x = 0;
  std::lock_guard<std::mutex> g(mu);
  fill_ms = compact_ms = 0;
}

// ------------------------------------------------------------------ GpuContext
namespace {
thread_local bool tl_background_thread = false; // set in the flusher / housekeeping threads
}
std::atomic<int64_t> &background_shootdowns() {
  static std::atomic<int64_t> v{0};
  return v;
}

GpuContext::GpuContext(int dev) : dev_(dev) {
  HIP_CHECK(hipSetDevice(dev_));
  int supports_vmm = 0;
  HIP_CHECK(hipDeviceGetAttribute(&supports_vmm, hipDeviceAttributeVirtualMemoryManagementSupported, dev_));
  if (!supports_vmm)
    throw InvalidError("VMM is not supported on HIP device " + std::to_string(dev_) +
                       ". kvcached requires GPU VMM support.");
  auto prop = make_alloc_prop(dev_, false);
  size_t gran = 0;
  HIP_CHECK(hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityMinimum));
  if (gran == 0 || KvAllocator::page_size() % gran != 0)
    throw InvalidError("Invalid page size: " + std::to_string(KvAllocator::page_size()) + " must be a multiple of HIP granularity " +
                       std::to_string(gran));
  HIP_CHECK(hipStreamCreateWithFlags(&stream_, hipStreamNonBlocking));
  HIP_CHECK(hipStreamCreateWithFlags(&scrub_stream_, hipStreamNonBlocking));
  if (env_bool("KVCACHED_KFD_TLB_FLUSH", true)) {
    std::string why;
    if (!kfd_flush_.open(dev_, &why))
      KVC_LOG(LOG_WARNING, "direct KFD TLB flush unavailable (%s): invalidating through hipMalloc + hipFree", why.c_str());
  }
}

GpuContext::~GpuContext() {
  {
    std::lock_guard<std::mutex> g(fl_mu_);
    fl_stop_ = true;
  }
  fl_cv_.notify_all();
  if (flusher_.joinable()) flusher_.join();
  (void)hipSetDevice(dev_);
  for (auto &t : inflight_) {
    (void)hipEventDestroy(t.a);
    (void)hipEventDestroy(t.b);
  }
  for (auto &e : free_events_) {
    (void)hipEventDestroy(e.first);
    (void)hipEventDestroy(e.second);
  }
  try {
    flush_limbo(); // parked pages go home before their pools go
  } catch (...) {
    (void)hipGetLastError();
  }
  if (scrub_stream_) (void)hipStreamSynchronize(scrub_stream_); // no fill may be running on an alias that is about to go
  primary_pool_.store(nullptr);
  extent_pools_[0].clear(); // idle extents go back to the driver (after the invalidation they may still be owed)
  extent_pools_[1].clear();
  lane_pools_.clear();
  for (auto &kv : zero_extents_) {
    (void)DrmVm::instance().clear(reinterpret_cast<void *>(kv.second.alias), kv.second.pages * kv.first);
    (void)DrmVm::instance().forget(kv.second.h);
  }
  zero_extents_.clear();
  for (auto &a : arenas_) (void)hipMemAddressFree(a.base, a.size);
  arenas_.clear();
  if (fallback_block_) (void)hipFree(fallback_block_);
  if (scrub_stream_) (void)hipStreamDestroy(scrub_stream_);
  for (auto &e : scrub_events_) (void)hipEventDestroy(e.second);
  for (auto ev : scrub_free_events_) (void)hipEventDestroy(ev);
  kfd_flush_.close();
  if (uniq_bitmap_) (void)hipFree(uniq_bitmap_);
  if (uniq_header_) (void)hipFree(uniq_header_);
  if (uniq_result_) (void)hipHostFree(uniq_result_);
  if (stream_) (void)hipStreamDestroy(stream_);
}

void GpuContext::bind() const { HIP_CHECK(hipSetDevice(dev_)); }

namespace {
// hipMemGetInfo is ~0.25 us on MI355X: cheap enough to ask on every release batch.
bool device_under_pressure() {
  size_t free_b = 0, total_b = 0;
  if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) {
    (void)hipGetLastError();
    return false;
  }
  static const double util = []() {
    const char *e = std::getenv("KVCACHED_GPU_UTILIZATION");
    return e ? std::atof(e) : 0.95;
  }();
  return free_b < static_cast<size_t>(total_b * (1.0 - util));
}
} // namespace

// What a pool creates and releases its buffers with. `unit_bytes`: bytes of one unit of the pool (a page; a lane).
ExtentDriver GpuContext::make_driver(size_t unit_bytes, bool exportable, bool aliases) {
  ExtentDriver d;
  const int dev = dev_;
  d.create = [this, dev, unit_bytes, exportable, aliases](size_t units, uint64_t *tag) -> phys_handle_t {
    if (fail_after_creates().load() >= 0 && fail_after_creates().fetch_sub(1) == 0) // fault injection (tests)
      hip_check(hipErrorOutOfMemory, "hipMemCreate(&p.h, granule_, &prop, 0) [injected]", __FILE__, __LINE__);
    const phys_handle_t h = vmm_create(dev, units * unit_bytes, exportable, true, (unsigned)units);
    if (tag) *tag = 0;
    if (tag && aliases) {
      // The buffer's second, permanent mapping: the address its pages are zeroed through after their slots are gone.
      // One more ioctl per buffer; failure only means that its pages are zeroed when they are mapped, as before.
      if (void *bo = DrmVm::instance().find(h)) {
        const uint64_t va = alias_alloc(units * unit_bytes);
        if (va && DrmVm::instance().map(bo, reinterpret_cast<void *>(va), units * unit_bytes, 0) == 0)
          *tag = va;
        else if (va)
          alias_free(va, units * unit_bytes);
      }
    }
    return h;
  };
  d.prepare_release = [this, unit_bytes](phys_handle_t, uint64_t tag, size_t units) {
    if (!tag) return;
    wait_all_scrubs(); // nothing may still be writing through the alias
    StaleAfter mark;   // a live translation goes away: the invalidation before_release performs covers it
    if (DrmVm::instance().clear(reinterpret_cast<void *>(tag), units * unit_bytes) != 0)
      KVC_LOG(LOG_ERROR, "dropping the alias mapping of a buffer failed");
    // the address is on offer again only AFTER the invalidation below: a new buffer mapped there earlier could have its
    // scrub go through a cached translation of the old one
    std::lock_guard<std::mutex> g(arena_mu_);
    alias_limbo_.emplace_back(tag, units * unit_bytes);
  };
  d.release = [](phys_handle_t h) {
    const bool ok = vmm_try_release(h);
    if (!ok) KVC_LOG(LOG_ERROR, "releasing a physical handle failed");
    (void)hipGetLastError();
    return ok;
  };
  d.before_release = [this]() { // memory leaves the process: no translation of it may survive
    std::vector<std::pair<uint64_t, size_t>> mine;
    {
      std::lock_guard<std::mutex> g(arena_mu_);
      mine.swap(alias_limbo_); // (what another thread adds from here on is covered by ITS invalidation, not by this one)
    }
    flush_deferred_shootdown(); // (an invalidation that failed has raised the flag again: do_shootdown)
    for (auto &a : mine) alias_free(a.first, a.second);
  };
  d.under_pressure = device_under_pressure;
  return d;
}

ExtentPool *GpuContext::extents(size_t page_bytes, bool exportable) {
  // Run-sized extents need a map offset (GEM_VA has one; HIP rejects it, ROCr ignores it), ranged unmaps and buffers of
  // our own making: the drm backend with pages straight from KFD. Everything else works with single pages.
  unsigned k = 1;
  if (!exportable && vmm_backend() == kVmmDrm && DrmVm::instance().kfd_ready() && DrmVm::instance().can_clear())
    k = (unsigned)std::min<int64_t>(kMaxExtentPages, std::max<int64_t>(1, options().phys_chunk_pages.load()));
  k = (unsigned)std::min<size_t>(k, std::max<size_t>(1, (256u << 20) / page_bytes)); // no buffer above 256 MiB (8 MiB pages, compound pages)
  std::unique_ptr<ExtentPool> old; // destroyed (drained) after mu_ is released: ~ExtentPool waits for scrubs, whose harvest takes mu_
  ExtentPool *p;
  {
    std::lock_guard<std::mutex> g(mu_);
    auto &m = extent_pools_[exportable ? 1 : 0];
    auto it = m.find(page_bytes);
    if (it == m.end() || it->second->max_extent_pages() != k) {
      // (re)made at the first use after an init that changed the extent size: no region exists then, every piece is home
      if (housekeepers_.load() > 0 && it != m.end())
        throw InvalidError("the extent size cannot change while a PageAllocator's housekeeping thread is alive: delete the PageAllocator first");
      const bool aliases = k > 1 && options().scrub_on_release.load() != 0; // (k > 1: drm backend, buffers of our own making)
      if (it != m.end()) {
        if (primary_pool_.load() == it->second.get()) primary_pool_.store(nullptr);
        old = std::move(it->second);
      }
      m[page_bytes] = std::make_unique<ExtentPool>(page_bytes, k, make_driver(page_bytes, exportable, aliases), &stats().vmm);
      it = m.find(page_bytes);
      it->second->set_defer_eviction(housekeepers_.load() > 0);
    }
    p = it->second.get();
  }
  if (old) {
    flush_limbo(); // (a parked finisher holds a pointer to its pool)
    old.reset();
  }
  p->set_cap_bytes((size_t)std::max<int64_t>(0, options().pool_bytes.load()));
  p->set_waste_frac((double)std::max<int64_t>(0, options().extent_waste_pct.load()) / 100.0);
  return p;
}

ExtentPool *GpuContext::lane_extents(size_t rows, size_t page_bytes) {
  if (rows < 2 || vmm_backend() != kVmmDrm || !DrmVm::instance().kfd_ready() || !DrmVm::instance().can_clear()) return nullptr;
  const size_t lane_bytes = rows * page_bytes;
  // Lanes per buffer: KVCACHED_LANES_PER_BUFFER (default 8), as far as they fit KVCACHED_LANE_EXTENT_MB (default 1024: 8 page ids
  // of the Llama-3-8B geometry, 128 MiB each). A buffer only goes back to the driver whole, so a page id that stays mapped - the
  // PageAllocator keeps up to 10 freed ids mapped for reuse - pins the other lanes of its buffer: 8 lanes keep that to
  // 7 lanes per straggler (64-lane buffers of a small geometry pinned 1 GiB each in tests/test_gpu_colocation.py), and
  // beyond 8 the gain is small (the ioctl count per call is rows x ceil(ids / lanes): 64 for 8 page ids either way).
  const size_t cap_b = (size_t)std::max<int64_t>(1, env_i64("KVCACHED_LANE_EXTENT_MB", 1024)) << 20;
  const size_t want_k = (size_t)std::min<int64_t>(kMaxExtentPages, std::max<int64_t>(1, env_i64("KVCACHED_LANES_PER_BUFFER", 8)));
  const unsigned k = (unsigned)std::min<size_t>(want_k, std::max<size_t>(1, cap_b / lane_bytes));
  std::unique_ptr<ExtentPool> old;
  ExtentPool *p;
  {
    std::lock_guard<std::mutex> g(mu_);
    auto &slot = lane_pools_[{rows, page_bytes}];
    if (!slot || slot->max_extent_pages() != k) {
      if (housekeepers_.load() > 0 && slot)
        throw InvalidError("the lane extent size cannot change while a PageAllocator's housekeeping thread is alive");
      if (slot && primary_pool_.load() == slot.get()) primary_pool_.store(nullptr);
      old = std::move(slot);
      slot = std::make_unique<ExtentPool>(lane_bytes, k, make_driver(lane_bytes, false, options().scrub_on_release.load() != 0), &stats().vmm, rows);
      slot->set_defer_eviction(housekeepers_.load() > 0);
    }
    p = slot.get();
  }
  if (old) {
    flush_limbo();
    old.reset();
  }
  p->set_cap_bytes((size_t)std::max<int64_t>(0, options().pool_bytes.load()));
  p->set_waste_frac((double)std::max<int64_t>(0, options().extent_waste_pct.load()) / 100.0);
  return p;
}

std::vector<ExtentPool *> GpuContext::all_pools() {
  std::lock_guard<std::mutex> g(mu_);
  std::vector<ExtentPool *> ps;
  for (auto &m : extent_pools_)
    for (auto &kv : m) ps.push_back(kv.second.get());
  for (auto &kv : lane_pools_)
    if (kv.second) ps.push_back(kv.second.get());
  return ps;
}

void GpuContext::drain_pools() {
  try {
    flush_limbo();
  } catch (...) {
    (void)hipGetLastError();
  }
  for (auto *p : all_pools()) p->drain(0);
}

size_t GpuContext::idle_pool_bytes() {
  size_t b = limbo_bytes_.load(); // (unmapped, on their way back to their pool: ours to reuse a moment from now)
  for (auto *p : all_pools()) b += p->idle_bytes();
  return b;
}

void GpuContext::add_housekeeper(int delta) {
  const bool on = housekeepers_.fetch_add(delta) + delta > 0;
  for (auto *p : all_pools()) {
    p->set_defer_eviction(on);
    if (!on) p->trim_to_cap((size_t)-1); // nobody will do it later
  }
}

void GpuContext::housekeeping() {
  std::unique_lock<std::mutex> one(hk_mu_, std::try_to_lock); // several PageAllocators' watchers may tick at once: one of them does the work
  if (!one.owns_lock()) return;
  auto ps = all_pools();
  if (ps.empty()) return;
  (void)hipSetDevice(dev_);
  tl_background_thread = true; // the allocator's watcher thread
  try {
    flush_deferred_shootdown(); // the invalidation unmap batches left to this thread (KVCACHED_ASYNC_SHOOTDOWN)
  } catch (const std::exception &e) {
    KVC_LOG(LOG_ERROR, "housekeeping: TLB invalidation failed: %s", e.what());
  }
  if (device_under_pressure()) {
    for (auto *p : ps) p->drain(0);
    return;
  }
  // Releasing memory the GPU has touched costs 50-70 us per 2 MiB (the kernel wipes it) and driver calls of one
  // process do not overlap: 256 pages per 100 ms tick keeps this thread's share of the driver under ~15 % while
  // still returning 5 GiB/s.
  constexpr size_t kPerTickBytes = 512u << 20; // (a pool's unit may be a lane of 128 MiB: the budget is in bytes)
  auto per_tick = [&](ExtentPool *p) { return std::max<size_t>(1, kPerTickBytes / p->page_bytes()); };
  for (auto *p : ps) p->trim_to_cap(per_tick(p)); // what release_batch left above the cap (deferred eviction)
  const int64_t idle_ms = options().pool_idle_ms.load();
  // the reserve never exceeds the pool's cap (KVCACHED_PHYS_POOL_MB=0, "pool off", means no reserve either: what would be
  // created here would be trimmed at the next tick, for ever)
  size_t reserve_b = reserve_target_bytes();
  limbo_peak_[(limbo_tick_.fetch_add(1) + 1) % kLimboTicks].store(limbo_bytes_.load()); // (the slot of the tick that starts now)
  ExtentPool *primary = primary_pool_.load();
  // An engine that has not mapped or unmapped anything for KVCACHED_RESERVE_IDLE_S (default 10 s) gives its reserve back as well:
  // idle memory is what a co-located engine could use (the reference releases every page on unmap, csrc/page.cpp:17). The
  // reserve comes back with the first ticks after the next call. (Not "nothing mapped": an engine's prealloc thread keeps a few
  // page ids mapped at all times.)
  if (primary) {
    const int64_t last = fg_last_ns_.load();
    if (fg_active_.load() == 0 && last != 0 && now_ns() - last >= std::max<int64_t>(1, env_i64("KVCACHED_RESERVE_IDLE_S", 10)) * 1000000000ll) reserve_b = 0;
  }
  // The reserve follows demand. What a map call has to CREATE it pays for on its caller's thread - 2 us per
  // buffer on VRAM the kernel has wiped, but ~50-80 us per 2 MiB on VRAM it has not handed out yet (cleared inside the
  // allocation at 30-40 GB/s: profiles/r03_bench_n1.json growth_burst_first_touch, kfd_alloc 3.3 ms per 128 MiB). So when
  // callers had to create memory during the last second, this thread creates AHEAD of them: up to twice that amount idle
  // (never above the pool's cap or KVCACHED_PHYS_RESERVE_MAX_MB, default 16384), at up to 2 GiB per tick - ~20 GB/s, two
  // thirds of what the kernel can clear - so that growth that goes on finds cleared memory. It is a target for creating,
  // not a licence to keep: the decay floor stays the base reserve, so a second after the growth has stopped the target is
  // the base again and what was made ahead and not used goes back like any idle memory (KVCACHED_POOL_IDLE_MS) - a
  // co-located engine sees it return (tests/test_gpu_colocation.py).
  size_t boost_units = 0, tick_units = 0;
  if (primary) {
    const size_t now_units = primary->demand_units();
    if (demand_pool_ != primary) { // (another pool took over: start afresh)
      demand_pool_ = primary;
      demand_seen_ = now_units;
      for (auto &d : demand_window_) d = 0;
      demand_quiet_ticks_ = 0;
      reserve_boost_units_ = 0;
    }
    const size_t delta = now_units - demand_seen_;
    demand_seen_ = now_units;
    demand_window_[demand_tick_++ % kDemandTicks] = delta;
    size_t last_second = 0;
    for (auto d : demand_window_) last_second += d;
    demand_quiet_ticks_ = delta ? 0 : demand_quiet_ticks_ + 1;
    const size_t max_units = std::min((size_t)std::max<int64_t>(0, options().pool_bytes.load()),
                                      (size_t)std::max<int64_t>(0, env_i64("KVCACHED_PHYS_RESERVE_MAX_MB", 16384)) << 20) / primary->page_bytes();
    if (env_bool("KVCACHED_ADAPTIVE_RESERVE", true) && reserve_b > 0) {
      reserve_boost_units_ = std::min(max_units, 2 * last_second); // (0 once a whole second has passed without a creation on anybody's path)
    } else {
      reserve_boost_units_ = 0;
    }
    boost_units = reserve_boost_units_;
    tick_units = boost_units ? std::max<size_t>(1, (2048ull << 20) / primary->page_bytes()) : 0;
  }
  for (auto *p : ps) {
    // the reserve belongs to the pool the engine's own pages come from (recorded at its first map call), not to the
    // exportable twin or to a pool a diagnostic happened to create
    const size_t base_pages = p == primary ? reserve_b / p->page_bytes() : 0;
    const size_t floor_pages = std::max(base_pages, p == primary ? boost_units : 0);
    if (idle_ms > 0 && !(p == primary && boost_units)) p->decay(now_ns(), idle_ms * 1000000ll, per_tick(p), base_pages);
    // Growth into VRAM the kernel has not cleared yet costs ~80 us per 2 MiB inside the allocation (one SDMA ring,
    // ~30 GB/s: profiles/r02_create_cost.jsonl); this thread pays that ahead of time, off every caller's path.
    if (floor_pages) p->refill_reserve(floor_pages, std::max(per_tick(p), p == primary ? tick_units : 0));
  }
}

void GpuContext::begin_timed(hipStream_t s, int kind) {
  if (!options().profile.load()) return;
  std::lock_guard<std::mutex> g(mu_);
  Timed t{nullptr, nullptr, kind};
  if (!free_events_.empty()) {
    t.a = free_events_.back().first;
    t.b = free_events_.back().second;
    free_events_.pop_back();
  } else {
    HIP_CHECK(hipEventCreate(&t.a));
    HIP_CHECK(hipEventCreate(&t.b));
  }
  HIP_CHECK(hipEventRecord(t.a, s));
  inflight_.push_back(t);
}
void GpuContext::end_timed(hipStream_t s) {
  if (!options().profile.load()) return;
  std::lock_guard<std::mutex> g(mu_);
  if (!inflight_.empty()) HIP_CHECK(hipEventRecord(inflight_.back().b, s));
}
void GpuContext::harvest() {
  std::lock_guard<std::mutex> g(mu_);
  if (inflight_.empty()) return;
  double fill = 0, comp = 0;
  std::vector<Timed> still;
  for (auto &t : inflight_) {
    if (hipEventQuery(t.b) == hipErrorNotReady) { // a launch on the other stream that has not finished: next time
      (void)hipGetLastError();
      still.push_back(t);
      continue;
    }
    float ms = 0;
    if (hipEventElapsedTime(&ms, t.a, t.b) == hipSuccess) (t.kind == 0 ? fill : comp) += ms;
    (void)hipGetLastError();
    free_events_.emplace_back(t.a, t.b);
  }
  inflight_.swap(still);
  std::lock_guard<std::mutex> g2(stats().mu);
  stats().fill_ms += fill;
  stats().compact_ms += comp;
}

void GpuContext::zero_fill(void *const *pages, size_t n, size_t page_bytes, hipStream_t s) {
  if (!s) s = stream_;
  const int variant = (int)options().fill_variant.load();
  for (size_t i = 0; i < n; i += kMaxPtrsPerLaunch) {
    int k = (int)std::min<size_t>(kMaxPtrsPerLaunch, n - i);
    begin_timed(s, 0);
    HIP_CHECK(launch_zero_fill_pages(pages + i, k, page_bytes, s, variant));
    end_timed(s);
    stats().fill_launches++;
    stats().fill_bytes += (int64_t)k * (int64_t)page_bytes;
  }
}

uint64_t GpuContext::scrub(const uint64_t *alias_addrs, size_t n, size_t page_bytes) {
  std::vector<void *> ptrs;
  ptrs.reserve(n);
  for (size_t i = 0; i < n; ++i)
    if (alias_addrs[i]) ptrs.push_back(reinterpret_cast<void *>(alias_addrs[i]));
  if (ptrs.empty()) return 0;
  std::lock_guard<std::mutex> g(scrub_mu_);
  bind();
  zero_fill(ptrs.data(), ptrs.size(), page_bytes, scrub_stream_);
  stats().pages_scrubbed += (int64_t)ptrs.size();
  const uint64_t ticket = scrub_issued_.fetch_add(1) + 1;
  // an event behind every scrub: whoever gets these pages next waits for THIS scrub, not for the ones queued after it
  // (the pool hands out the pages that have been idle longest: as a rule their scrub is long over and the wait is a query)
  while (!scrub_events_.empty() && hipEventQuery(scrub_events_.front().second) == hipSuccess) retire_scrubs_locked(scrub_events_.front().first);
  (void)hipGetLastError();
  hipEvent_t ev = nullptr;
  if (!scrub_free_events_.empty()) {
    ev = scrub_free_events_.back();
    scrub_free_events_.pop_back();
  } else if (hipEventCreateWithFlags(&ev, hipEventDisableTiming) != hipSuccess) {
    (void)hipGetLastError();
    ev = nullptr;
  }
  if (ev && hipEventRecord(ev, scrub_stream_) == hipSuccess) {
    scrub_events_.emplace_back(ticket, ev);
  } else { // no event: wait_scrub falls back to the whole stream
    (void)hipGetLastError();
    if (ev) scrub_free_events_.push_back(ev);
  }
  return ticket;
}

void GpuContext::retire_scrubs_locked(uint64_t through) {
  while (!scrub_events_.empty() && scrub_events_.front().first <= through) {
    scrub_free_events_.push_back(scrub_events_.front().second);
    scrub_events_.pop_front();
  }
  uint64_t cur = scrub_done_.load();
  while (cur < through && !scrub_done_.compare_exchange_weak(cur, through)) {
  }
}

void GpuContext::wait_scrub(uint64_t ticket) {
  if (!ticket || scrub_done_.load() >= ticket) return;
  hipEvent_t ev = nullptr;
  uint64_t covers = 0;
  {
    std::lock_guard<std::mutex> g(scrub_mu_);
    if (scrub_done_.load() >= ticket) return;
    for (auto &e : scrub_events_) // the first scrub at or behind `ticket` that still has its event (stream order: it covers `ticket`)
      if (e.first >= ticket) {
        ev = e.second;
        covers = e.first; // once it has fired, everything up to it is over
        break;
      }
  }
  static const bool whole_stream = KVC_TEST_HOOK("SCRUB_WAIT_STREAM"); // diagnostics: the previous behaviour
  if (whole_stream) ev = nullptr;
  if (ev) {
    HIP_CHECK(hipEventSynchronize(ev)); // (should the event be recycled meanwhile, this waits for a later scrub: longer, never shorter)
    std::lock_guard<std::mutex> g(scrub_mu_);
    retire_scrubs_locked(covers);
  } else { // no event behind it (creation failed, or a ticket from before the events): the whole stream
    const uint64_t covered = scrub_issued_.load(); // every scrub launched so far is ahead of the sync below
    HIP_CHECK(hipStreamSynchronize(scrub_stream_));
    std::lock_guard<std::mutex> g(scrub_mu_);
    retire_scrubs_locked(covered);
  }
  harvest();
}

phys_handle_t GpuContext::zero_extent(size_t page_bytes, size_t *pages) {
  std::lock_guard<std::mutex> g(zero_mu_); // (not mu_: the fill below takes that one when launches are being timed)
  auto it = zero_extents_.find(page_bytes);
  if (it != zero_extents_.end()) {
    *pages = it->second.pages;
    return it->second.h;
  }
  if (vmm_backend() != kVmmDrm || !DrmVm::instance().kfd_ready() || !DrmVm::instance().can_clear()) return 0;
  // 64 pages (128 MiB of zeros for 2 MiB pages: 0.04 % of the HBM), never more than 256 MiB
  const size_t n = std::max<size_t>(1, std::min<size_t>(kMaxExtentPages, (256u << 20) / page_bytes));
  phys_handle_t h = 0;
  uint64_t alias = 0;
  try {
    h = DrmVm::instance().create(n * page_bytes, (unsigned)n);
    void *bo = DrmVm::instance().find(h);
    alias = alias_alloc(n * page_bytes);
    if (!bo || !alias || DrmVm::instance().map(bo, reinterpret_cast<void *>(alias), n * page_bytes, 0) != 0) throw GpuError("mapping the zero extent failed");
    std::vector<void *> ptrs;
    for (size_t i = 0; i < n; ++i) ptrs.push_back(reinterpret_cast<void *>(alias + i * page_bytes));
    bind();
    zero_fill(ptrs.data(), ptrs.size(), page_bytes, stream_); // what the driver hands out is zero today; not a documented guarantee
    HIP_CHECK(hipStreamSynchronize(stream_));
  } catch (const std::exception &e) {
    KVC_LOG(LOG_WARNING, "no zero extent (%s): compat mode aliases sharded zero pages through ROCr", e.what());
    if (alias) alias_free(alias, n * page_bytes);
    if (h) (void)DrmVm::instance().forget(h);
    (void)hipGetLastError();
    return 0;
  }
  zero_extents_[page_bytes] = ZeroExtent{h, n, alias};
  *pages = n;
  return h;
}

size_t GpuContext::zero_extent_pages(size_t page_bytes) {
  std::lock_guard<std::mutex> g(zero_mu_);
  auto it = zero_extents_.find(page_bytes);
  return it == zero_extents_.end() ? 0 : it->second.pages;
}

uint64_t GpuContext::alias_alloc(size_t bytes) {
  std::lock_guard<std::mutex> g(arena_mu_);
  auto it = alias_free_.find(bytes);
  if (it != alias_free_.end() && !it->second.empty()) {
    const uint64_t va = it->second.back();
    it->second.pop_back();
    return va;
  }
  constexpr size_t kArena = 64ull << 30;
  if (arenas_.empty() || arenas_.back().size - arenas_.back().used < bytes) {
    void *p = nullptr;
    const size_t want = std::max(kArena, bytes);
    if (hipMemAddressReserve(&p, want, kBasePage, nullptr, 0) != hipSuccess) {
      (void)hipGetLastError();
      return 0;
    }
    arenas_.push_back(Arena{static_cast<char *>(p), want, 0});
  }
  Arena &a = arenas_.back();
  const uint64_t va = reinterpret_cast<uint64_t>(a.base + a.used);
  a.used += bytes;
  return va;
}

void GpuContext::alias_free(uint64_t va, size_t bytes) {
  std::lock_guard<std::mutex> g(arena_mu_);
  alias_free_[bytes].push_back(va);
}

void GpuContext::compact(void *const *bases, size_t n_regions, const int64_t *src, const int64_t *dst, size_t n_moves,
                         size_t block_bytes, hipStream_t s) {
  if (!s) s = stream_;
  const int variant = (int)options().compact_variant.load();
  for (size_t r = 0; r < n_regions; r += kMaxRegionsPerLaunch) {
    int nr = (int)std::min<size_t>(kMaxRegionsPerLaunch, n_regions - r);
    for (size_t m = 0; m < n_moves; m += kMaxMovesPerLaunch) {
      int nm = (int)std::min<size_t>(kMaxMovesPerLaunch, n_moves - m);
      begin_timed(s, 1);
      HIP_CHECK(launch_compact_blocks(bases + r, nr, src + m, dst + m, nm, block_bytes, s, variant));
      end_timed(s);
      stats().compact_launches++;
      stats().compact_bytes += 2ll * nr * nm * (int64_t)block_bytes;
    }
  }
}

// ---------------------------------------------------------------- block id <-> token index glue
namespace {
// Block ids beyond the kernarg budget go through a stream-ordered device buffer (rare: > 1024 new blocks in
// one scheduler step). hipMemcpyAsync from pageable memory returns once the source has been consumed.
struct StagedIds {
  int64_t *dev = nullptr;
  hipStream_t s;
  StagedIds(const int64_t *host, size_t n, hipStream_t stream) : s(stream) {
    HIP_CHECK(hipMallocAsync(reinterpret_cast<void **>(&dev), n * sizeof(int64_t), s));
    hipError_t st = hipMemcpyAsync(dev, host, n * sizeof(int64_t), hipMemcpyHostToDevice, s);
    if (st != hipSuccess) {
      (void)hipFreeAsync(dev, s);
      HIP_CHECK(st);
    }
  }
  ~StagedIds() {
    if (dev) (void)hipFreeAsync(dev, s);
  }
};
} // namespace

void GpuContext::expand_block_ids(const int64_t *ids, size_t n, int64_t tpb, int64_t *out, hipStream_t s) {
  // (s == NULL is the device's default stream here - what torch.cuda.current_stream() is unless the caller switched: these
  // kernels consume and produce tensors of the caller's stream, the library's own non-blocking stream would race with them)
  for (size_t i = 0; i < n; i += kMaxIdsPerLaunch) {
    const size_t k = std::min<size_t>(kMaxIdsPerLaunch, n - i);
    HIP_CHECK(launch_expand_block_ids(ids + i, nullptr, k, tpb, out + i * (size_t)tpb, s));
    stats().index_launches++;
  }
}

void GpuContext::alloc_extend_indices(const int64_t *pre_lens, const int64_t *seq_lens, const int64_t *last_loc, size_t bs,
                                      const int64_t *ids, size_t n_ids, int64_t tpb, int64_t *out, size_t out_len,
                                      hipStream_t s) {
  // (s == NULL is the device's default stream here - what torch.cuda.current_stream() is unless the caller switched: these
  // kernels consume and produce tensors of the caller's stream, the library's own non-blocking stream would race with them)
  if (n_ids <= (size_t)kMaxIdsPerLaunch) {
    HIP_CHECK(launch_alloc_extend(ids, nullptr, n_ids, pre_lens, seq_lens, last_loc, bs, tpb, out, out_len, s));
  } else {
    StagedIds staged(ids, n_ids, s);
    HIP_CHECK(launch_alloc_extend(nullptr, staged.dev, n_ids, pre_lens, seq_lens, last_loc, bs, tpb, out, out_len, s));
  }
  stats().index_launches++;
}

void GpuContext::alloc_decode_indices(const int64_t *seq_lens, const int64_t *last_loc, size_t bs, const int64_t *ids,
                                      size_t n_ids, int64_t tpb, int64_t *out, hipStream_t s) {
  // (s == NULL is the device's default stream here - what torch.cuda.current_stream() is unless the caller switched: these
  // kernels consume and produce tensors of the caller's stream, the library's own non-blocking stream would race with them)
  if (n_ids <= (size_t)kMaxIdsPerLaunch) {
    HIP_CHECK(launch_alloc_decode(ids, nullptr, n_ids, seq_lens, last_loc, bs, tpb, out, s));
  } else {
    StagedIds staged(ids, n_ids, s);
    HIP_CHECK(launch_alloc_decode(nullptr, staged.dev, n_ids, seq_lens, last_loc, bs, tpb, out, s));
  }
  stats().index_launches++;
}

void GpuContext::reset_unique_scratch() {
  static const unsigned init[4] = {0xffffffffu, 0u, 0u, 0u};
  if (uniq_bitmap_) (void)hipMemset(uniq_bitmap_, 0, uniq_words_ * sizeof(unsigned));
  if (uniq_header_) (void)hipMemcpy(uniq_header_, init, sizeof(init), hipMemcpyHostToDevice);
}

int64_t GpuContext::unique_block_ids(const int64_t *idx, size_t n, int64_t tpb, int64_t n_blocks, int64_t *out_host,
                                     size_t cap, hipStream_t s) {
  // (s == NULL is the device's default stream here - what torch.cuda.current_stream() is unless the caller switched: these
  // kernels consume and produce tensors of the caller's stream, the library's own non-blocking stream would race with them)
  if (tpb <= 0 || n_blocks <= 0) throw InvalidError("tokens_per_block and num_blocks must be positive");
  std::lock_guard<std::mutex> g(uniq_mu_);
  const size_t words = ((size_t)n_blocks + 31) / 32;
  if (words > uniq_words_) {
    if (uniq_bitmap_) HIP_CHECK(hipFree(uniq_bitmap_));
    uniq_bitmap_ = nullptr;
    uniq_words_ = 0;
    const size_t want = std::max<size_t>(words, 4096);
    HIP_CHECK(hipMalloc(reinterpret_cast<void **>(&uniq_bitmap_), want * sizeof(unsigned)));
    uniq_words_ = want;
    if (!uniq_header_) HIP_CHECK(hipMalloc(&uniq_header_, 16));
    reset_unique_scratch();
    HIP_CHECK(hipDeviceSynchronize());
  }
  const size_t need = std::min<size_t>(n, (size_t)n_blocks) + 1;
  if (need > uniq_result_cap_) {
    if (uniq_result_) HIP_CHECK(hipHostFree(uniq_result_));
    uniq_result_ = nullptr;
    uniq_result_cap_ = 0;
    const size_t want = std::max<size_t>(need, 8192);
    HIP_CHECK(hipHostMalloc(reinterpret_cast<void **>(&uniq_result_), want * sizeof(int64_t), hipHostMallocDefault));
    uniq_result_cap_ = want;
  }
  void *result_dev = nullptr;
  HIP_CHECK(hipHostGetDevicePointer(&result_dev, uniq_result_, 0));
  try {
    HIP_CHECK(launch_unique_block_ids(idx, n, tpb, n_blocks, uniq_bitmap_, uniq_header_, static_cast<int64_t *>(result_dev),
                                      uniq_result_cap_ - 1, s));
    stats().index_launches += 2;
    HIP_CHECK(hipStreamSynchronize(s));
  } catch (...) {
    (void)hipDeviceSynchronize();
    reset_unique_scratch();
    throw;
  }
  const int64_t count = uniq_result_[0];
  if (count < 0)
    throw InvalidError(std::to_string(-count) + " token indices are outside [0, num_blocks * tokens_per_block)");
  if (out_host && (size_t)count <= cap) std::copy(uniq_result_ + 1, uniq_result_ + 1 + count, out_host);
  return count;
}

void GpuContext::ensure_flushed() {
  {
    std::lock_guard<std::mutex> g(flush_mu_);
    if (tlb_stale().load()) do_shootdown();
  }
  drain_limbo(); // (outside flush_mu_: a finisher may release memory to the driver, which asks for the flush lock itself)
}

void GpuContext::request_async_flush(int64_t within_us) {
  {
    std::lock_guard<std::mutex> g(fl_mu_);
    if (fl_stop_) return;
    if (!flusher_.joinable()) flusher_ = std::thread([this] { flusher_loop(); });
    fl_kick_ = true;
    if (within_us > 0) {
      const int64_t d = now_ns() + within_us * 1000;
      fl_deadline_ns_ = fl_deadline_ns_ ? std::min(fl_deadline_ns_, d) : d;
    }
  }
  fl_cv_.notify_one();
}

void GpuContext::park(uint64_t epoch, size_t bytes, std::function<void()> finish) {
  std::lock_guard<std::mutex> g(limbo_mu_);
  limbo_.push_back(Parked{epoch, bytes, std::move(finish)});
  const size_t now = (limbo_bytes_ += bytes);
  auto &peak = limbo_peak_[limbo_tick_.load() % kLimboTicks];
  if (now > peak.load()) peak.store(now);
}

size_t GpuContext::reserve_target_bytes() const {
  const size_t cap = (size_t)std::max<int64_t>(0, options().pool_bytes.load());
  const size_t base = std::min((size_t)std::max<int64_t>(0, options().phys_reserve_bytes.load()), cap);
  if (!base) return 0;
  size_t parked = 0;
  for (auto &p : limbo_peak_) parked = std::max(parked, p.load());
  // (at most the base again: the allowance is for the NEXT map batch, and a free() of many GiB must not make the housekeeping
  // thread create as much on top of what a following alloc() takes back)
  return std::min(base + std::min(parked, base), cap);
}

void GpuContext::drain_limbo() {
  if (!limbo_bytes_.load()) return;
  for (;;) {
    Parked p;
    {
      std::lock_guard<std::mutex> g(limbo_mu_);
      if (limbo_.empty() || !flushed_through(limbo_.front().epoch)) return; // (epochs are stamped in order: the front is the oldest)
      p = std::move(limbo_.front());
      limbo_.pop_front();
    }
    try {
      p.finish();
    } catch (const std::exception &e) {
      KVC_LOG(LOG_ERROR, "giving parked pages back failed: %s", e.what());
    }
    limbo_bytes_ -= std::min(p.bytes, limbo_bytes_.load());
  }
}

void GpuContext::flush_limbo() {
  uint64_t newest = 0;
  {
    std::lock_guard<std::mutex> g(limbo_mu_);
    if (!limbo_.empty()) newest = limbo_.back().epoch;
  }
  if (newest) ensure_flushed_through(newest);
  drain_limbo();
}

void GpuContext::flusher_loop() {
  (void)hipSetDevice(dev_);
  tl_background_thread = true;
  std::unique_lock<std::mutex> lk(fl_mu_);
  while (!fl_stop_) {
    fl_cv_.wait(lk, [&] { return fl_stop_ || fl_kick_; });
    if (fl_stop_) break;
    fl_kick_ = false;
    lk.unlock();
    // The invalidation and the page-table ioctls of a map or unmap call serialise in the kernel (a REPLACE issued while
    // the KFD pair is in flight waits for most of its 0.4 ms). Nobody is waiting for THIS invalidation - whoever needs
    // one earlier performs it himself (ensure_flushed*) - so it yields to foreground calls: it starts once none has
    // been active for 150 us, or after 2 ms at the latest (an engine that frees and allocates in one scheduler step
    // gets its alloc through first; a tight loop is invalidated every few iterations instead of behind every one).
    const int64_t t0 = now_ns();
    for (;;) {
      int64_t deadline;
      {
        std::lock_guard<std::mutex> g(fl_mu_);
        deadline = fl_deadline_ns_;
      }
      const int64_t now = now_ns();
      if (!foreground_busy() || now - t0 >= 2000000 || (deadline && now >= deadline)) break;
      std::this_thread::sleep_for(std::chrono::microseconds(deadline ? 20 : 40));
    }
    {
      std::lock_guard<std::mutex> g(fl_mu_);
      fl_deadline_ns_ = 0;
    }
    try {
      ensure_flushed();
    } catch (const std::exception &e) {
      KVC_LOG(LOG_ERROR, "background TLB invalidation failed: %s", e.what());
      (void)hipGetLastError();
    }
    lk.lock();
  }
}

// Unconditional invalidation. Serialised with ensure_flushed(): the flag is cleared when an invalidation STARTS, so
// whoever finds it clear must be able to rely on that invalidation having finished - both take flush_mu_.
void GpuContext::tlb_shootdown() {
  {
    std::lock_guard<std::mutex> g(flush_mu_);
    do_shootdown();
  }
  drain_limbo();
}

void GpuContext::ensure_flushed_through(uint64_t epoch) {
  // (at most two rounds: an invalidation that was already in flight when the translation went away does not count,
  // the one after it does)
  for (int i = 0; i < 4 && !flushed_through(epoch); ++i) {
    std::lock_guard<std::mutex> g(flush_mu_);
    if (flushed_through(epoch)) break;
    do_shootdown();
  }
  drain_limbo();
}

void GpuContext::do_shootdown() {
  tlb_stale().store(false); // before the flush: an unmap racing with it stays owed
  const uint64_t epoch = flush_started_.fetch_add(1) + 1;
  struct Done {
    std::atomic<uint64_t> &d;
    uint64_t e;
    bool ok = false;
    ~Done() {
      if (ok)
        d.store(std::max(d.load(), e)); // (serialised by flush_mu_)
      else
        tlb_stale().store(true); // the flush threw: what it was to cover is still owed (flusher_loop and housekeeping only log)
    }
  } done{flush_done_, epoch};
  if (!options().tlb_shootdown.load()) {
    done.ok = true;
    return;
  }
  static const bool broken_for_test = KVC_TEST_HOOK("BREAK_TLB_FLUSH"); // hook: the init self test must notice
  const int64_t t0 = now_ns();
  if (broken_for_test) {
    // nothing: what a runtime that stopped flushing would look like
  } else if (kfd_flush_.ready()) {
    // The ioctl pair that ends in KFD's heavyweight flush, on our own buffer: nothing between us and the kernel can
    // answer it from a cache.
    int64_t map_ns = 0;
    if (!kfd_flush_.flush(&map_ns)) throw GpuError(std::string("KFD TLB flush failed: ") + strerror(errno));
    stats().seg[19] += map_ns; // (diagnostics, option 149: the re-MAP half of the pair; the UNMAP half is the flush itself)
  } else {
    // Fallback where /dev/kfd cannot be driven directly: an allocation that reaches KFD. 2 MiB is the smallest size
    // ROCr does not serve from its sub-allocator (measured: 4 KiB has no effect). What invalidates is the FREE (KFD's
    // unmap ends in the heavyweight flush); the allocation's map ends in a flush of its own when the address space has
    // changed since KFD last flushed - so, as with the ioctl pair (KfdTlbFlush::flush), a block is kept in hand: the
    // invalidation is its hipFree, and the hipMalloc of the next one right behind it finds nothing new to flush for.
    // A real trip to the kernel takes >150 us on MI355X; a hipFree that returns in <20 us was answered from a cache and
    // invalidated nothing, so a block too large for any cache is used instead. (Only this fallback watches the clock;
    // init's self test has checked that it invalidates at all.)
    if (!fallback_block_) HIP_CHECK(hipMalloc(&fallback_block_, 2u << 20));
    const int64_t t1 = now_ns();
    void *p = fallback_block_;
    fallback_block_ = nullptr;
    HIP_CHECK(hipFree(p));
    if (now_ns() - t1 < 20000) {
      static std::atomic<bool> warned{false};
      if (!warned.exchange(true)) KVC_LOG(LOG_WARNING, "TLB shootdown: freeing a 2 MiB allocation did not reach the driver; using 64 MiB blocks");
      HIP_CHECK(hipMalloc(&p, 64u << 20));
      HIP_CHECK(hipFree(p));
    }
    if (hipMalloc(&fallback_block_, 2u << 20) != hipSuccess) { // (for the next one; without it that one allocates first)
      (void)hipGetLastError();
      fallback_block_ = nullptr;
    }
  }
  done.ok = true;
  stats().tlb_shootdowns++;
  if (tl_background_thread) background_shootdowns()++;
  stats().shootdown_ns += now_ns() - t0;
}

void GpuContext::sync(hipStream_t s) {
  if (!s) s = stream_;
  HIP_CHECK(hipStreamSynchronize(s));
  harvest();
}

} // namespace kvc
