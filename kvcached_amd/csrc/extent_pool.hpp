// extent_pool.hpp — where physical pages come from and go back to: run-sized extents, page-sized pieces.
//
// No HIP or driver headers here: the pool is pure bookkeeping over an ExtentDriver (create / release callbacks), so its
// placement policy is exercised on the CPU (tests/native/extent_pool_check.cpp, benchmarks/sim_placement.py) and on the
// GPU by the same code.
//
// Why extents. A page-table update costs one DRM_AMDGPU_GEM_VA ioctl (2.2 us) per MAPPING, not per byte, and GEM_VA maps
// a range of one buffer object at an offset: n adjacent slots backed by n adjacent pages of one buffer are ONE ioctl
// (0.05-0.2 us per page, tools/drm_chunk_probe.cpp). Round 1 allocated fixed chunks of k pages and handed out pieces to
// whoever came: fast, but unrelated requests shared chunks and a chunk only goes back to the driver whole - 1.8x the
// mapped memory held at p90 on the soak, 4.7x at worst. Here an extent is born FROM A RUN: a map call that backs n
// adjacent unbacked slots creates one buffer of exactly n pages (n <= max_extent_pages, single slots get single pages).
// The pages of a run were asked for by one alloc() of one request and are, as a rule, given back together, so an
// extent empties as a whole; what does not (a straggler page pinning its extent) is measured - free pieces inside
// partly used extents are the pool's WASTE - and governed: pieces of partly used extents are always handed out before
// anything new is created, and while the waste exceeds `waste_frac` of the pages in use, new extents shrink (halving
// down to single pages, which cannot fragment); they grow back when the waste is gone.
// benchmarks/sim_placement.py replays the soak's real map/unmap sequence through this policy: held/mapped p50 1.00,
// p90 1.02-1.04, against 1.30 / 1.80 for fixed 16-page chunks (which the GPU soak had measured: 1.18 / 1.79).
//
// Whole idle extents are the recycling pool (bounded by a byte cap, decaying when idle, drained under memory pressure,
// evicted oldest-created first - ROCr's creation cost is O(live handles) with the oldest cheapest to release,
// DESIGN.md §4.5); with max_extent_pages = 1 (every backend but drm, exportable pools) the class is exactly that and
// nothing else.
#pragma once

#include <algorithm>
#include <atomic>
#include <cstdint>
#include <functional>
#include <mutex>
#include <set>
#include <stdexcept>
#include <unordered_map>
#include <utility>
#include <vector>

namespace kvc {

using phys_handle_t = uint64_t; // hipMemGenericAllocationHandle_t (a pointer), hsa_amd_vmem_alloc_handle_t::handle, or
                                // (drm backend, pages allocated straight from KFD) KFD's buffer handle

// A handle names ONE page-sized piece of an extent. For a one-page extent - every backend but drm, and single slots there -
// the piece id IS the buffer handle, untouched (ROCr's handles use bits above 48: nothing is ever masked out of them).
// A piece of a multi-page extent - those are KFD buffer handles, (gpu_id << 32 | idr) < 2^48 - is tagged with bit 63 and
// carries the extent's size - 1 in bits 48-55 and the piece index in bits 56-62.
constexpr int kPieceShift = 56, kPagesShift = 48;
constexpr uint64_t kPieceTag = 1ull << 63;
constexpr unsigned kMaxExtentPages = 64; // pieces of an extent are tracked in one 64-bit mask
inline bool is_piece(phys_handle_t h) { return (h & kPieceTag) != 0; }
inline phys_handle_t piece_id(phys_handle_t extent, unsigned piece, unsigned pages = 1) {
  if (pages <= 1) return extent;
  return extent | kPieceTag | (static_cast<uint64_t>(pages - 1) << kPagesShift) | (static_cast<uint64_t>(piece) << kPieceShift);
}
inline phys_handle_t chunk_of(phys_handle_t h) { return is_piece(h) ? (h & ((1ull << kPagesShift) - 1)) : h; }
inline unsigned piece_of(phys_handle_t h) { return is_piece(h) ? static_cast<unsigned>((h >> kPieceShift) & 0x7f) : 0; }
inline unsigned pages_of(phys_handle_t h) { return is_piece(h) ? static_cast<unsigned>((h >> kPagesShift) & 0xff) + 1 : 1; }

// Groups a batch of keys without sorting it (a batch of 1024 pages names a few dozen extents in any order): open addressing
// over a table of indices into the list of groups, which keeps the order of first appearance.
template <class V> class KeyGroups {
public:
  explicit KeyGroups(size_t n) {
    size_t cap = 16;
    while (cap < 2 * n) cap <<= 1;
    slot_.assign(cap, kNone);
    mask_ = cap - 1;
    items_.reserve(std::min<size_t>(n, 64));
  }
  V &at(uint64_t key) {
    if (last_ != kNone && items_[last_].first == key) return items_[last_].second;
    size_t i = (size_t)((key * 0x9E3779B97F4A7C15ull) >> 20) & mask_;
    for (; slot_[i] != kNone; i = (i + 1) & mask_)
      if (items_[slot_[i]].first == key) return items_[last_ = slot_[i]].second;
    slot_[i] = last_ = (uint32_t)items_.size();
    items_.emplace_back(key, V{});
    return items_.back().second;
  }
  std::vector<std::pair<uint64_t, V>> &items() { return items_; }

private:
  static constexpr uint32_t kNone = 0xffffffffu;
  std::vector<uint32_t> slot_;
  std::vector<std::pair<uint64_t, V>> items_;
  size_t mask_ = 0;
  uint32_t last_ = kNone;
};

struct VmmCounters { // in pages
  std::atomic<int64_t> created{0}, released{0}, reused{0};
};

// A physical page plus the creation order of its extent.
// scrub_ticket: 0 = the page may hold old data and must be zeroed by whoever maps it; otherwise it has been zeroed (or is
// being zeroed) by scrub number `scrub_ticket` on the owner's stream - wait for that one, write nothing.
// wait_ticket: the last scrub that was ever queued on the page's extent - whoever hands the page out waits for it in any
// case (a page that came back unscrubbed after a scrubbed life may still have that older fill in flight).
struct Phys {
  phys_handle_t h{};
  uint64_t seq = 0;
  uint64_t scrub_ticket = 0;
  uint64_t wait_ticket = 0;
};

struct ExtentDriver {
  // one buffer of `pages` pages; throws on failure. *tag (optional, 0 = none): an address at which the whole buffer stays
  // mapped for the pool's owner - the alias its pages are zeroed through after they come back (see scrub_addresses)
  std::function<phys_handle_t(size_t pages, uint64_t *tag)> create;
  std::function<void(phys_handle_t extent, uint64_t tag, size_t pages)> prepare_release; // per victim, before before_release (drop the alias)
  std::function<bool(phys_handle_t extent)> release; // back to the driver
  std::function<void()> before_release;              // once before a batch of releases (a TLB invalidation that was owed)
  std::function<bool()> under_pressure;              // the device is short of free memory: keep nothing idle
};

class ExtentPool {
public:
  // `ctr_scale`: what one unit of this pool counts for in the counters (a pool of lanes: the pages of a lane)
  ExtentPool(size_t page_bytes, unsigned max_extent_pages, ExtentDriver drv, VmmCounters *ctr, size_t ctr_scale = 1)
      : page_bytes_(page_bytes), kmax_cfg_(std::min(std::max(max_extent_pages, 1u), kMaxExtentPages)), kmax_cur_(kmax_cfg_),
        drv_(std::move(drv)), ctr_(ctr), scale_(ctr_scale ? ctr_scale : 1) {}
  ~ExtentPool() { drain(0); }

  size_t page_bytes() const { return page_bytes_; }
  size_t counter_scale() const { return scale_; }
  unsigned max_extent_pages() const { return kmax_cfg_; }
  bool multi_page() const { return kmax_cfg_ > 1; }
  unsigned current_extent_pages() {
    std::lock_guard<std::mutex> g(mu_);
    return kmax_cur_;
  }
  void set_cap_bytes(size_t b) { cap_pages_.store(b / page_bytes_); }
  void set_waste_frac(double f) { waste_frac_.store(f < 0 ? 0 : f); }
  void set_recover_pages(size_t n) {
    std::lock_guard<std::mutex> g(mu_);
    recover_pages_ = n;
  }
  // With a housekeeping thread around, what exceeds the cap is not released on the caller's free() path (the KFD free of
  // a used buffer costs 40-70 us: it is wiped) but by trim_to_cap() from that thread. Under memory pressure the release
  // is immediate either way.
  void set_defer_eviction(bool on) { defer_eviction_.store(on); }

  // Up to `want` pieces with consecutive indices in ONE extent - behind `want` adjacent slots they are one map ioctl.
  // Returns how many (>= 1; the caller comes back for the rest), or 0 when nothing is idle and !may_create.
  // *recycled: the memory may hold old data. Order of preference: an idle extent of exactly this size; a free run inside a
  // partly used extent; a larger idle extent; then - nothing has `want` neighbours - whatever free pieces partly used
  // extents still have (a shorter run: waste is used up before anything is created), smaller idle extents, and only then
  // a new extent of min(want, current extent size) pages.
  size_t acquire_run(size_t want, Phys *out, bool *recycled, bool may_create) {
    want = std::min<size_t>(std::max<size_t>(want, 1), kmax_cfg_);
    std::unique_lock<std::mutex> lk(mu_);
    if (take_idle_locked((unsigned)want, (unsigned)want, out, recycled)) return want;
    for (unsigned b = (unsigned)want; b <= kMaxExtentPages; ++b)
      if (!by_run_[b].empty()) return take_free_run_locked(*by_run_[b].begin(), (unsigned)want, out, recycled);
    for (unsigned n = (unsigned)want + 1; n <= kmax_cfg_; ++n)
      if (take_idle_locked(n, (unsigned)want, out, recycled)) return want;
    for (unsigned b = (unsigned)want - 1; b >= 1; --b)
      if (!by_run_[b].empty()) return take_free_run_locked(*by_run_[b].begin(), b, out, recycled);
    for (unsigned n = (unsigned)want - 1; n >= 1; --n)
      if (take_idle_locked(n, n, out, recycled)) return n;
    if (!may_create) return 0;
    const unsigned n = (unsigned)std::min<size_t>(want, kmax_cur_);
    lk.unlock();
    uint64_t tag = 0;
    const phys_handle_t h = drv_.create(n, &tag); // may throw: nothing of ours has changed yet
    lk.lock();
    if (kmax_cfg_ > 1 && (h >> kPagesShift) != 0) { // cannot be told apart from a piece id: never happens with KFD handles
      lk.unlock();
      if (drv_.prepare_release) drv_.prepare_release(h, tag, n);
      (void)drv_.release(h);
      throw std::runtime_error("extent pool: a multi-page buffer handle does not fit 48 bits");
    }
    Extent &e = tracked_[h];
    e.seq = ++next_seq_;
    e.n = (uint8_t)n;
    e.free_mask = full_mask(n);
    e.used_mask = 0;
    e.bucket = 0;
    e.tag = tag;
    e.clean_mask = 0; // what the driver hands out is zero today, but that is not a documented guarantee
    e.ticket = 0;
    free_pieces_ += n;
    held_pages_ += n;
    ctr_->created += n * scale_;
    ++creations_;
    demand_units_ += n;
    return take_pieces_locked(h, e, 0, n, out, recycled);
  }

  // Where the pages `ps` can be zeroed after their slots were unmapped: out[i] = the address of piece i inside its
  // extent's alias mapping, or 0 if the extent has none. The owner launches its fill on those addresses FIRST and then
  // gives the pages back with release_batch(ps, n, ticket) - a page must never be on offer before its scrub is queued.
  size_t scrub_addresses(const Phys *ps, size_t n, uint64_t *out) {
    std::lock_guard<std::mutex> g(mu_);
    size_t have = 0;
    for (size_t i = 0; i < n; ++i) {
      auto it = tracked_.find(key_of(ps[i].h));
      out[i] = (it != tracked_.end() && it->second.tag) ? it->second.tag + (uint64_t)idx_of(ps[i].h) * page_bytes_ : 0;
      have += out[i] != 0;
    }
    return have;
  }

  // the alias address of the buffer a handle is a piece of (0: none, or unknown)
  uint64_t tag_of(phys_handle_t h) {
    std::lock_guard<std::mutex> g(mu_);
    auto it = tracked_.find(key_of(h));
    return it == tracked_.end() ? 0 : it->second.tag;
  }

  void release(Phys p) { release_batch(&p, 1); }
  // `scrub_ticket` != 0: the pages that have an alias address were queued for zeroing as scrub number `scrub_ticket`
  // (scrub_addresses above); whoever gets them next waits for that scrub instead of zeroing them again.
  void release_batch(const Phys *ps, size_t n, uint64_t scrub_ticket = 0) {
    if (n == 0) return;
    std::vector<Victim> victims;
    const bool pressure = drv_.under_pressure && drv_.under_pressure();
    // pieces of one extent are handled together (one lookup, one re-bucketing): a batch lists a few dozen extents
    struct Group {
      uint64_t bits = 0;
      uint32_t cnt = 0;
      bool twice = false;
    };
    KeyGroups<Group> groups(n);
    for (size_t i = 0; i < n; ++i) {
      Group &g = groups.at(key_of(ps[i].h));
      const unsigned idx = idx_of(ps[i].h);
      const uint64_t bit = idx < 64 ? 1ull << idx : 0;
      g.twice = g.twice || bit == 0 || (g.bits & bit);
      g.bits |= bit;
      ++g.cnt;
    }
    {
      std::lock_guard<std::mutex> g(mu_);
      for (auto &kv : groups.items()) {
        const phys_handle_t h = kv.first;
        const uint64_t bits = kv.second.bits;
        const size_t cnt = kv.second.cnt;
        auto it = tracked_.find(h);
        if (it == tracked_.end() || kv.second.twice || (bits & it->second.free_mask) || (bits & ~full_mask(it->second.n))) {
          bad_releases_ += cnt; // pieces of an unknown extent, listed twice, or not out: never corrupt the masks
          continue;
        }
        Extent &e = it->second;
        e.free_mask |= bits;
        if (scrub_ticket && e.tag) {
          e.clean_mask |= bits;
          e.ticket = std::max(e.ticket, scrub_ticket);
        } else {
          e.clean_mask &= ~bits;
        }
        free_pieces_ += cnt;
        out_pieces_ -= cnt;
        if (e.free_mask == full_mask(e.n)) { // the extent is whole again: idle, ours to reuse or to give back
          unbucket_locked(h, e);
          free_pieces_ -= e.n;
          idle_insert_locked(h, IdleInfo{e.seq, e.n, true, e.tag, e.clean_mask, e.ticket});
          tracked_.erase(it);
        } else {
          rebucket_locked(h, e);
        }
      }
      const size_t keep = pressure ? 0 : (defer_eviction_.load() && cap_pages_.load() > 0 ? (size_t)-1 : cap_pages_.load());
      while (idle_pages_ > keep && pop_oldest_idle_locked(&victims)) {
      }
      low_water_ = std::min(low_water_, idle_pages_);
      govern_locked();
    }
    to_driver(victims);
  }

  // ---- housekeeping (the allocator's 10 Hz thread)
  // what release_batch left above the cap (deferred eviction), at most `max_pages` per call, oldest first
  size_t trim_to_cap(size_t max_pages) {
    std::vector<Victim> victims;
    size_t pages = 0;
    {
      std::lock_guard<std::mutex> g(mu_);
      while (idle_pages_ > cap_pages_.load() && pages < max_pages) {
        const size_t before = idle_pages_;
        if (!pop_oldest_idle_locked(&victims)) break;
        pages += before - idle_pages_;
      }
      low_water_ = std::min(low_water_, idle_pages_);
    }
    to_driver(victims);
    return pages;
  }
  // Give idle memory back to the driver, keeping at most `keep_pages` (the youngest extents).
  void drain(size_t keep_pages) {
    std::vector<Victim> victims;
    {
      std::lock_guard<std::mutex> g(mu_);
      while (idle_pages_ > keep_pages && pop_oldest_idle_locked(&victims)) {
      }
      low_water_ = std::min(low_water_, idle_pages_);
    }
    to_driver(victims);
  }
  // Idle-time decay: the pages nobody needed during a whole window of `idle_ns` - the low-water mark of the idle set
  // over that window - go back to the driver, at most `max_pages` per call, never below `floor_pages` (the reserve).
  // The pool is a recycling buffer for alloc/free churn, not a place to keep memory: a co-located engine computes what
  // it may use from hipMemGetInfo and never sees what is parked here (the reference releases on every unmap,
  // csrc/page.cpp:17).
  size_t decay(int64_t now_ns, int64_t idle_ns, size_t max_pages, size_t floor_pages = 0) {
    std::vector<Victim> victims;
    size_t pages = 0;
    {
      std::lock_guard<std::mutex> g(mu_);
      if (window_start_ns_ == 0 || idle_pages_ == 0) {
        window_start_ns_ = now_ns;
        low_water_ = idle_pages_;
        return 0;
      }
      if (now_ns - window_start_ns_ < idle_ns) return 0;
      size_t surplus = std::min(low_water_, idle_pages_);
      surplus = surplus > floor_pages ? surplus - floor_pages : 0;
      while (pages < surplus && pages < max_pages) {
        const size_t before = idle_pages_;
        if (!pop_oldest_idle_locked(&victims)) break;
        pages += before - idle_pages_;
      }
      if (pages < surplus) {
        low_water_ = floor_pages + surplus - pages; // keep going at the next tick
      } else {
        window_start_ns_ = now_ns;
        low_water_ = idle_pages_;
      }
    }
    to_driver(victims);
    return pages;
  }
  // Pre-create clean extents (never used: nothing to wipe, nothing to hide) until `target_pages` are idle, at most
  // `max_pages` per call: a growth burst then finds memory the kernel has already cleared. Allocating VRAM that has not
  // been handed out since boot costs ~80 us per 2 MiB (the kernel clears it on one SDMA ring at ~30 GB/s,
  // profiles/r02_create_cost.jsonl) against 1.8 us for memory that was wiped on release. The reserve is made of extents of
  // the LARGEST size in use right now (the governor's): a run of any length up to that is then served out of one of them with
  // one ioctl per run - and what it leaves over is handed out before anything else (acquire_run's second preference) -
  // whereas a reserve shaped like whatever happened to be created first (single pages, say) turns every later run into as
  // many ioctls as it has pages (profiles/r03_engine_geometry_before_lanes_compat.jsonl: 512 ioctls for 8 page ids).
  size_t refill_reserve(size_t target_pages, size_t max_pages) {
    size_t made = 0;
    while (made < max_pages) {
      unsigned n;
      {
        std::lock_guard<std::mutex> g(mu_);
        if (idle_pages_ + free_pieces_ >= target_pages) break;
        n = (unsigned)std::min<size_t>(kmax_cur_, target_pages - idle_pages_ - free_pieces_);
      }
      if (drv_.under_pressure && drv_.under_pressure()) break;
      phys_handle_t h;
      uint64_t tag = 0;
      try {
        h = drv_.create(n, &tag);
      } catch (...) {
        break; // no memory: the reserve is a convenience
      }
      std::lock_guard<std::mutex> g(mu_);
      held_pages_ += n;
      ctr_->created += n * scale_;
      idle_insert_locked(h, IdleInfo{++next_seq_, (uint8_t)n, false, tag, 0, 0});
      made += n;
    }
    return made;
  }

  // ---- accounting
  size_t creations() const { return creations_.load(); }
  size_t demand_units() const { return demand_units_.load(); } // units created ON DEMAND (inside somebody's map call), ever
  size_t idle_pages() {
    std::lock_guard<std::mutex> g(mu_);
    return idle_pages_;
  }
  // bytes that are ours to reuse and invisible to hipMemGetInfo: whole idle extents + free pieces of partly used ones
  size_t idle_bytes() {
    std::lock_guard<std::mutex> g(mu_);
    return (idle_pages_ + free_pieces_) * page_bytes_;
  }
  struct Footprint {
    size_t held_pages, out_pages, idle_pages, free_pieces, partial_extents, bad_releases;
    unsigned extent_pages_now;
  };
  Footprint footprint() {
    std::lock_guard<std::mutex> g(mu_);
    size_t partial = 0;
    for (unsigned b = 1; b <= kMaxExtentPages; ++b) partial += by_run_[b].size();
    return Footprint{held_pages_, out_pieces_, idle_pages_, free_pieces_, partial, bad_releases_, kmax_cur_};
  }
  // how many pieces of this extent are handed out (0: unknown or whole)
  unsigned pieces_out(phys_handle_t extent) {
    std::lock_guard<std::mutex> g(mu_);
    auto it = tracked_.find(key_of(extent));
    return it == tracked_.end() ? 0 : (unsigned)__builtin_popcountll(full_mask(it->second.n) & ~it->second.free_mask);
  }

private:
  struct Extent {
    uint64_t seq = 0;
    uint64_t free_mask = 0;
    uint64_t used_mask = 0;  // pieces that have been handed out before (their memory may hold old data)
    uint64_t clean_mask = 0; // free pieces that were queued for zeroing when they came back ...
    uint64_t ticket = 0;     // ... by scrubs up to this number
    uint64_t tag = 0;        // the driver's alias address of the buffer (0: none)
    uint8_t n = 1;
    uint8_t bucket = 0;      // longest free run (0: none free, not in by_run_)
  };
  struct IdleInfo {
    uint64_t seq;
    uint8_t n;
    bool used;
    uint64_t tag, clean_mask, ticket;
    uint64_t order = 0; // key in idle_n_
  };
  struct Victim {
    phys_handle_t h;
    uint64_t seq;
    unsigned n;
    uint64_t tag;
  };
  // Handles are only ever taken apart in a pool that makes multi-page extents (its buffers are KFD's, below 2^48);
  // a single-page pool passes whatever its driver returns through untouched (ROCr's handles use the high bits).
  phys_handle_t key_of(phys_handle_t h) const { return kmax_cfg_ > 1 ? chunk_of(h) : h; }
  unsigned idx_of(phys_handle_t h) const { return kmax_cfg_ > 1 ? piece_of(h) : 0; }
  static uint64_t full_mask(unsigned n) { return n >= 64 ? ~0ull : ((1ull << n) - 1); }
  // lowest start of `want` consecutive set bits, or -1
  static int find_run(uint64_t mask, unsigned want) {
    uint64_t m = mask;
    for (unsigned s = 1; s < want && m; ++s) m &= mask >> s;
    return m ? __builtin_ctzll(m) : -1;
  }
  static unsigned longest_run(uint64_t mask) {
    unsigned n = 0;
    while (mask) {
      mask &= mask << 1;
      ++n;
    }
    return n;
  }
  void unbucket_locked(phys_handle_t h, Extent &e) {
    if (e.bucket) by_run_[e.bucket].erase(h);
    e.bucket = 0;
  }
  void rebucket_locked(phys_handle_t h, Extent &e) {
    const unsigned b = longest_run(e.free_mask);
    if (b == e.bucket) return;
    unbucket_locked(h, e);
    e.bucket = (uint8_t)b;
    if (b) by_run_[b].insert(h);
  }
  // hands out pieces [first, first+n) of a tracked extent
  size_t take_pieces_locked(phys_handle_t h, Extent &e, unsigned first, unsigned n, Phys *out, bool *recycled) {
    unsigned old = 0;
    for (unsigned i = 0; i < n; ++i) {
      const uint64_t bit = 1ull << (first + i);
      e.free_mask &= ~bit;
      old += (e.used_mask & bit) != 0;
      e.used_mask |= bit;
      out[i] = Phys{piece_id(h, first + i, e.n), e.seq, (e.clean_mask & bit) ? std::max<uint64_t>(e.ticket, 1) : 0, e.ticket};
      e.clean_mask &= ~bit; // in somebody's hands now
    }
    free_pieces_ -= n;
    out_pieces_ += n;
    handed_out_since_clamp_ += n;
    ctr_->reused += old * scale_;
    *recycled = old > 0;
    rebucket_locked(h, e);
    govern_locked();
    return n;
  }
  size_t take_free_run_locked(phys_handle_t h, unsigned n, Phys *out, bool *recycled) {
    Extent &e = tracked_[h];
    const int first = find_run(e.free_mask, n);
    return take_pieces_locked(h, e, (unsigned)first, n, out, recycled);
  }
  // an idle extent of exactly `size` pages (the youngest), `take` pieces of it handed out
  bool take_idle_locked(unsigned size, unsigned take, Phys *out, bool *recycled) {
    auto &s = idle_n_[size];
    if (s.empty()) return false;
    // Which one? A pool of single pages hands out the youngest buffer (ROCr's creation cost is O(live handles) with the
    // oldest cheapest to release: they stay at the eviction end). A pool of extents hands out the one that has been idle
    // longest: its pages were zeroed on their way back, and the longer ago that was queued the surer it has finished.
    const auto pos = kmax_cfg_ > 1 ? s.begin() : std::prev(s.end());
    const auto key = *pos;
    s.erase(pos);
    const phys_handle_t h = key.second;
    const IdleInfo info = idle_info_[h];
    idle_all_.erase({info.seq, h});
    idle_info_.erase(h);
    idle_pages_ -= size;
    low_water_ = std::min(low_water_, idle_pages_);
    Extent &e = tracked_[h];
    e.seq = info.seq;
    e.n = (uint8_t)size;
    e.free_mask = full_mask(size);
    e.used_mask = info.used ? full_mask(size) : 0;
    e.bucket = 0;
    e.tag = info.tag;
    e.clean_mask = info.clean_mask;
    e.ticket = info.ticket;
    free_pieces_ += size;
    take_pieces_locked(h, e, 0, take, out, recycled);
    return true;
  }
  void idle_insert_locked(phys_handle_t h, IdleInfo info) {
    info.order = kmax_cfg_ > 1 ? ++idle_stamp_ : info.seq; // extents: by the time they became idle; single pages: by creation
    idle_n_[info.n].insert({info.order, h});
    idle_all_.insert({info.seq, h});
    idle_info_[h] = info;
    idle_pages_ += info.n;
  }
  bool pop_oldest_idle_locked(std::vector<Victim> *victims) {
    if (idle_all_.empty()) return false;
    const auto key = *idle_all_.begin();
    idle_all_.erase(idle_all_.begin());
    const IdleInfo info = idle_info_[key.second];
    idle_info_.erase(key.second);
    idle_n_[info.n].erase({info.order, key.second});
    idle_pages_ -= info.n;
    victims->push_back(Victim{key.second, key.first, info.n, info.tag});
    return true;
  }
  // Waste = free pieces inside partly used extents. Above waste_frac of the pages in use, new extents halve at once
  // (single pages cannot fragment). Trust comes back slowly: one doubling per `recover_pages` pages handed out with the
  // waste well inside its budget - a workload that keeps breaking its runs apart (memory that grows in runs and ebbs
  // page by page) is served with single pages after its first tide, while one clamp costs a steady workload little.
  void govern_locked() {
    if (kmax_cfg_ == 1) return;
    const double budget = waste_frac_.load() * (double)std::max<size_t>(out_pieces_, 64);
    if ((double)free_pieces_ > budget) {
      if (kmax_cur_ > 1) kmax_cur_ /= 2;
      handed_out_since_clamp_ = 0;
    } else if (kmax_cur_ < kmax_cfg_ && (double)free_pieces_ <= 0.25 * budget && handed_out_since_clamp_ >= recover_pages_) {
      kmax_cur_ = std::min(kmax_cfg_, kmax_cur_ * 2);
      handed_out_since_clamp_ = 0;
    }
  }
  void to_driver(std::vector<Victim> &v) {
    if (v.empty()) return;
    if (drv_.prepare_release)
      for (auto &p : v) drv_.prepare_release(p.h, p.tag, p.n); // aliases go first: their translations are covered by ...
    if (drv_.before_release) drv_.before_release();            // ... the invalidation owed before memory leaves
    std::sort(v.begin(), v.end(), [](const Victim &a, const Victim &b) { return a.seq < b.seq; }); // oldest first
    size_t pages = 0;
    for (auto &p : v) {
      if (!drv_.release(p.h)) ++failed_releases_;
      ctr_->released += p.n * scale_;
      pages += p.n;
    }
    std::lock_guard<std::mutex> g(mu_);
    held_pages_ -= std::min(held_pages_, pages);
  }

  const size_t page_bytes_;
  const unsigned kmax_cfg_;
  unsigned kmax_cur_;
  ExtentDriver drv_;
  VmmCounters *ctr_;
  const size_t scale_;
  std::atomic<size_t> cap_pages_{0};
  std::atomic<double> waste_frac_{0.05};
  std::atomic<bool> defer_eviction_{false};
  std::mutex mu_;
  uint64_t next_seq_ = 0, idle_stamp_ = 0;
  std::atomic<size_t> creations_{0}; // extents created on demand (not for the reserve)
  std::atomic<size_t> demand_units_{0};
  size_t handed_out_since_clamp_ = 0;
  size_t recover_pages_ = 4096; // 8 GiB of 2 MiB pages handed out between two steps back up
  size_t held_pages_ = 0;   // everything obtained from the driver and not given back
  size_t out_pieces_ = 0;   // pages handed out (mapped by the caller)
  size_t free_pieces_ = 0;  // free pieces inside tracked (partly used) extents: the waste
  size_t idle_pages_ = 0;   // pages of whole idle extents
  size_t bad_releases_ = 0, failed_releases_ = 0;
  std::unordered_map<phys_handle_t, Extent> tracked_;          // extents with at least one piece handed out
  std::set<phys_handle_t> by_run_[kMaxExtentPages + 1];         // ... bucketed by their longest free run
  std::set<std::pair<uint64_t, phys_handle_t>> idle_n_[kMaxExtentPages + 1]; // whole idle extents by size, in hand-out order (IdleInfo::order)
  std::set<std::pair<uint64_t, phys_handle_t>> idle_all_;       // ... all of them, oldest first (eviction order)
  std::unordered_map<phys_handle_t, IdleInfo> idle_info_;
  size_t low_water_ = 0;        // smallest idle_pages_ since window_start_ns_
  int64_t window_start_ns_ = 0;
};

} // namespace kvc
