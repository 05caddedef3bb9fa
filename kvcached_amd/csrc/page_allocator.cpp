// page_allocator.cpp — see page_allocator.hpp. Reference behaviour is cited per function as
// csrc/page_allocator.cpp:<lines>; the golden traces in tests/golden pin it.
#include "page_allocator.hpp"

#include <algorithm>

#include "kv_allocator.hpp"

namespace kvc {

namespace {
// Environment defaults are read once per process like the reference's file-scope constants (:24-37).
int64_t env_min_reserved() {
  static const int64_t v = env_i64("KVCACHED_MIN_RESERVED_PAGES", 5);
  return v;
}
int64_t env_max_reserved() {
  static const int64_t v = env_i64("KVCACHED_MAX_RESERVED_PAGES", 10);
  return v;
}
double env_gpu_utilization() {
  static const double v = []() {
    const char *e = std::getenv("KVCACHED_GPU_UTILIZATION");
    return e ? std::atof(e) : 0.95;
  }();
  return v;
}
} // namespace

// ------------------------------------------------------------------ InternalPage
void InternalPage::init(int64_t block_mem_size) { // :44-53
  auto range = get_block_range(page_id, page_size, block_mem_size);
  num_kv_blocks_ = range.second - range.first;
  free_list_.clear();
  if (num_kv_blocks_ > 0) free_list_.reserve((size_t)num_kv_blocks_);
  for (int64_t b = range.first; b < range.second; ++b) free_list_.push_back(b);
}

std::vector<int64_t> InternalPage::alloc(int64_t num_blocks) { // :55-67 — the FIRST n free blocks
  if (num_blocks < 0 || free_list_.size() < static_cast<size_t>(num_blocks))
    throw std::runtime_error("Not enough free blocks in page");
  std::vector<int64_t> out(free_list_.begin(), free_list_.begin() + num_blocks);
  free_list_.erase(free_list_.begin(), free_list_.begin() + num_blocks);
  return out;
}

// ------------------------------------------------------------------ PageAllocator
PageAllocator::PageAllocator(int64_t num_layers, int64_t mem_size_per_layer, int64_t page_size, int64_t world_size,
                             int64_t pp_rank, bool async_sched, bool contiguous_layout, bool enable_page_prealloc,
                             int64_t num_kv_buffers, int64_t group_id, const std::string &ipc_name)
    : num_layers_(num_layers), mem_size_per_layer_(mem_size_per_layer), page_size_(page_size), world_size_(world_size),
      pp_rank_(pp_rank), num_kv_buffers_(num_kv_buffers), group_id_(group_id), async_sched_(async_sched),
      contiguous_layout_(contiguous_layout), enable_page_prealloc_(enable_page_prealloc),
      gpu_utilization_(env_gpu_utilization()), num_free_pages_(mem_size_per_layer / page_size),
      num_total_pages_(mem_size_per_layer / page_size) {
  if (num_layers <= 0 || page_size <= 0 || num_kv_buffers <= 0 || mem_size_per_layer < 0)
    throw InvalidError("PageAllocator: num_layers, page_size and num_kv_buffers must be positive");
  const int64_t n = num_free_pages_.load();
  min_reserved_ = std::min(n, env_min_reserved());
  max_reserved_ = std::min(n, env_max_reserved());
  for (int64_t i = 0; i < n; ++i) free_list_.push_back(i);
  tracker_ = std::make_unique<MemInfoTracker>(mem_size_per_layer * num_layers * num_kv_buffers, group_id, ipc_name);
  // the reference prints this line to stdout (:131-148); it is diagnostics, so it goes to the logger
  KVC_LOG(LOG_INFO,
          "Init PageAllocator: num_layers=%ld, mem_size_per_layer=%ldMB, total_mem_size=%ldMB, page_size=%ldMB, "
          "world_size=%ld, pp_rank=%ld, async_sched=%d, contiguous_layout=%d, enable_prealloc=%d, num_kv_buffers=%ld, "
          "group_id=%ld, min_reserved_pages=%ld, max_reserved_pages=%ld",
          (long)num_layers, (long)(mem_size_per_layer >> 20),
          (long)((num_kv_buffers * num_layers * mem_size_per_layer) >> 20), (long)(page_size >> 20), (long)world_size,
          (long)pp_rank, (int)async_sched, (int)contiguous_layout, (int)enable_page_prealloc, (long)num_kv_buffers,
          (long)group_id, (long)min_reserved_, (long)max_reserved_);
}

PageAllocator::~PageAllocator() {
  try {
    stop_threads();
  } catch (...) {
  }
}

// caller holds lock_ (:688-701)
void PageAllocator::publish_usage() {
  const int64_t unit = num_layers_ * page_size_ * num_kv_buffers_;
  tracker_->update_memory_usage(get_num_inuse_pages() * unit, static_cast<int64_t>(reserved_list_.size()) * unit);
}

page_id_t PageAllocator::alloc_page() { // :161-237
  std::unique_lock<std::mutex> lk(lock_);
  page_id_t pid = -1;
  while (pid == -1) {
    if (!reserved_list_.empty()) { // fast path: an already-mapped page
      pid = reserved_list_.front();
      reserved_list_.pop_front();
      num_free_pages_--;
      if (reserved_list_.size() < static_cast<size_t>(min_reserved_)) {
        prealloc_needed_ = true;
        cond_.notify_all();
      }
      publish_usage();
      return pid;
    }
    if (!free_list_.empty()) { // slow path: take a virtual page, back it below
      pid = free_list_.front();
      free_list_.pop_front();
      num_free_pages_--;
      break;
    }
    if (num_free_pages_.load() <= 0) throw NoPagesError("No free pages left");
    if (!enable_page_prealloc_ || !prealloc_running_)
      throw std::runtime_error("Inconsistent page allocator state: no free pages available");
    cond_.wait(lk); // the prealloc thread holds the remaining free pages; wait for it to publish them
  }
  lk.unlock();

  try {
    map_pages(&pid, 1);
  } catch (const std::exception &e) {
    std::lock_guard<std::mutex> g(lock_);
    free_list_.push_front(pid);
    num_free_pages_++;
    cond_.notify_all();
    throw std::runtime_error("Failed to map page " + std::to_string(pid) + ": " + e.what());
  }

  std::lock_guard<std::mutex> g(lock_);
  if (enable_page_prealloc_) { // trigger_preallocation, :226-228
    prealloc_needed_ = true;
    cond_.notify_all();
  }
  publish_usage();
  return pid;
}

// Addition (no reference counterpart): what n consecutive alloc_page() calls would hand out - reserved pages
// from the front first, then virtual pages from the front of the free list - but the pages that need backing go
// through ONE map call (one TLB invalidation, one broadcast to the TP workers) instead of n. All-or-nothing: a
// failure puts every id back where it was.
std::vector<page_id_t> PageAllocator::alloc_pages(int64_t n) {
  std::vector<page_id_t> out, to_map;
  if (n <= 0) return out;
  out.reserve(static_cast<size_t>(n));
  size_t from_reserved = 0;
  {
    std::unique_lock<std::mutex> lk(lock_);
    while (static_cast<int64_t>(out.size()) < n && !reserved_list_.empty()) {
      out.push_back(reserved_list_.front());
      reserved_list_.pop_front();
      num_free_pages_--;
    }
    from_reserved = out.size();
    while (static_cast<int64_t>(out.size() + to_map.size()) < n && !free_list_.empty()) {
      to_map.push_back(free_list_.front());
      free_list_.pop_front();
      num_free_pages_--;
    }
    if (from_reserved) {
      if (reserved_list_.size() < static_cast<size_t>(min_reserved_)) {
        prealloc_needed_ = true;
        cond_.notify_all();
      }
      publish_usage();
    }
  }
  if (!to_map.empty()) {
    try {
      map_pages(to_map.data(), to_map.size());
    } catch (const std::exception &e) {
      std::lock_guard<std::mutex> g(lock_);
      free_list_.insert(free_list_.begin(), to_map.begin(), to_map.end());
      reserved_list_.insert(reserved_list_.begin(), out.begin(), out.begin() + static_cast<std::ptrdiff_t>(from_reserved));
      num_free_pages_ += static_cast<int64_t>(to_map.size() + from_reserved);
      publish_usage();
      cond_.notify_all();
      throw std::runtime_error("Failed to map page " + std::to_string(to_map.front()) + ": " + e.what());
    }
    out.insert(out.end(), to_map.begin(), to_map.end());
    std::lock_guard<std::mutex> g(lock_);
    if (enable_page_prealloc_) {
      prealloc_needed_ = true;
      cond_.notify_all();
    }
    publish_usage();
  }
  // pages the prealloc thread is holding right now (or none left at all): the one-by-one path knows how to wait
  try {
    while (static_cast<int64_t>(out.size()) < n) out.push_back(alloc_page());
  } catch (...) {
    if (!out.empty()) free_pages(out.data(), out.size());
    throw;
  }
  return out;
}

void PageAllocator::free_page(page_id_t page_id) { // :239-262
  {
    std::lock_guard<std::mutex> g(lock_);
    num_free_pages_++;
    if (reserved_list_.size() < static_cast<size_t>(max_reserved_)) {
      reserved_list_.push_back(page_id);
      publish_usage();
      cond_.notify_all();
      return;
    }
  }
  unmap_pages(&page_id, 1);
  std::lock_guard<std::mutex> g(lock_);
  free_list_.push_back(page_id);
  publish_usage();
  cond_.notify_all();
}

void PageAllocator::free_pages(const page_id_t *page_ids, size_t n) { // :264-310
  std::vector<page_id_t> to_unmap;
  {
    std::lock_guard<std::mutex> g(lock_);
    num_free_pages_ += static_cast<int64_t>(n);
    const int64_t room = max_reserved_ - static_cast<int64_t>(reserved_list_.size());
    size_t keep = 0;
    if (room > 0) { // the first `room` ids stay mapped
      keep = std::min(static_cast<size_t>(room), n);
      reserved_list_.insert(reserved_list_.end(), page_ids, page_ids + keep);
    }
    to_unmap.assign(page_ids + keep, page_ids + n);
    if (room > 0 && to_unmap.empty()) {
      publish_usage();
      cond_.notify_all();
      return;
    }
  }
  unmap_pages(to_unmap.data(), to_unmap.size());
  std::lock_guard<std::mutex> g(lock_);
  free_list_.insert(free_list_.end(), to_unmap.begin(), to_unmap.end());
  publish_usage();
  cond_.notify_all();
}

bool PageAllocator::resize(int64_t new_mem_size) { // :312-401
  const int64_t new_pages = new_mem_size / page_size_;
  std::vector<page_id_t> to_unmap;
  // A smaller budget means "give memory back now" (kvctl limit): what the handle pool holds goes too, without
  // waiting for its idle decay. Done on scope exit, i.e. after lock_ is released.
  struct DrainOnShrink {
    bool armed = false;
    ~DrainOnShrink() {
      if (!armed) return;
      KvAllocator::flush_all_unmaps(); // queued (async) unmaps first: their handles are part of what goes back
      if (GpuContext *ctx = KvAllocator::gpu()) ctx->drain_pools();
    }
  } drain;
  {
    std::lock_guard<std::mutex> g(lock_);
    const int64_t total = num_total_pages_.load();
    if (new_pages < get_num_inuse_pages()) return false;
    if (new_pages == total) return true;
    if (new_pages > total) { // grow: reclaimed ids first (FIFO), then brand-new ids
      int64_t grow = new_pages - total;
      const int64_t reuse = std::min(static_cast<int64_t>(reclaimed_list_.size()), grow);
      for (int64_t i = 0; i < reuse; ++i) {
        free_list_.push_back(reclaimed_list_.front());
        reclaimed_list_.pop_front();
      }
      grow -= reuse;
      for (int64_t id = total; id < total + grow; ++id) free_list_.push_back(id);
      num_free_pages_ += reuse + grow;
      num_total_pages_ = new_pages;
      publish_usage();
      return true;
    }
    const int64_t shrink = total - new_pages;
    if (free_list_.size() >= static_cast<size_t>(shrink)) { // shrink: pop from the BACK of the free list
      for (int64_t i = 0; i < shrink; ++i) {
        reclaimed_list_.push_back(free_list_.back());
        free_list_.pop_back();
      }
      num_free_pages_ -= shrink;
      num_total_pages_ = new_pages;
      drain.armed = true;
      return true;
    }
    if (reserved_list_.empty()) return false;
    to_unmap.assign(reserved_list_.begin(), reserved_list_.end()); // need the reserved pages too
    reserved_list_.clear();
  }
  unmap_pages(to_unmap.data(), to_unmap.size());
  drain.armed = true;
  std::lock_guard<std::mutex> g(lock_);
  const int64_t shrink = num_total_pages_.load() - new_pages;
  free_list_.insert(free_list_.end(), to_unmap.begin(), to_unmap.end());
  publish_usage();
  if (free_list_.size() < static_cast<size_t>(shrink)) return false;
  for (int64_t i = 0; i < shrink; ++i) {
    reclaimed_list_.push_back(free_list_.back());
    free_list_.pop_back();
  }
  num_free_pages_ -= shrink;
  num_total_pages_ = new_pages;
  return true;
}

void PageAllocator::trim() { // :403-427
  std::vector<page_id_t> to_unmap;
  {
    std::lock_guard<std::mutex> g(lock_);
    to_unmap.assign(reserved_list_.begin(), reserved_list_.end());
    reserved_list_.clear();
    if (to_unmap.empty()) {
      publish_usage();
    }
  }
  if (!to_unmap.empty()) {
    unmap_pages(to_unmap.data(), to_unmap.size());
    std::lock_guard<std::mutex> g(lock_);
    free_list_.insert(free_list_.end(), to_unmap.begin(), to_unmap.end());
    publish_usage();
  }
  // trim() means "give physical memory back": queued (async) unmaps and idle pooled handles count as well
  KvAllocator::flush_all_unmaps();
  if (GpuContext *ctx = KvAllocator::gpu()) ctx->drain_pools();
}

void PageAllocator::reset_free_page_order() { // :703-709
  std::lock_guard<std::mutex> g(lock_);
  std::sort(free_list_.begin(), free_list_.end());
}

int64_t PageAllocator::get_num_reserved_pages() const {
  std::lock_guard<std::mutex> g(lock_);
  return static_cast<int64_t>(reserved_list_.size());
}

int64_t PageAllocator::get_avail_physical_pages() const { // :442-455
  size_t avail = 0, total = 0;
  mem_get_info(&avail, &total);
  const size_t headroom = total * (1.0 - gpu_utilization_);
  // the reference subtracts unsigned values (wraps to ~2^64 when free < headroom); clamp instead
  avail = avail > headroom ? avail - headroom : 0;
  const int64_t pages = static_cast<int64_t>(avail / static_cast<size_t>(page_size_));
  return pages / num_layers_ / num_kv_buffers_;
}

int64_t PageAllocator::check_and_get_resize_target(int64_t current_mem_size) const { // :462-469
  return tracker_->check_and_get_resize_target(current_mem_size, num_layers_, num_kv_buffers_);
}

std::unordered_map<page_id_t, std::vector<int64_t>>
PageAllocator::group_indices_by_page(const int64_t *indices, size_t n, int64_t block_mem_size) const { // :471-498
  // std::unordered_map + this exact reserve(): the iteration order becomes the order in which
  // KVCacheManager.free() visits pages, hence part of the bit-exact contract.
  std::unordered_map<page_id_t, std::vector<int64_t>> result;
  const int64_t blocks_per_page = page_size_ / block_mem_size;
  if (blocks_per_page <= 0) throw InvalidError("group_indices_by_page: block_mem_size larger than the page size");
  result.reserve(n / static_cast<size_t>(blocks_per_page) + 1);
  for (size_t i = 0; i < n; ++i) result[get_page_id(indices[i], block_mem_size)].push_back(indices[i]);
  return result;
}

std::vector<page_id_t> PageAllocator::page_list(int which) const {
  std::lock_guard<std::mutex> g(lock_);
  const auto &l = which == 0 ? free_list_ : which == 1 ? reserved_list_ : reclaimed_list_;
  return std::vector<page_id_t>(l.begin(), l.end());
}

void PageAllocator::set_broadcast_map_callback(BroadcastFn f) {
  std::lock_guard<std::mutex> g(lock_);
  map_cb_ = std::move(f);
}
void PageAllocator::set_broadcast_unmap_callback(BroadcastFn f) {
  std::lock_guard<std::mutex> g(lock_);
  unmap_cb_ = std::move(f);
}
void PageAllocator::set_should_use_worker_ipc_callback(BoolFn f) {
  std::lock_guard<std::mutex> g(lock_);
  worker_ipc_cb_ = std::move(f);
}

// page ids -> byte offsets of the K slot in one layer (:619-631)
std::vector<offset_t> PageAllocator::offsets_of(const page_id_t *ids, size_t n) const {
  std::vector<offset_t> off(n);
  const int64_t stride = contiguous_layout_ ? page_size_ * num_layers_ * num_kv_buffers_ : page_size_;
  for (size_t i = 0; i < n; ++i) off[i] = ids[i] * stride;
  return off;
}

bool PageAllocator::use_broadcast() const { // :633, :757-762
  if (world_size_ > 1) return true;
  BoolFn f;
  {
    std::lock_guard<std::mutex> g(lock_);
    f = worker_ipc_cb_;
  }
  return f ? f() : false;
}

void PageAllocator::map_pages(const page_id_t *ids, size_t n) { // :619-646
  auto offsets = offsets_of(ids, n);
  BroadcastFn cb;
  {
    std::lock_guard<std::mutex> g(lock_);
    cb = map_cb_;
  }
  if (cb && use_broadcast()) {
    if (cb(world_size_, offsets.data(), offsets.size()) != 0) throw CallbackError("broadcast map callback failed");
  } else {
    if (!KvAllocator::global(group_id_)->map_to_kv_tensors(offsets.data(), offsets.size()))
      throw std::runtime_error("Failed to map pages to KV tensors");
  }
}

void PageAllocator::unmap_pages(const page_id_t *ids, size_t n) { // :648-686
  auto offsets = offsets_of(ids, n);
  BroadcastFn cb;
  {
    std::lock_guard<std::mutex> g(lock_);
    cb = unmap_cb_;
  }
  if (cb && use_broadcast()) {
    if (cb(world_size_, offsets.data(), offsets.size()) != 0) throw CallbackError("broadcast unmap callback failed");
  } else {
    if (async_sched_) device_synchronize(); // in-flight kernels may still read the pages (:670-673)
    if (!KvAllocator::global(group_id_)->unmap_from_kv_tensors(offsets.data(), offsets.size()))
      throw std::runtime_error("Failed to unmap pages from KV tensors");
  }
}

// ------------------------------------------------------------------ background threads
void PageAllocator::start_prealloc_thread() { // :524-528, :717-733
  if (!enable_page_prealloc_) return;
  {
    std::lock_guard<std::mutex> g(lock_);
    if (!prealloc_thread_) {
      prealloc_running_ = true;
      prealloc_needed_ = true; // initial trigger
      prealloc_thread_ = std::make_unique<std::thread>(&PageAllocator::prealloc_worker, this);
    }
  }
  std::lock_guard<std::mutex> g(watcher_mu_);
  if (!watcher_thread_) {
    watcher_running_ = true;
    watcher_thread_ = std::make_unique<std::thread>(&PageAllocator::resize_watcher, this);
    if (GpuContext *ctx = KvAllocator::gpu()) ctx->add_housekeeper(+1); // it also runs the handle pools' housekeeping
  }
}

void PageAllocator::stop_prealloc_thread() { // :530-534
  if (enable_page_prealloc_) stop_threads();
}

void PageAllocator::stop_threads() { // :735-755
  std::unique_ptr<std::thread> t;
  {
    std::lock_guard<std::mutex> g(lock_);
    prealloc_running_ = false;
    cond_.notify_all();
    t = std::move(prealloc_thread_);
  }
  if (t && t->joinable()) t->join();
  std::unique_ptr<std::thread> w;
  {
    std::lock_guard<std::mutex> g(watcher_mu_);
    watcher_running_ = false;
    watcher_cv_.notify_all();
    w = std::move(watcher_thread_);
  }
  if (w && w->joinable()) {
    w->join();
    if (GpuContext *ctx = KvAllocator::gpu()) ctx->add_housekeeper(-1);
  }
}

void PageAllocator::prealloc_worker() { // :536-617
  std::unique_lock<std::mutex> lk(lock_);
  while (prealloc_running_) {
    cond_.wait(lk, [&] { return prealloc_needed_ || !prealloc_running_; });
    if (!prealloc_running_) break;
    prealloc_needed_ = false;

    int64_t want = std::max<int64_t>(0, min_reserved_ - static_cast<int64_t>(reserved_list_.size()));
    want = std::min(want, static_cast<int64_t>(free_list_.size()));
    if (want > 0) {
      int64_t phys = 0;
      try {
        phys = get_avail_physical_pages();
      } catch (const std::exception &e) {
        KVC_LOG(LOG_ERROR, "prealloc: cannot read free memory: %s", e.what());
      }
      want = std::min(want, phys);
    }
    if (want <= 0) continue;

    std::vector<page_id_t> batch(free_list_.begin(), free_list_.begin() + want); // front of the free list
    free_list_.erase(free_list_.begin(), free_list_.begin() + want);
    lk.unlock();
    bool ok = true;
    try {
      map_pages(batch.data(), batch.size()); // ONE batched map for the whole refill
    } catch (const std::exception &e) {
      ok = false;
      KVC_LOG(LOG_ERROR, "Failed to preallocate %zu pages: %s", batch.size(), e.what());
    }
    lk.lock();
    if (ok) {
      reserved_list_.insert(reserved_list_.end(), batch.begin(), batch.end());
      publish_usage();
    } else {
      free_list_.insert(free_list_.begin(), batch.begin(), batch.end());
    }
    cond_.notify_all();
  }
}

void PageAllocator::resize_watcher() { // :764-778 (100 ms poll; NB compares with the ctor-time size)
  std::unique_lock<std::mutex> lk(watcher_mu_);
  while (watcher_running_) {
    watcher_cv_.wait_for(lk, std::chrono::milliseconds(100), [&] { return !watcher_running_; });
    if (!watcher_running_) break;
    tracker_->revalidate();
    if (GpuContext *ctx = KvAllocator::gpu()) ctx->housekeeping(); // idle pooled handles must not starve a neighbour
    resize_target_.store(tracker_->check_and_get_resize_target(mem_size_per_layer_, num_layers_, num_kv_buffers_),
                         std::memory_order_relaxed);
  }
}

} // namespace kvc
