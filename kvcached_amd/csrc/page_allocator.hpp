// page_allocator.hpp — block/page bookkeeping of the elastic KV cache.
//
// Same observable behaviour as the reference's InternalPage / PageAllocator
// (csrc/page_allocator.cpp, csrc/inc/page_allocator.hpp): identical page-id and block-id
// sequences on identical call traces (tests/golden). Differences are in how it runs:
// no Python GIL anywhere (callbacks are plain C function pointers, so the prealloc thread can
// never deadlock against a caller that holds the GIL — SURVEY §5), the shm segment is a
// persistent mapping (mem_info.hpp), counters read without the lock are atomics, GPU errors
// surface as exceptions that the rollback paths handle, and the free-memory underflow of
// get_avail_physical_pages is clamped.
#pragma once

#include <atomic>
#include <condition_variable>
#include <cstdint>
#include <deque>
#include <functional>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <unordered_map>
#include <vector>

#include "common.hpp"
#include "mem_info.hpp"

namespace kvc {

class InternalPage {
public:
  InternalPage(page_id_t id, int64_t size) : page_id(id), page_size(size) {}

  const page_id_t page_id;
  const int64_t page_size;

  void init(int64_t block_mem_size);
  std::vector<int64_t> alloc(int64_t num_blocks);
  void free(int64_t block_id) { free_list_.push_back(block_id); }
  void free_batch(const int64_t *ids, size_t n) { free_list_.insert(free_list_.end(), ids, ids + n); }
  bool empty() const { return free_list_.size() == static_cast<size_t>(num_kv_blocks_); }
  bool full() const { return free_list_.empty(); }
  int64_t num_free_blocks() const { return static_cast<int64_t>(free_list_.size()); }
  const std::vector<int64_t> &free_blocks() const { return free_list_; }

  // first whole block at/after the page start, first block that no longer fits (page_allocator.cpp:87-95)
  static std::pair<int64_t, int64_t> get_block_range(page_id_t page_id, int64_t page_size, int64_t block_mem_size) {
    return {(page_id * page_size + block_mem_size - 1) / block_mem_size, ((page_id + 1) * page_size) / block_mem_size};
  }
  static int64_t get_num_blocks(int64_t page_size, int64_t block_mem_size) { return page_size / block_mem_size; }

private:
  int64_t num_kv_blocks_ = 0;
  std::vector<int64_t> free_list_;
};

// returns nonzero on failure
using BroadcastFn = std::function<int(int64_t world_size, const offset_t *offsets, size_t n)>;
using BoolFn = std::function<bool()>;

class PageAllocator {
public:
  PageAllocator(int64_t num_layers, int64_t mem_size_per_layer, int64_t page_size, int64_t world_size, int64_t pp_rank,
                bool async_sched, bool contiguous_layout, bool enable_page_prealloc, int64_t num_kv_buffers,
                int64_t group_id, const std::string &ipc_name);
  ~PageAllocator();

  page_id_t alloc_page();
  // n pages with ONE map call for those that need backing (addition; same ids, same offsets in the same order
  // as n alloc_page() calls, all-or-nothing)
  std::vector<page_id_t> alloc_pages(int64_t n);
  void free_page(page_id_t page_id);
  void free_pages(const page_id_t *page_ids, size_t n);
  bool resize(int64_t new_mem_size);
  void trim();
  void reset_free_page_order();

  int64_t get_num_free_pages() const { return num_free_pages_.load(std::memory_order_relaxed); }
  int64_t get_num_inuse_pages() const {
    return num_total_pages_.load(std::memory_order_relaxed) - num_free_pages_.load(std::memory_order_relaxed);
  }
  int64_t get_num_total_pages() const { return num_total_pages_.load(std::memory_order_relaxed); }
  int64_t get_num_reserved_pages() const;
  int64_t get_avail_physical_pages() const;
  int64_t check_and_get_resize_target(int64_t current_mem_size) const;
  int64_t get_resize_target() const { return resize_target_.load(std::memory_order_relaxed); }
  page_id_t get_page_id(int64_t block_id, int64_t block_mem_size) const { return block_id * block_mem_size / page_size_; }
  std::unordered_map<page_id_t, std::vector<int64_t>> group_indices_by_page(const int64_t *indices, size_t n,
                                                                            int64_t block_mem_size) const;
  void start_prealloc_thread();
  void stop_prealloc_thread();

  void set_broadcast_map_callback(BroadcastFn f);
  void set_broadcast_unmap_callback(BroadcastFn f);
  void set_should_use_worker_ipc_callback(BoolFn f);

  std::vector<page_id_t> page_list(int which) const; // 0 free, 1 reserved, 2 reclaimed
  const std::string &ipc_name() const { return tracker_->ipc_name(); }
  int64_t page_size() const { return page_size_; }

private:
  void prealloc_worker();
  void resize_watcher();
  std::vector<offset_t> offsets_of(const page_id_t *ids, size_t n) const;
  void map_pages(const page_id_t *ids, size_t n);
  void unmap_pages(const page_id_t *ids, size_t n);
  void publish_usage(); // caller holds lock_
  bool use_broadcast() const;
  void stop_threads();

  const int64_t num_layers_, mem_size_per_layer_, page_size_, world_size_;
  [[maybe_unused]] const int64_t pp_rank_; // carried for the broadcast callbacks' owner (tp_ipc_util picks the sockets by it)
  const int64_t num_kv_buffers_, group_id_;
  const bool async_sched_, contiguous_layout_, enable_page_prealloc_;
  const double gpu_utilization_;

  std::atomic<int64_t> num_free_pages_, num_total_pages_;
  std::deque<page_id_t> free_list_, reserved_list_, reclaimed_list_;
  int64_t min_reserved_, max_reserved_;

  mutable std::mutex lock_;
  std::condition_variable cond_;
  bool prealloc_running_ = false, prealloc_needed_ = false;
  std::unique_ptr<std::thread> prealloc_thread_;

  std::atomic<int64_t> resize_target_{-1};
  std::mutex watcher_mu_;
  std::condition_variable watcher_cv_;
  bool watcher_running_ = false;
  std::unique_ptr<std::thread> watcher_thread_;

  std::unique_ptr<MemInfoTracker> tracker_;
  BroadcastFn map_cb_, unmap_cb_;
  BoolFn worker_ipc_cb_;
};

} // namespace kvc
