// oracle/kvc_oracle.cpp — TEST INFRASTRUCTURE, NOT PRODUCT CODE.
//
// A plain, single-threaded CPU restatement of the reference algorithm for the
// hot path, written to be read side by side with the reference.
// Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load
// this library; kvcached_amd/ never does.
//
// Parity status: PINNED. tests/golden/*.json were produced by running the real
// reference (csrc compiled from /root/reference by oracle/Makefile into
// oracle/_ref/vmm_ops.so, driven by the reference's own Python
// kvcached/kv_cache_manager.py) through oracle/gen_golden.py;
// tests/test_oracle_golden.py replays every fixture through this file.
//
// What is restated (reference file:line):
//   OInternalPage      csrc/page_allocator.cpp:40-100
//   OPageAllocator     csrc/page_allocator.cpp:103-782   (threads replaced by
//                      explicit, deterministic prealloc_step()/watcher_tick())
//   shm triple         csrc/inc/mem_info_tracker.hpp:25-36,176-204
//   OKVCacheManager    kvcached/kv_cache_manager.py:58-506
//   page-offset math   csrc/page_allocator.cpp:619-631, csrc/allocator.cpp:161-257
//   zero_fill / compact_blocks: north-star additions with no reference symbol;
//                      restated as memset / memcpy loops.
//
// std::unordered_map is used ON PURPOSE in group_indices_by_page: the reference
// leaks libstdc++'s iteration order into a Python dict
// (csrc/page_allocator.cpp:471-498), so the order is part of the contract.

#include <algorithm>
#include <cstdint>
#include <cstring>
#include <deque>
#include <map>
#include <memory>
#include <stdexcept>
#include <string>
#include <unordered_map>
#include <utility>
#include <vector>

namespace okvc {

// ---------------------------------------------------------------- InternalPage
struct OInternalPage {
  int64_t page_id, page_size;
  int64_t start_block = 0, end_block = 0, num_kv_blocks = 0;
  std::vector<int64_t> free_list;

  OInternalPage(int64_t id, int64_t size) : page_id(id), page_size(size) {}

  // csrc/page_allocator.cpp:87-95
  static std::pair<int64_t, int64_t> get_block_range(int64_t page_id, int64_t page_size,
                                                     int64_t block_mem_size) {
    int64_t start = (page_id * page_size + block_mem_size - 1) / block_mem_size;
    int64_t end = ((page_id + 1) * page_size) / block_mem_size;
    return {start, end};
  }
  // :97-100
  static int64_t get_num_blocks(int64_t page_size, int64_t block_mem_size) {
    return page_size / block_mem_size;
  }
  // :44-53
  void init(int64_t block_mem_size) {
    auto r = get_block_range(page_id, page_size, block_mem_size);
    start_block = r.first;
    end_block = r.second;
    num_kv_blocks = end_block - start_block;
    free_list.clear();
    for (int64_t i = start_block; i < end_block; ++i) free_list.push_back(i);
  }
  // :55-67  (first n, erased from the front)
  std::vector<int64_t> alloc(int64_t n) {
    if (free_list.size() < static_cast<size_t>(n))
      throw std::runtime_error("Not enough free blocks in page");
    std::vector<int64_t> out(free_list.begin(), free_list.begin() + n);
    free_list.erase(free_list.begin(), free_list.begin() + n);
    return out;
  }
  void free_one(int64_t b) { free_list.push_back(b); }                   // :69
  void free_batch(const int64_t *ids, size_t n) { free_list.insert(free_list.end(), ids, ids + n); } // :71-73
  bool empty() const { return free_list.size() == static_cast<size_t>(num_kv_blocks); } // :75-77
  bool full() const { return free_list.empty(); }                        // :79
  int64_t num_free_blocks() const { return (int64_t)free_list.size(); }  // :81-83
};

// ---------------------------------------------------------------- PageAllocator
struct MapEvent {
  int kind; // 0 = map, 1 = unmap
  std::vector<int64_t> offsets;
};

struct OPageAllocator {
  int64_t num_layers, mem_size_per_layer, page_size, world_size, num_kv_buffers;
  bool contiguous_layout, enable_prealloc;
  int64_t num_free_pages, num_total_pages;
  std::deque<int64_t> free_list, reserved_list, reclaimed_list;
  int64_t min_reserved, max_reserved;
  bool prealloc_needed = false;
  // shm triple, csrc/inc/mem_info_tracker.hpp:25-36
  int64_t shm_total, shm_used = 0, shm_prealloc = 0;
  int64_t resize_target = -1;
  // injected in place of hipMemGetInfo (csrc/page_allocator.cpp:442-455)
  int64_t avail_phys_pages = INT64_MAX / 4;
  std::vector<MapEvent> log;

  // csrc/page_allocator.cpp:103-149 (env defaults 5/10 at :24-32)
  OPageAllocator(int64_t L, int64_t mem, int64_t P, int64_t ws, bool contig, bool prealloc,
                 int64_t kv, int64_t min_res_env, int64_t max_res_env)
      : num_layers(L), mem_size_per_layer(mem), page_size(P), world_size(ws), num_kv_buffers(kv),
        contiguous_layout(contig), enable_prealloc(prealloc), num_free_pages(mem / P),
        num_total_pages(mem / P) {
    min_reserved = std::min(num_free_pages, min_res_env);
    max_reserved = std::min(num_free_pages, max_res_env);
    for (int64_t i = 0; i < num_free_pages; ++i) free_list.push_back(i);
    shm_total = mem * L * kv; // init_kv_cache_limit, mem_info_tracker.hpp:210-219
  }

  int64_t get_num_inuse_pages() const { return num_total_pages - num_free_pages; } // :431-433

  // :688-701
  void update_memory_usage() {
    shm_used = get_num_inuse_pages() * num_layers * page_size * num_kv_buffers;
    shm_prealloc = (int64_t)reserved_list.size() * num_layers * page_size * num_kv_buffers;
  }

  // :619-631 / :648-662 — page ids -> byte offsets handed to map/unmap_to_kv_tensors
  std::vector<int64_t> to_offsets(const std::vector<int64_t> &pids) const {
    std::vector<int64_t> off;
    off.reserve(pids.size());
    for (int64_t pid : pids)
      off.push_back(contiguous_layout ? pid * page_size * num_layers * num_kv_buffers
                                      : pid * page_size);
    return off;
  }
  void map_pages(const std::vector<int64_t> &pids) { log.push_back({0, to_offsets(pids)}); }
  void unmap_pages(const std::vector<int64_t> &pids) { log.push_back({1, to_offsets(pids)}); }

  // :161-237. Returns page id; throws like the reference.
  int64_t alloc_page() {
    int64_t page_id = -1;
    while (page_id == -1) {
      if (!reserved_list.empty()) { // fast path :169-190
        page_id = reserved_list.front();
        reserved_list.pop_front();
        num_free_pages--;
        if (reserved_list.size() < (size_t)min_reserved) prealloc_needed = true;
        update_memory_usage();
        return page_id;
      }
      if (!free_list.empty()) { // slow path :193-198
        page_id = free_list.front();
        free_list.pop_front();
        num_free_pages--;
        break;
      }
      if (num_free_pages <= 0) throw std::runtime_error("No free pages left"); // :200-202
      // :204-207 (without a prealloc thread there is nobody to wait for)
      throw std::runtime_error("Inconsistent page allocator state: no free pages available");
    }
    map_pages({page_id}); // :216
    if (enable_prealloc) prealloc_needed = true; // :226-228
    update_memory_usage();
    return page_id;
  }

  // :239-262
  void free_page(int64_t pid) {
    num_free_pages++;
    if (reserved_list.size() < (size_t)max_reserved) {
      reserved_list.push_back(pid);
      update_memory_usage();
      return;
    }
    unmap_pages({pid});
    free_list.push_back(pid);
    update_memory_usage();
  }

  // :264-310
  void free_pages(const std::vector<int64_t> &pids) {
    std::vector<int64_t> to_unmap;
    num_free_pages += (int64_t)pids.size();
    int64_t num_to_reserve = max_reserved - (int64_t)reserved_list.size();
    if (num_to_reserve > 0) {
      size_t k = std::min((size_t)num_to_reserve, pids.size());
      reserved_list.insert(reserved_list.end(), pids.begin(), pids.begin() + k);
      to_unmap.assign(pids.begin() + k, pids.end());
      if (to_unmap.empty()) {
        update_memory_usage();
        return;
      }
    } else {
      to_unmap = pids;
    }
    unmap_pages(to_unmap);
    free_list.insert(free_list.end(), to_unmap.begin(), to_unmap.end());
    update_memory_usage();
  }

  // :312-401
  bool resize(int64_t new_mem_size) {
    int64_t new_num_pages = new_mem_size / page_size;
    std::vector<int64_t> to_unmap;
    if (new_num_pages < get_num_inuse_pages()) return false;
    if (new_num_pages == num_total_pages) return true;
    if (new_num_pages > num_total_pages) {
      int64_t num_to_expand = new_num_pages - num_total_pages;
      int64_t num_to_reuse = std::min((int64_t)reclaimed_list.size(), num_to_expand);
      if (num_to_reuse > 0) {
        for (int64_t i = 0; i < num_to_reuse; ++i) {
          free_list.push_back(reclaimed_list.front());
          reclaimed_list.pop_front();
        }
        num_to_expand -= num_to_reuse;
        num_free_pages += num_to_reuse;
      }
      if (num_to_expand > 0) {
        for (int64_t i = num_total_pages; i < num_total_pages + num_to_expand; ++i)
          free_list.push_back(i);
        num_free_pages += num_to_expand;
      }
      num_total_pages = new_num_pages;
      update_memory_usage();
      return true;
    }
    int64_t num_to_reclaim = num_total_pages - new_num_pages;
    if (free_list.size() < (size_t)num_to_reclaim) {
      if (!reserved_list.empty()) {
        to_unmap.assign(reserved_list.begin(), reserved_list.end());
        reserved_list.clear();
      } else {
        return false;
      }
    } else {
      for (int64_t i = 0; i < num_to_reclaim; ++i) {
        reclaimed_list.push_back(free_list.back());
        free_list.pop_back();
      }
      num_free_pages -= num_to_reclaim;
      num_total_pages = new_num_pages;
      return true;
    }
    unmap_pages(to_unmap); // :379
    free_list.insert(free_list.end(), to_unmap.begin(), to_unmap.end());
    update_memory_usage();
    if (free_list.size() < (size_t)num_to_reclaim) return false;
    for (int64_t i = 0; i < num_to_reclaim; ++i) {
      reclaimed_list.push_back(free_list.back());
      free_list.pop_back();
    }
    num_free_pages -= num_to_reclaim;
    num_total_pages = new_num_pages;
    return true;
  }

  // :403-427
  void trim() {
    std::vector<int64_t> to_unmap(reserved_list.begin(), reserved_list.end());
    reserved_list.clear();
    if (to_unmap.empty()) {
      update_memory_usage();
      return;
    }
    unmap_pages(to_unmap);
    free_list.insert(free_list.end(), to_unmap.begin(), to_unmap.end());
    update_memory_usage();
  }

  // :703-709
  void reset_free_page_order() { std::sort(free_list.begin(), free_list.end()); }

  // :457-460
  int64_t get_page_id(int64_t block_id, int64_t block_mem_size) const {
    return block_id * block_mem_size / page_size;
  }

  // :471-498
  std::unordered_map<int64_t, std::vector<int64_t>>
  group_indices_by_page(const int64_t *idx, size_t n, int64_t block_mem_size) const {
    std::unordered_map<int64_t, std::vector<int64_t>> result;
    int64_t blocks_per_page = page_size / block_mem_size;
    result.reserve(n / blocks_per_page + 1);
    for (size_t i = 0; i < n; ++i) result[get_page_id(idx[i], block_mem_size)].push_back(idx[i]);
    return result;
  }

  // One complete pass of prealloc_worker's loop body (:552-608), run synchronously.
  // Returns the number of pages moved free -> reserved.
  int64_t prealloc_step() {
    if (!enable_prealloc || !prealloc_needed) return 0;
    prealloc_needed = false;
    int64_t current = (int64_t)reserved_list.size();
    int64_t to_reserve = std::max<int64_t>(0, min_reserved - current);
    to_reserve = std::min({to_reserve, (int64_t)free_list.size(), avail_phys_pages});
    if (to_reserve <= 0) return 0;
    std::vector<int64_t> pages;
    for (int64_t i = 0; i < to_reserve && !free_list.empty(); ++i) {
      pages.push_back(free_list.front());
      free_list.pop_front();
    }
    if (pages.empty()) return 0;
    map_pages(pages);
    reserved_list.insert(reserved_list.end(), pages.begin(), pages.end());
    update_memory_usage();
    return (int64_t)pages.size();
  }

  // resize_watcher body (:771-775) + MemInfoTracker::check_and_get_resize_target
  // (mem_info_tracker.hpp:191-204). NB compares with the ctor-time size (never updated).
  void watcher_tick() {
    int64_t new_mem = shm_total / num_layers / num_kv_buffers;
    resize_target = (new_mem != mem_size_per_layer) ? new_mem : -1;
  }
};

// ---------------------------------------------------------------- KVCacheManager
// kvcached/kv_cache_manager.py:58-506. Python containers restated:
//   avail_pages (dict, popitem() = LIFO)  -> vector of (page_id, page) in insertion order
//   full_pages  (dict)                    -> std::map (only keyed lookups + clear())
struct OKVCacheManager {
  int64_t num_blocks, block_mem_size, num_layers, num_kv_buffers, page_size, mem_size;
  bool reserve_null_block;
  OPageAllocator pa;
  int64_t num_avail_blocks = 0;
  std::vector<std::pair<int64_t, std::shared_ptr<OInternalPage>>> avail_pages;
  std::vector<std::pair<int64_t, std::shared_ptr<OInternalPage>>> full_pages; // insertion order kept for clear()
  std::vector<int64_t> reserved_blocks;
  bool in_shrink = false;
  int64_t target_num_blocks = -1;
  bool null_block_ok = true;

  // :60-188 (block_mem_size > page_size is rejected by the caller, :104-116)
  OKVCacheManager(int64_t nb, int64_t block_size, int64_t cell_size, int64_t L, int64_t ws,
                  bool null_blk, int64_t kv, int64_t P, bool contig, bool prealloc,
                  int64_t min_res, int64_t max_res)
      : num_blocks(nb), block_mem_size(block_size * cell_size), num_layers(L), num_kv_buffers(kv),
        page_size(P), mem_size(nb * block_size * cell_size), reserve_null_block(null_blk),
        pa(L, nb * block_size * cell_size, P, ws, contig, prealloc, kv, min_res, max_res) {}

  // _post_init (:190-227): reserve the null block, then the prealloc thread's first trigger.
  void post_init() {
    if (reserve_null_block) {
      std::vector<int64_t> out;
      bool ok = alloc(1, out);
      null_block_ok = ok && out.size() == 1 && out[0] == 0; // :233-243
    }
    if (pa.enable_prealloc) pa.prealloc_needed = true; // start_prealloc_thread -> trigger (:717-725)
  }

  template <class V> static typename V::iterator find_page(V &v, int64_t pid) {
    return std::find_if(v.begin(), v.end(), [&](auto &kv) { return kv.first == pid; });
  }

  // :411-423
  int64_t available_size() {
    int64_t avail_blocks = num_avail_blocks + (int64_t)reserved_blocks.size();
    int64_t from_free = 0;
    if (!in_shrink) {
      int64_t virtual_free = pa.num_free_pages;
      int64_t physical_free = pa.avail_phys_pages + (int64_t)pa.reserved_list.size();
      from_free = std::min(virtual_free, physical_free) *
                  OInternalPage::get_num_blocks(page_size, block_mem_size);
    }
    return avail_blocks + from_free;
  }

  // _alloc :249-304. Returns false for Python's None.
  bool alloc(int64_t need, std::vector<int64_t> &ret) {
    ret.clear();
    if (pa.resize_target > 0) resize(pa.resize_target); // :258-260
    if (available_size() < need) return false;           // :262-265
    int64_t remaining = need;
    if (!reserved_blocks.empty()) { // :272-277
      int64_t k = std::min<int64_t>((int64_t)reserved_blocks.size(), remaining);
      ret.assign(reserved_blocks.begin(), reserved_blocks.begin() + k);
      reserved_blocks.erase(reserved_blocks.begin(), reserved_blocks.begin() + k);
      remaining -= k;
    }
    while (remaining > 0) { // :279-302
      std::shared_ptr<OInternalPage> page;
      if (avail_pages.empty()) {
        int64_t pid = pa.alloc_page();
        page = std::make_shared<OInternalPage>(pid, page_size);
        page->init(block_mem_size);
        if (page->num_free_blocks() == 0) { // :287-289
          full_pages.emplace_back(pid, page);
          continue;
        }
        num_avail_blocks += page->num_free_blocks();
      } else {
        page = avail_pages.back().second; // dict.popitem()
        avail_pages.pop_back();
      }
      int64_t k = std::min(page->num_free_blocks(), remaining);
      auto got = page->alloc(k);
      ret.insert(ret.end(), got.begin(), got.end());
      if (page->full())
        full_pages.emplace_back(page->page_id, page);
      else
        avail_pages.emplace_back(page->page_id, page);
      num_avail_blocks -= k;
      remaining -= k;
    }
    return true;
  }

  // :492-506
  int64_t get_num_alloced_blocks() const {
    int64_t bpp = OInternalPage::get_num_blocks(page_size, block_mem_size);
    return (int64_t)full_pages.size() * bpp + (int64_t)avail_pages.size() * bpp - num_avail_blocks +
           (int64_t)reserved_blocks.size();
  }

  // :306-360 (SANITY_CHECK off; unknown pages are skipped, :330-338)
  void free(const int64_t *idx, size_t n) {
    if (n == 0) return;
    auto groups = pa.group_indices_by_page(idx, n, block_mem_size);
    std::vector<int64_t> pages_to_free;
    for (auto &g : groups) { // unordered_map iteration order == the Python dict's order
      int64_t pid = g.first;
      std::shared_ptr<OInternalPage> page;
      auto itf = find_page(full_pages, pid);
      if (itf != full_pages.end()) {
        page = itf->second;
        full_pages.erase(itf);
      } else {
        auto ita = find_page(avail_pages, pid);
        if (ita == avail_pages.end()) continue;
        page = ita->second;
        avail_pages.erase(ita);
      }
      num_avail_blocks += (int64_t)g.second.size();
      page->free_batch(g.second.data(), g.second.size());
      if (page->empty()) {
        pages_to_free.push_back(page->page_id);
        num_avail_blocks -= page->num_free_blocks();
      } else {
        avail_pages.emplace_back(pid, page);
      }
    }
    if (!pages_to_free.empty()) pa.free_pages(pages_to_free);
    if (in_shrink && get_num_alloced_blocks() <= target_num_blocks) { // :354-360
      pa.resize(target_num_blocks * block_mem_size);
      in_shrink = false;
      target_num_blocks = -1;
    }
  }

  // :362-372
  bool try_to_reserve(int64_t need) {
    if (available_size() < need) return false;
    std::vector<int64_t> got;
    if (!alloc(need, got)) return false;
    reserved_blocks.insert(reserved_blocks.end(), got.begin(), got.end());
    return true;
  }

  // :374-378
  void free_reserved() {
    if (!reserved_blocks.empty()) {
      std::vector<int64_t> tmp = reserved_blocks; // free() reads the list it was handed
      free(tmp.data(), tmp.size());
      reserved_blocks.clear();
    }
  }

  // :380-401. Returns -1 where the reference's assert (:394-395) would fire.
  int resize(int64_t new_mem_size) {
    if (pa.resize(new_mem_size)) {
      if (in_shrink) {
        in_shrink = false;
        target_num_blocks = -1;
      }
      return 1;
    }
    if (!reserved_blocks.empty()) return -1;
    in_shrink = true;
    target_num_blocks = new_mem_size / block_mem_size;
    free_reserved();
    return 0;
  }

  void trim() { pa.trim(); } // :403-409

  // clear() :443-489 with the evident intent (the snapshot calls a method name the binding
  // does not export and raises AttributeError; see DESIGN.md "reference quirks").
  void clear() {
    free_reserved();
    std::vector<int64_t> pages_to_free;
    for (auto &kv : avail_pages) pages_to_free.push_back(kv.second->page_id);
    for (auto &kv : full_pages) pages_to_free.push_back(kv.second->page_id);
    if (!pages_to_free.empty()) pa.free_pages(pages_to_free);
    avail_pages.clear();
    full_pages.clear();
    trim();
    pa.reset_free_page_order();
    target_num_blocks = -1;
    in_shrink = false;
    num_avail_blocks = 0;
    post_init();
  }
};

} // namespace okvc

// =============================================================== C ABI (ctypes)
using namespace okvc;

static thread_local std::string g_err;
#define GUARD(default_ret, body)                                                                   \
  try {                                                                                            \
    body                                                                                           \
  } catch (const std::exception &e) {                                                              \
    g_err = e.what();                                                                              \
    return default_ret;                                                                            \
  }

static int64_t copy_out(const std::vector<int64_t> &v, int64_t *out, int64_t cap) {
  int64_t n = (int64_t)v.size();
  if (out && cap >= n) std::memcpy(out, v.data(), (size_t)n * 8);
  return n;
}
static int64_t copy_out(const std::deque<int64_t> &v, int64_t *out, int64_t cap) {
  int64_t n = (int64_t)v.size();
  if (out && cap >= n) std::copy(v.begin(), v.end(), out);
  return n;
}

extern "C" {

const char *okvc_last_error() { return g_err.c_str(); }

// ---- InternalPage statics
void okvc_get_block_range(int64_t pid, int64_t P, int64_t B, int64_t *start, int64_t *end) {
  auto r = OInternalPage::get_block_range(pid, P, B);
  *start = r.first;
  *end = r.second;
}
int64_t okvc_get_num_blocks(int64_t P, int64_t B) { return OInternalPage::get_num_blocks(P, B); }

// ---- InternalPage object
void *okvc_page_new(int64_t pid, int64_t P) { return new OInternalPage(pid, P); }
void okvc_page_delete(void *p) { delete (OInternalPage *)p; }
void okvc_page_init(void *p, int64_t B) { ((OInternalPage *)p)->init(B); }
int64_t okvc_page_alloc(void *p, int64_t n, int64_t *out) {
  GUARD(-1, {
    auto v = ((OInternalPage *)p)->alloc(n);
    return copy_out(v, out, n);
  })
}
void okvc_page_free(void *p, int64_t b) { ((OInternalPage *)p)->free_one(b); }
void okvc_page_free_batch(void *p, const int64_t *ids, int64_t n) { ((OInternalPage *)p)->free_batch(ids, (size_t)n); }
int okvc_page_empty(void *p) { return ((OInternalPage *)p)->empty(); }
int okvc_page_full(void *p) { return ((OInternalPage *)p)->full(); }
int64_t okvc_page_num_free(void *p) { return ((OInternalPage *)p)->num_free_blocks(); }
int64_t okvc_page_free_blocks(void *p, int64_t *out, int64_t cap) { return copy_out(((OInternalPage *)p)->free_list, out, cap); }

// ---- PageAllocator
void *okvc_pa_new(int64_t L, int64_t mem, int64_t P, int64_t ws, int contig, int prealloc, int64_t kv,
                  int64_t min_res, int64_t max_res) {
  return new OPageAllocator(L, mem, P, ws, contig != 0, prealloc != 0, kv, min_res, max_res);
}
void okvc_pa_delete(void *h) { delete (OPageAllocator *)h; }
int64_t okvc_pa_alloc_page(void *h) { GUARD(-1, { return ((OPageAllocator *)h)->alloc_page(); }) }
void okvc_pa_free_page(void *h, int64_t pid) { ((OPageAllocator *)h)->free_page(pid); }
void okvc_pa_free_pages(void *h, const int64_t *p, int64_t n) { ((OPageAllocator *)h)->free_pages(std::vector<int64_t>(p, p + n)); }
int okvc_pa_resize(void *h, int64_t m) { return ((OPageAllocator *)h)->resize(m); }
void okvc_pa_trim(void *h) { ((OPageAllocator *)h)->trim(); }
void okvc_pa_reset_free_page_order(void *h) { ((OPageAllocator *)h)->reset_free_page_order(); }
int64_t okvc_pa_prealloc_step(void *h) { return ((OPageAllocator *)h)->prealloc_step(); }
void okvc_pa_set_prealloc_needed(void *h, int v) { ((OPageAllocator *)h)->prealloc_needed = v != 0; }
// The reference's get_avail_physical_pages arithmetic (csrc/page_allocator.cpp:442-455) on a given hipMemGetInfo reading,
// restated with its types: size_t subtraction (WRAPS when free < headroom - std::max(x - y, 0) on unsigned values is a no-op),
// then int64 divisions. Pinned by tests/golden/avail_physical_pages.json (the real reference on fed readings).
int64_t okvc_ref_avail_physical_pages(uint64_t free_b, uint64_t total_b, double gpu_utilization, int64_t page_size,
                                      int64_t num_layers, int64_t num_kv_buffers) {
  size_t avail = (size_t)free_b;
  const size_t headroom = (size_t)((double)total_b * (1.0 - gpu_utilization));
  avail = std::max(avail - headroom, static_cast<size_t>(0));
  const int64_t avail_phy_pages = (int64_t)(avail / (size_t)page_size);
  return avail_phy_pages / num_layers / num_kv_buffers;
}
void okvc_pa_set_avail_phys_pages(void *h, int64_t v) { ((OPageAllocator *)h)->avail_phys_pages = v; }
void okvc_pa_set_shm_total(void *h, int64_t v) { ((OPageAllocator *)h)->shm_total = v; }
void okvc_pa_watcher_tick(void *h) { ((OPageAllocator *)h)->watcher_tick(); }
int64_t okvc_pa_get_resize_target(void *h) { return ((OPageAllocator *)h)->resize_target; }
int64_t okvc_pa_get_page_id(void *h, int64_t b, int64_t B) { return ((OPageAllocator *)h)->get_page_id(b, B); }
// stats: [free, inuse, total, reserved, shm_total, shm_used, shm_prealloc]
void okvc_pa_stats(void *h, int64_t *o) {
  auto *a = (OPageAllocator *)h;
  o[0] = a->num_free_pages;
  o[1] = a->get_num_inuse_pages();
  o[2] = a->num_total_pages;
  o[3] = (int64_t)a->reserved_list.size();
  o[4] = a->shm_total;
  o[5] = a->shm_used;
  o[6] = a->shm_prealloc;
}
// which: 0 free, 1 reserved, 2 reclaimed
int64_t okvc_pa_list(void *h, int which, int64_t *out, int64_t cap) {
  auto *a = (OPageAllocator *)h;
  return copy_out(which == 0 ? a->free_list : which == 1 ? a->reserved_list : a->reclaimed_list, out, cap);
}
// group_indices_by_page flattened in ITERATION ORDER: keys[k], counts[k], values concatenated.
int64_t okvc_pa_group_indices(void *h, const int64_t *idx, int64_t n, int64_t B, int64_t *keys,
                              int64_t *counts, int64_t *values) {
  auto g = ((OPageAllocator *)h)->group_indices_by_page(idx, (size_t)n, B);
  int64_t k = 0, w = 0;
  for (auto &kv : g) {
    keys[k] = kv.first;
    counts[k] = (int64_t)kv.second.size();
    for (int64_t v : kv.second) values[w++] = v;
    ++k;
  }
  return k;
}
// event log: returns number of events; drains into flat arrays when out != NULL.
// layout: for each event: kind, n, offsets[n]
int64_t okvc_pa_drain_log(void *h, int64_t *out, int64_t cap) {
  auto *a = (OPageAllocator *)h;
  int64_t need = 0;
  for (auto &e : a->log) need += 2 + (int64_t)e.offsets.size();
  if (!out || cap < need) return need;
  int64_t w = 0;
  for (auto &e : a->log) {
    out[w++] = e.kind;
    out[w++] = (int64_t)e.offsets.size();
    for (int64_t o : e.offsets) out[w++] = o;
  }
  a->log.clear();
  return need;
}

// ---- KVCacheManager
void *okvc_mgr_new(int64_t nb, int64_t block_size, int64_t cell_size, int64_t L, int64_t ws, int null_blk,
                   int64_t kv, int64_t P, int contig, int prealloc, int64_t min_res, int64_t max_res) {
  return new OKVCacheManager(nb, block_size, cell_size, L, ws, null_blk != 0, kv, P, contig != 0,
                             prealloc != 0, min_res, max_res);
}
void okvc_mgr_delete(void *h) { delete (OKVCacheManager *)h; }
void *okvc_mgr_pa(void *h) { return &((OKVCacheManager *)h)->pa; }
int okvc_mgr_post_init(void *h) {
  auto *m = (OKVCacheManager *)h;
  GUARD(-1, {
    m->post_init();
    return m->null_block_ok ? 0 : 1;
  })
}
// returns count, -1 for None, -2 on exception
int64_t okvc_mgr_alloc(void *h, int64_t need, int64_t *out) {
  GUARD(-2, {
    std::vector<int64_t> r;
    if (!((OKVCacheManager *)h)->alloc(need, r)) return -1;
    return copy_out(r, out, need);
  })
}
int okvc_mgr_free(void *h, const int64_t *idx, int64_t n) {
  GUARD(-2, {
    ((OKVCacheManager *)h)->free(idx, (size_t)n);
    return 0;
  })
}
int okvc_mgr_try_to_reserve(void *h, int64_t n) { GUARD(-2, { return ((OKVCacheManager *)h)->try_to_reserve(n); }) }
int okvc_mgr_free_reserved(void *h) {
  GUARD(-2, {
    ((OKVCacheManager *)h)->free_reserved();
    return 0;
  })
}
int okvc_mgr_resize(void *h, int64_t m) { GUARD(-2, { return ((OKVCacheManager *)h)->resize(m); }) }
void okvc_mgr_trim(void *h) { ((OKVCacheManager *)h)->trim(); }
int okvc_mgr_clear(void *h) {
  GUARD(-2, {
    ((OKVCacheManager *)h)->clear();
    return 0;
  })
}
int64_t okvc_mgr_available_size(void *h) { return ((OKVCacheManager *)h)->available_size(); }
int64_t okvc_mgr_reserved_blocks(void *h, int64_t *out, int64_t cap) { return copy_out(((OKVCacheManager *)h)->reserved_blocks, out, cap); }
// [num_avail_blocks, n_avail_pages, n_full_pages, n_reserved_blocks, in_shrink, target_num_blocks, alloced_blocks]
void okvc_mgr_stats(void *h, int64_t *o) {
  auto *m = (OKVCacheManager *)h;
  o[0] = m->num_avail_blocks;
  o[1] = (int64_t)m->avail_pages.size();
  o[2] = (int64_t)m->full_pages.size();
  o[3] = (int64_t)m->reserved_blocks.size();
  o[4] = m->in_shrink;
  o[5] = m->target_num_blocks;
  o[6] = m->get_num_alloced_blocks();
}

// ---- north-star kernels, CPU statement
// zero_fill_pages: every listed page becomes page_bytes of 0x00.
void okvc_zero_fill_pages(void *const *pages, int64_t n, int64_t page_bytes) {
  for (int64_t i = 0; i < n; ++i) std::memset(pages[i], 0, (size_t)page_bytes);
}
// compact_blocks: for every region base r (one per layer x K/V buffer) and every move m:
//   memcpy(base[r] + dst[m]*block_bytes, base[r] + src[m]*block_bytes, block_bytes)
// Moves are independent (src and dst sets are disjoint; the planner guarantees it).
void okvc_compact_blocks(void *const *region_bases, int64_t n_regions, const int64_t *src, const int64_t *dst,
                         int64_t n_moves, int64_t block_bytes) {
  for (int64_t r = 0; r < n_regions; ++r)
    for (int64_t m = 0; m < n_moves; ++m)
      std::memcpy((char *)region_bases[r] + dst[m] * block_bytes, (const char *)region_bases[r] + src[m] * block_bytes,
                  (size_t)block_bytes);
}

} // extern "C"
