// mem_info.hpp — the /dev/shm/<ipc_name> usage segment that kvctl / kvtop and co-located
// engines read and write.
//
// Wire format is the reference's, byte for byte (csrc/inc/mem_info_tracker.hpp:25-36,
// kvcached/cli/utils.py): int64 little-endian {total_size, used_size, prealloc_size},
// 24 bytes, file mode 0666, writers of total_size hold flock(LOCK_EX).
//
// What differs is the access pattern. The reference re-opens, flock()s, mmap()s, munmap()s and
// closes the file on EVERY page event (5 syscalls; SURVEY §8 a13 measured it as the dominant
// cost of the allocator fast path). Here the segment is mapped once; used_size / prealloc_size
// are published with two relaxed 8-byte atomic stores and total_size is read with one atomic
// load. This is safe with stock kvctl because the fields have single writers: only the engine
// writes [1] and [2], only the controller writes [0] after creation, and an aligned 8-byte
// store cannot tear; a controller's read-modify-write that races with one of our stores and writes a
// stale used/prealloc back is repaired by the next page event or, at the latest, the next 100 ms
// watcher tick (revalidate()). The creation write (all three fields) still takes flock(LOCK_EX) like the
// reference. If the file is deleted or replaced underneath us (kvctl delete), revalidate()
// notices the inode change and re-creates it, as the reference's open-by-path would.
#pragma once

#include <fcntl.h>
#include <sys/file.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <atomic>
#include <cstdint>
#include <mutex>
#include <string>

#include "common.hpp"

namespace kvc {

class MemInfoTracker {
public:
  static constexpr size_t kShmSize = 3 * sizeof(int64_t);

  MemInfoTracker(int64_t total_mem_size, int64_t group_id, const std::string &ipc_name)
      : ipc_name_(ipc_name), total_mem_size_(total_mem_size) {
    if (ipc_name_.empty()) {
      ipc_name_ = default_ipc_name();
      if (group_id != 0) ipc_name_ += "_g" + std::to_string(group_id); // mem_info_tracker.hpp:160-164
    }
    path_ = ipc_name_[0] == '/' ? ipc_name_ : std::string("/dev/shm/") + ipc_name_;
    open_segment(/*init=*/true);
  }
  ~MemInfoTracker() {
    close_segment();
    ::unlink(path_.c_str()); // mem_info_tracker.hpp:222-225
  }

  const std::string &ipc_name() const { return ipc_name_; }

  void update_memory_usage(int64_t used, int64_t prealloc) {
    std::lock_guard<std::mutex> g(mu_);
    pub_used_ = used;
    pub_prealloc_ = prealloc;
    if (!arr_) return;
    __atomic_store_n(&arr_[1], used, __ATOMIC_RELAXED);
    __atomic_store_n(&arr_[2], prealloc, __ATOMIC_RELAXED);
  }

  // mem_info_tracker.hpp:191-204
  int64_t check_and_get_resize_target(int64_t current_mem_size, int64_t num_layers, int64_t num_kv_buffers) {
    std::lock_guard<std::mutex> g(mu_);
    if (!arr_) return -1;
    int64_t total = __atomic_load_n(&arr_[0], __ATOMIC_RELAXED);
    int64_t new_mem = total / num_layers / num_kv_buffers;
    return new_mem != current_mem_size ? new_mem : -1;
  }

  // Called from the 10 Hz watcher: if our inode is no longer the file at path_, map the new one
  // (creating it with our last known values if it vanished).
  void revalidate() {
    std::lock_guard<std::mutex> g(mu_);
    struct stat now {};
    if (::stat(path_.c_str(), &now) == 0 && now.st_ino == ino_ && now.st_dev == dev_) {
      // kvctl's limit update is a read-modify-write of all three fields under flock: racing with one of our
      // lock-free stores it can write a stale used/prealloc back. We are their only writer, so re-assert.
      if (arr_ && __atomic_load_n(&arr_[1], __ATOMIC_RELAXED) != pub_used_)
        __atomic_store_n(&arr_[1], pub_used_, __ATOMIC_RELAXED);
      if (arr_ && __atomic_load_n(&arr_[2], __ATOMIC_RELAXED) != pub_prealloc_)
        __atomic_store_n(&arr_[2], pub_prealloc_, __ATOMIC_RELAXED);
      return;
    }
    close_segment();
    open_segment(/*init=*/false);
    if (arr_) {
      if (arr_[0] == 0) arr_[0] = total_mem_size_;
      arr_[1] = pub_used_;
      arr_[2] = pub_prealloc_;
    }
  }

private:
  static std::string default_ipc_name() { // mem_info_tracker.hpp:228-240
    const char *env = std::getenv("KVCACHED_IPC_NAME");
    if (env && env[0] != '\0') return env;
    return "kvcached_engine_" + std::to_string((int)getpgrp());
  }

  void open_segment(bool init) {
    fd_ = ::open(path_.c_str(), O_RDWR | O_CREAT, 0666);
    if (fd_ < 0) {
      KVC_LOG(LOG_ERROR, "MemInfoTracker: failed to create shm: %s", path_.c_str());
      return;
    }
    struct stat st {};
    if (fstat(fd_, &st) == 0 && (size_t)st.st_size < kShmSize) (void)ftruncate(fd_, kShmSize);
    (void)fstat(fd_, &st);
    ino_ = st.st_ino;
    dev_ = st.st_dev;
    void *p = mmap(nullptr, kShmSize, PROT_READ | PROT_WRITE, MAP_SHARED, fd_, 0);
    if (p == MAP_FAILED) {
      KVC_LOG(LOG_ERROR, "MemInfoTracker: mmap failed for %s", path_.c_str());
      ::close(fd_);
      fd_ = -1;
      return;
    }
    arr_ = static_cast<int64_t *>(p);
    if (init) { // init_kv_cache_limit, mem_info_tracker.hpp:210-219
      (void)flock(fd_, LOCK_EX);
      arr_[0] = total_mem_size_;
      arr_[1] = 0;
      arr_[2] = 0;
      (void)flock(fd_, LOCK_UN);
    }
  }
  void close_segment() {
    if (arr_) munmap(arr_, kShmSize);
    arr_ = nullptr;
    if (fd_ >= 0) ::close(fd_);
    fd_ = -1;
  }

  std::string ipc_name_, path_;
  int64_t total_mem_size_;
  int64_t pub_used_ = 0, pub_prealloc_ = 0; // what this engine last published (guarded by mu_)
  std::mutex mu_;
  int fd_ = -1;
  int64_t *arr_ = nullptr;
  ino_t ino_ = 0;
  dev_t dev_ = 0;
};

} // namespace kvc
