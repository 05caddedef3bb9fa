// run_scan.hpp — the slots of a batch as runs of neighbours, without sorting the batch. No HIP or driver headers: exercised
// on the CPU by tests/native/run_scan_check.cpp.
#pragma once

#include <algorithm>
#include <cstddef>
#include <cstdint>
#include <vector>

namespace kvc {

// The slots of a batch -> maximal runs of neighbours per region, in address order, without sorting the batch: each slot
// sets its bit in the region's scratch bitmap (the region's `mark`: std::vector<uint64_t>, all clear at rest) and the runs are read back from the
// words in between. O(n + span / 64) against the 30 us a sort of 1024 shuffled slots costs. Needs the allocator's lock.
template <class Region> struct SlotRunOf {
  Region *r;
  size_t first, count;
};
template <class Region> class RunScanOf {
public:
  using Run = SlotRunOf<Region>;
  ~RunScanOf() { wipe(); } // (an exception between add() and collect(): the marks must not survive)
  // false: the slot is already part of this scan (listed twice in one call)
  bool add(Region *r, size_t index) {
    if (last_ == kNone || seen_[last_].r != r) {
      last_ = kNone;
      for (size_t i = 0; i < seen_.size(); ++i)
        if (seen_[i].r == r) last_ = i;
      if (last_ == kNone) {
        last_ = seen_.size();
        seen_.push_back({r, index >> 6, index >> 6});
      }
    }
    Seen &t = seen_[last_];
    uint64_t &w = r->mark[index >> 6];
    const uint64_t bit = 1ull << (index & 63);
    if (w & bit) return false;
    w |= bit;
    t.lo = std::min(t.lo, index >> 6);
    t.hi = std::max(t.hi, index >> 6);
    ++n_;
    return true;
  }
  size_t size() const { return n_; }
  // Regions in the order they first appeared, runs in ascending address order; a run never crosses a multiple of
  // `group(region)` slots nor exceeds `max_len`. The marks are clear again afterwards.
  template <class G> std::vector<Run> collect(G &&group, size_t max_len = (size_t)-1) {
    std::vector<Run> out;
    auto emit = [&](Region *r, size_t first, size_t count) {
      const size_t g = group(*r);
      while (count) {
        size_t take = std::min(count, max_len);
        if (g != (size_t)-1) take = std::min(take, g - first % g);
        out.push_back({r, first, take});
        first += take;
        count -= take;
      }
    };
    for (auto &t : seen_) {
      size_t first = 0, len = 0;
      for (size_t w = t.lo; w <= t.hi; ++w) {
        uint64_t m = t.r->mark[w];
        t.r->mark[w] = 0;
        while (m) {
          const unsigned b = (unsigned)__builtin_ctzll(m);
          const uint64_t rest = ~(m >> b); // zeros where the run of ones that starts at bit b goes on
          const unsigned ones = rest ? (unsigned)__builtin_ctzll(rest) : 64u;
          const unsigned run = std::min(ones, 64u - b);
          const size_t idx = w * 64 + b;
          if (len && first + len == idx) {
            len += run;
          } else {
            if (len) emit(t.r, first, len);
            first = idx;
            len = run;
          }
          m = b + run >= 64 ? 0 : m & ~(((1ull << run) - 1) << b);
        }
      }
      if (len) emit(t.r, first, len);
    }
    seen_.clear();
    last_ = kNone;
    n_ = 0;
    return out;
  }
  std::vector<Run> collect() {
    return collect([](const Region &) { return (size_t)-1; });
  }

private:
  struct Seen {
    Region *r;
    size_t lo, hi; // words of r->mark that may hold bits
  };
  static constexpr size_t kNone = (size_t)-1;
  void wipe() {
    for (auto &t : seen_)
      for (size_t w = t.lo; w <= t.hi; ++w) t.r->mark[w] = 0;
    seen_.clear();
  }
  std::vector<Seen> seen_;
  size_t last_ = kNone, n_ = 0;
};


} // namespace kvc
