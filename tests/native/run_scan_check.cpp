// run_scan_check.cpp — kvcached_amd/csrc/run_scan.hpp (slots of a batch -> runs of neighbours, no sort) and KeyGroups of
// extent_pool.hpp (keys of a batch -> groups, no sort) against the sort-based statement of the same thing, on the CPU.
// Built and run by tests/test_run_scan.py (g++ with ASan/UBSan). Exits non-zero on the first difference.
#include <cstdio>
#include <cstdlib>
#include <map>
#include <random>
#include <set>
#include <tuple>
#include <vector>

#include "../../kvcached_amd/csrc/extent_pool.hpp"
#include "../../kvcached_amd/csrc/run_scan.hpp"

using namespace kvc;

#define REQUIRE(cond, ...)                                                                                             \
  do {                                                                                                                 \
    if (!(cond)) {                                                                                                     \
      fprintf(stderr, "FAILED %s:%d %s: ", __FILE__, __LINE__, #cond);                                                 \
      fprintf(stderr, __VA_ARGS__);                                                                                    \
      fprintf(stderr, "\n");                                                                                           \
      exit(1);                                                                                                         \
    }                                                                                                                  \
  } while (0)

struct Region {
  std::vector<uint64_t> mark;
  size_t group = (size_t)-1;
  explicit Region(size_t slots) : mark((slots + 63) / 64, 0) {}
};
using Scan = RunScanOf<Region>;
using Run = SlotRunOf<Region>;

// what the scan replaces: sort (region in order of first appearance, index), drop repeats, cut into runs
static std::vector<std::tuple<int, size_t, size_t>> by_sorting(const std::vector<std::pair<int, size_t>> &batch, const std::vector<Region *> &regions,
                                                               size_t max_len, size_t *repeats) {
  std::vector<int> order;
  std::map<int, std::set<size_t>> per;
  *repeats = 0;
  for (auto &s : batch) {
    if (!per.count(s.first)) order.push_back(s.first);
    if (!per[s.first].insert(s.second).second) ++*repeats;
  }
  std::vector<std::tuple<int, size_t, size_t>> out;
  for (int ri : order) {
    const size_t g = regions[ri]->group;
    size_t first = 0, len = 0;
    auto emit = [&]() {
      while (len) {
        size_t take = std::min(len, max_len);
        if (g != (size_t)-1) take = std::min(take, g - first % g);
        out.emplace_back(ri, first, take);
        first += take;
        len -= take;
      }
    };
    for (size_t idx : per[ri]) {
      if (len && first + len == idx) {
        ++len;
      } else {
        emit();
        first = idx;
        len = 1;
      }
    }
    emit();
  }
  return out;
}

int main() {
  std::mt19937_64 rng(7);
  size_t cases = 0, runs_total = 0;
  for (int round = 0; round < 4000; ++round) {
    const size_t slots = (round % 5 == 0) ? 64 : (round % 5 == 1) ? 65 : (round % 5 == 2) ? 1000 : 32768;
    const int nreg = 1 + (int)(rng() % 4);
    std::vector<Region> store;
    store.reserve(nreg);
    for (int i = 0; i < nreg; ++i) store.emplace_back(slots);
    std::vector<Region *> regions;
    for (auto &r : store) regions.push_back(&r);
    if (round % 3 == 0)
      for (auto *r : regions) r->group = (size_t[]){1, 2, 64, 100}[rng() % 4];
    const size_t max_len = (round % 7 == 0) ? 1 + rng() % 70 : (size_t)-1;
    // batches shaped like the callers': a shuffled window, scattered singles, everything, with a few repeats
    std::vector<std::pair<int, size_t>> batch;
    const int shape = (int)(rng() % 4);
    for (int ri = 0; ri < nreg; ++ri) {
      if (shape == 0) { // a window of up to 1024 neighbours
        const size_t n = std::min<size_t>(slots, 1 + rng() % 1024), base = rng() % (slots - n + 1);
        for (size_t i = 0; i < n; ++i) batch.emplace_back(ri, base + i);
      } else if (shape == 1) { // scattered
        for (size_t i = 0, n = 1 + rng() % 300; i < n; ++i) batch.emplace_back(ri, rng() % slots);
      } else if (shape == 2) { // every slot
        for (size_t i = 0; i < slots; ++i) batch.emplace_back(ri, i);
      } else { // edges of the bitmap words
        for (size_t w = 0; w * 64 < slots; w += 1 + rng() % 3)
          for (size_t b : {(size_t)0, (size_t)62, (size_t)63})
            if (w * 64 + b < slots && rng() % 4) batch.emplace_back(ri, w * 64 + b);
        batch.emplace_back(ri, slots - 1);
      }
    }
    if (rng() % 2) std::shuffle(batch.begin(), batch.end(), rng); // (shuffling interleaves the regions too)
    size_t want_repeats = 0;
    const auto want = by_sorting(batch, regions, max_len, &want_repeats);
    Scan scan;
    size_t repeats = 0;
    for (auto &s : batch) repeats += !scan.add(regions[s.first], s.second);
    REQUIRE(repeats == want_repeats, "round %d: %zu repeats reported, %zu listed", round, repeats, want_repeats);
    REQUIRE(scan.size() == batch.size() - repeats, "size()");
    const std::vector<Run> got = scan.collect([](const Region &r) { return r.group; }, max_len);
    REQUIRE(got.size() == want.size(), "round %d (shape %d): %zu runs, sorting gives %zu", round, shape, got.size(), want.size());
    for (size_t i = 0; i < got.size(); ++i)
      REQUIRE(got[i].r == regions[std::get<0>(want[i])] && got[i].first == std::get<1>(want[i]) && got[i].count == std::get<2>(want[i]),
              "round %d run %zu: (%zu, %zu) against (%zu, %zu)", round, i, got[i].first, got[i].count, std::get<1>(want[i]), std::get<2>(want[i]));
    for (auto *r : regions)
      for (uint64_t w : r->mark) REQUIRE(w == 0, "round %d: marks left behind", round);
    REQUIRE(scan.size() == 0 && scan.collect().empty(), "a collected scan is empty");
    runs_total += got.size();
    ++cases;
  }
  { // a scan that is dropped before collect() (an exception on the way) leaves no marks
    Region r(300);
    {
      Scan scan;
      for (size_t i : {(size_t)0, (size_t)63, (size_t)64, (size_t)299}) scan.add(&r, i);
    }
    for (uint64_t w : r.mark) REQUIRE(w == 0, "marks survive a dropped scan");
  }
  // KeyGroups: counts per key, in order of first appearance, whatever the keys look like (KFD handles, ROCr pointers)
  size_t group_cases = 0;
  for (int round = 0; round < 2000; ++round) {
    const size_t n = 1 + rng() % 1500, distinct = 1 + rng() % (round % 2 ? 40 : n);
    std::vector<uint64_t> keys(distinct);
    for (auto &k : keys) k = (round % 3 == 0) ? (rng() | (1ull << 63)) : (round % 3 == 1) ? (0x2a00000000ull | (rng() % 100000)) : (rng() % 64) << 20;
    std::vector<uint64_t> batch(n);
    for (auto &k : batch) k = keys[rng() % distinct];
    std::vector<uint64_t> order;
    std::map<uint64_t, uint32_t> want;
    for (uint64_t k : batch)
      if (!want[k]++) order.push_back(k);
    KeyGroups<uint32_t> groups(n);
    for (uint64_t k : batch) ++groups.at(k);
    REQUIRE(groups.items().size() == order.size(), "round %d: %zu groups for %zu keys", round, groups.items().size(), order.size());
    for (size_t i = 0; i < order.size(); ++i)
      REQUIRE(groups.items()[i].first == order[i] && groups.items()[i].second == want[order[i]], "round %d: group %zu", round, i);
    ++group_cases;
  }
  printf("{\"scan_cases\": %zu, \"runs\": %zu, \"group_cases\": %zu}\n", cases, runs_total, group_cases);
  return 0;
}
