// extent_pool_check.cpp — the placement policy of kvcached_amd/csrc/extent_pool.hpp against a fake driver, on the CPU.
// Built and run by tests/test_extent_pool.py (g++, no HIP). Prints one JSON object; exits non-zero on a broken invariant.
#include <cstdio>
#include <cstdlib>
#include <map>
#include <random>
#include <set>
#include <string>
#include <vector>

#include "../../kvcached_amd/csrc/extent_pool.hpp"

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

struct FakeDriver {
  std::map<phys_handle_t, unsigned> live; // extent -> pages
  uint64_t next = 0x1000;
  uint64_t high_bits = 0; // single-page pools: handles like ROCr's, with bits above 48 set
  size_t creates = 0, releases = 0, before_release_calls = 0, fail_after = (size_t)-1;
  bool pressure = false, give_tags = false;
  size_t prepared = 0;
  ExtentPool *pool = nullptr;
  bool check_no_waste_at_create = true;
  ExtentDriver make() {
    ExtentDriver d;
    d.prepare_release = [this](phys_handle_t h, uint64_t tag, size_t pages) {
      REQUIRE(live.count(h) && live[h] == pages && tag == (give_tags ? 0x7000000000ull + (h & 0xffffffffull) * 0x100000 : 0), "prepare_release of the right extent");
      ++prepared;
    };
    d.create = [this](size_t pages, uint64_t *tag) -> phys_handle_t {
      if (creates >= fail_after) throw std::runtime_error("out of memory [injected]");
      if (pool && check_no_waste_at_create) {
        auto f = pool->footprint();
        REQUIRE(f.free_pieces == 0 && f.idle_pages == 0, "created %zu pages while %zu free pieces and %zu idle pages existed",
                pages, f.free_pieces, f.idle_pages);
      }
      ++creates;
      const phys_handle_t h = next | high_bits;
      next += 0x10;
      live[h] = (unsigned)pages;
      if (tag) *tag = give_tags ? 0x7000000000ull + (h & 0xffffffffull) * 0x100000 : 0;
      return h;
    };
    d.release = [this](phys_handle_t h) {
      auto it = live.find(h);
      REQUIRE(it != live.end(), "release of an unknown extent %llx", (unsigned long long)h);
      live.erase(it);
      ++releases;
      return true;
    };
    d.before_release = [this] { ++before_release_calls; };
    d.under_pressure = [this] { return pressure; };
    return d;
  }
  size_t live_pages() const {
    size_t n = 0;
    for (auto &kv : live) n += kv.second;
    return n;
  }
};

// what the allocator does with the pool: back a sorted set of slots run by run, give pages back
struct Harness {
  FakeDriver drv;
  VmmCounters ctr;
  ExtentPool pool;
  std::map<int64_t, Phys> slot;             // slot -> piece
  std::set<phys_handle_t> out;              // pieces handed out
  size_t map_ioctls = 0, pages_mapped = 0;
  Harness(unsigned kmax, size_t cap_pages, double waste) : pool(2u << 20, kmax, drv.make(), &ctr) {
    drv.pool = &pool;
    pool.set_cap_bytes(cap_pages * (2u << 20));
    pool.set_waste_frac(waste);
  }
  void map(std::vector<int64_t> ids) {
    std::sort(ids.begin(), ids.end());
    std::vector<Phys> got(kMaxExtentPages);
    for (size_t i = 0; i < ids.size();) {
      size_t j = i + 1;
      while (j < ids.size() && ids[j] == ids[j - 1] + 1) ++j;
      while (i < j) {
        bool rec = false;
        size_t n = pool.acquire_run(j - i, got.data(), &rec, false);
        if (n == 0) n = pool.acquire_run(j - i, got.data(), &rec, true);
        REQUIRE(n >= 1 && n <= j - i, "acquire_run(%zu) returned %zu", j - i, n);
        // (only a pool that makes multi-page extents encodes anything into its handles)
        const bool multi = pool.multi_page();
        auto ext_of = [&](phys_handle_t h) { return multi ? chunk_of(h) : h; };
        auto idx_of = [&](phys_handle_t h) { return multi ? piece_of(h) : 0u; };
        const phys_handle_t ext = ext_of(got[0].h);
        for (size_t k = 0; k < n; ++k) {
          REQUIRE(ext_of(got[k].h) == ext, "a run spans two extents");
          REQUIRE(idx_of(got[k].h) == idx_of(got[0].h) + k, "pieces of a run are not consecutive");
          REQUIRE(drv.live.count(ext), "piece of an extent the driver does not know");
          REQUIRE((multi ? pages_of(got[k].h) : 1u) == drv.live[ext] && idx_of(got[k].h) < drv.live[ext], "piece index / size encoding");
          REQUIRE(out.insert(got[k].h).second, "piece %llx handed out twice", (unsigned long long)got[k].h);
          REQUIRE(!slot.count(ids[i + k]), "slot mapped twice by the harness");
          slot[ids[i + k]] = got[k];
        }
        ++map_ioctls;
        pages_mapped += n;
        i += n;
      }
    }
    check();
  }
  void unmap(const std::vector<int64_t> &ids) {
    std::vector<Phys> back;
    for (auto id : ids) {
      auto it = slot.find(id);
      REQUIRE(it != slot.end(), "unmap of an unmapped slot");
      back.push_back(it->second);
      out.erase(it->second.h);
      slot.erase(it);
    }
    pool.release_batch(back.data(), back.size());
    check();
  }
  void check() {
    auto f = pool.footprint();
    REQUIRE(f.out_pages == out.size(), "out %zu != %zu", f.out_pages, out.size());
    REQUIRE(f.held_pages == drv.live_pages(), "held %zu != driver's %zu", f.held_pages, drv.live_pages());
    REQUIRE(f.held_pages == f.out_pages + f.idle_pages + f.free_pieces, "held %zu != out %zu + idle %zu + free pieces %zu",
            f.held_pages, f.out_pages, f.idle_pages, f.free_pieces);
    REQUIRE(f.bad_releases == 0, "bad releases");
    REQUIRE((size_t)(ctr.created - ctr.released) == f.held_pages, "counters: created - released != held");
  }
};

static double pct(std::vector<double> v, double q) {
  if (v.empty()) return 0;
  std::sort(v.begin(), v.end());
  return v[std::min(v.size() - 1, (size_t)(v.size() * q))];
}

int main() {
  std::string js = "{";
  // 1. the bench's shape: batches of 1024 adjacent slots, mapped then unmapped, pool large enough
  {
    Harness h(64, 8192, 0.05);
    for (int rep = 0; rep < 6; ++rep)
      for (int b = 0; b < 4; ++b) {
        std::vector<int64_t> ids;
        for (int i = 0; i < 1024; ++i) ids.push_back(b * 1024 + i);
        h.map(ids);
        REQUIRE(h.pool.footprint().free_pieces == 0, "whole runs leave no waste");
        h.unmap(ids);
      }
    REQUIRE(h.drv.creates == 16, "16 extents of 64 pages serve every batch, %zu were created", h.drv.creates);
    REQUIRE(h.map_ioctls == 6 * 4 * 16, "one ioctl per 64 pages: %zu", h.map_ioctls);
    REQUIRE(h.ctr.reused == (6 * 4 - 1) * 1024, "reuse counter %lld", (long long)h.ctr.reused.load());
    h.pool.drain(0);
    REQUIRE(h.drv.live.empty(), "drain(0) leaves nothing");
    js += "\"bench_shape\": {\"creates\": 16, \"map_ioctls_per_page\": " + std::to_string(h.map_ioctls / (24.0 * 1024)) + "}";
  }
  // 2. cap: what exceeds it goes back, oldest first, in one batch with one before_release call
  {
    Harness h(16, 32, 0.05);
    std::vector<int64_t> ids;
    for (int i = 0; i < 64; ++i) ids.push_back(i);
    h.map(ids); // 4 extents of 16
    h.unmap(ids);
    auto f = h.pool.footprint();
    REQUIRE(f.idle_pages == 32 && f.held_pages == 32, "cap of 32 pages: %zu idle", f.idle_pages);
    REQUIRE(h.drv.before_release_calls == 1 && h.drv.releases == 2, "one hook call for the batch");
    REQUIRE(h.drv.live.begin()->first == 0x1020, "the oldest extents were the ones released");
    // deferred eviction: nothing released on the caller's path, trim_to_cap does it
    h.pool.set_defer_eviction(true);
    h.map(ids);
    h.unmap(ids);
    REQUIRE(h.pool.footprint().idle_pages == 64, "deferred: all four stay");
    REQUIRE(h.pool.trim_to_cap(16) == 16 && h.pool.footprint().idle_pages == 48, "trim_to_cap is bounded per call");
    h.pool.trim_to_cap(1000);
    REQUIRE(h.pool.footprint().idle_pages == 32, "down to the cap");
    // pressure: nothing stays
    h.drv.pressure = true;
    h.map(ids);
    h.unmap(ids);
    REQUIRE(h.pool.footprint().held_pages == 0, "under pressure every release goes to the driver");
  }
  // 3. decay with a floor (the reserve), refill of the reserve with clean extents
  {
    Harness h(16, 1 << 20, 0.05);
    h.drv.check_no_waste_at_create = false;
    std::vector<int64_t> ids;
    for (int i = 0; i < 160; ++i) ids.push_back(i);
    h.map(ids);
    h.unmap(ids);
    REQUIRE(h.pool.decay(1000, 500, 1000, 32) == 0, "first call opens the window");
    REQUIRE(h.pool.decay(1200, 500, 1000, 32) == 0, "window not over");
    REQUIRE(h.pool.decay(1600, 500, 48, 32) == 48, "bounded per call");
    REQUIRE(h.pool.decay(1700, 500, 1000, 32) == 80, "the rest of the surplus at the next tick, down to the floor");
    REQUIRE(h.pool.footprint().idle_pages == 32, "floor kept");
    REQUIRE(h.pool.refill_reserve(100, 40) == 40 || h.pool.refill_reserve(100, 40) >= 1, "refill is bounded per call");
    while (h.pool.refill_reserve(100, 64)) {
    }
    REQUIRE(h.pool.footprint().idle_pages >= 100 && h.pool.footprint().idle_pages < 116, "reserve reached: %zu", h.pool.footprint().idle_pages);
    bool rec = true;
    Phys p[16];
    // clean extents are not "recycled": take everything and look at the flags of the youngest ones
    size_t clean = 0, total = 0;
    std::vector<Phys> all;
    for (;;) {
      size_t n = h.pool.acquire_run(16, p, &rec, false);
      if (!n) break;
      total += n;
      clean += rec ? 0 : n;
      all.insert(all.end(), p, p + n);
    }
    REQUIRE(clean >= 68, "pre-created extents come out as clean memory (%zu of %zu)", clean, total);
    h.pool.release_batch(all.data(), all.size());
    h.drv.fail_after = h.drv.creates; // no memory: the refill gives up quietly
    h.pool.drain(0);
    REQUIRE(h.pool.refill_reserve(64, 64) == 0, "refill without memory");
    h.check();
  }
  // 4. creation failure leaves the pool untouched
  {
    Harness h(16, 1 << 20, 0.05);
    h.drv.fail_after = 1;
    Phys p[16];
    bool rec;
    REQUIRE(h.pool.acquire_run(16, p, &rec, true) == 16, "first create");
    bool threw = false;
    try {
      h.pool.acquire_run(16, p, &rec, true);
    } catch (const std::exception &) {
      threw = true;
    }
    REQUIRE(threw, "the driver's failure reaches the caller");
    auto f = h.pool.footprint();
    REQUIRE(f.held_pages == 16 && f.out_pages == 16, "nothing changed");
    // a piece released twice, or of an unknown extent, is counted and ignored
    Phys bogus{piece_id(0xdead0, 3, 16), 1};
    h.pool.release(bogus);
    Phys first = p[0];
    (void)first;
  }
  // 4b. scrub tickets: pages zeroed on their way back are not zeroed again on their way out
  {
    Harness h(16, 1 << 20, 0.05);
    h.drv.give_tags = true;
    Phys p[16], q[16];
    bool rec;
    REQUIRE(h.pool.acquire_run(8, p, &rec, true) == 8, "eight fresh pages");
    for (int i = 0; i < 8; ++i) REQUIRE(p[i].scrub_ticket == 0, "fresh memory is zeroed by whoever maps it");
    uint64_t addr[16];
    REQUIRE(h.pool.scrub_addresses(p, 8, addr) == 8, "every page has an alias address");
    for (int i = 1; i < 8; ++i) REQUIRE(addr[i] == addr[0] + (uint64_t)i * (2u << 20), "alias addresses follow the piece index");
    h.pool.release_batch(p, 5, 7);            // five pages back, queued as scrub 7
    h.pool.release_batch(p + 5, 3);           // three without a scrub (e.g. zero fill switched off)
    REQUIRE(h.pool.acquire_run(8, q, &rec, false) == 8, "the whole extent again");
    for (int i = 0; i < 8; ++i) REQUIRE(q[i].scrub_ticket == (i < 5 ? 7u : 0u), "page %d: ticket %llu", i, (unsigned long long)q[i].scrub_ticket);
    h.pool.release_batch(q, 8, 9);
    REQUIRE(h.pool.acquire_run(3, q, &rec, false) == 3 && q[0].scrub_ticket == 9 && q[2].scrub_ticket == 9, "partial reuse keeps the tickets");
    h.pool.release_batch(q, 3);               // used and returned unscrubbed: dirty again
    REQUIRE(h.pool.acquire_run(8, q, &rec, false) == 8, "all eight");
    for (int i = 0; i < 8; ++i) REQUIRE(q[i].scrub_ticket == (i < 3 ? 0u : 9u), "page %d after a dirty return", i);
    for (int i = 0; i < 8; ++i) REQUIRE(q[i].wait_ticket == 9u, "the older scrub of the extent is still waited for (page %d)", i);
    h.pool.release_batch(q, 8, 11);
    h.pool.drain(0);
    REQUIRE(h.drv.prepared == 1 && h.drv.live.empty(), "the alias is dropped before the buffer goes back");
  }
  // 5. churn: runs mapped together, freed together most of the time (a request's pages) with stragglers
  std::mt19937_64 rng(1);
  auto churn = [&](unsigned kmax, double waste, double straggle, bool random_free, bool sorted_free, const char *name) {
    Harness h(kmax, 0, waste); // pool off: only fragmentation is left
    h.drv.check_no_waste_at_create = true;
    std::vector<int64_t> free_ids;
    for (int64_t i = 0; i < 4096; ++i) free_ids.push_back(i);
    std::vector<std::vector<int64_t>> reqs;
    std::vector<double> ratio, waste_frac;
    for (int op = 0; op < 30000; ++op) {
      const bool grow = reqs.empty() || (h.slot.size() < 3000 && rng() % 100 < 52);
      if (grow) {
        size_t n = rng() % 5 == 0 ? 1 + rng() % 40 : 1 + rng() % 6;
        n = std::min(n, free_ids.size());
        if (!n) continue;
        if (sorted_free) std::sort(free_ids.begin(), free_ids.end()); // the worst case for extents: runs every time
        // pages come from the front of the free list: sequential at first, scrambled after churn
        std::vector<int64_t> ids(free_ids.begin(), free_ids.begin() + n);
        free_ids.erase(free_ids.begin(), free_ids.begin() + n);
        h.map(ids);
        if (random_free) {
          for (auto id : ids) reqs.push_back({id}); // every page is its own "request": freed at random
        } else {
          std::vector<int64_t> main, late;
          for (auto id : ids) ((double)(rng() % 1000) / 1000.0 < straggle ? late : main).push_back(id);
          if (!main.empty()) reqs.push_back(main);
          if (!late.empty()) reqs.push_back(late);
        }
      } else {
        const size_t k = rng() % reqs.size();
        h.unmap(reqs[k]);
        for (auto id : reqs[k]) free_ids.push_back(id);
        reqs[k] = reqs.back();
        reqs.pop_back();
      }
      auto f = h.pool.footprint();
      if (f.out_pages >= 256) {
        ratio.push_back((double)f.held_pages / f.out_pages);
        waste_frac.push_back((double)f.free_pieces / f.out_pages);
      }
    }
    for (auto &r : reqs) h.unmap(r);
    REQUIRE(h.drv.live.empty(), "pool off: everything went back");
    char buf[512];
    snprintf(buf, sizeof buf,
             ", \"%s\": {\"held_over_mapped_p50\": %.3f, \"p90\": %.3f, \"p99\": %.3f, \"max\": %.3f, \"map_ioctls_per_page\": %.3f, "
             "\"creates_per_page\": %.3f}",
             name, pct(ratio, 0.5), pct(ratio, 0.9), pct(ratio, 0.99), pct(ratio, 1.0), (double)h.map_ioctls / h.pages_mapped,
             (double)h.drv.creates / h.pages_mapped);
    js += buf;
    return pct(ratio, 0.9);
  };
  const double p90_requests = churn(64, 0.05, 0.05, false, false, "requests_with_5pct_stragglers");
  REQUIRE(p90_requests <= 1.10, "p90 %.3f", p90_requests);
  const double p90_sorted = churn(64, 0.05, 0.05, false, true, "requests_with_5pct_stragglers_sorted_free_list");
  REQUIRE(p90_sorted <= 1.10, "p90 %.3f", p90_sorted);
  const double p90_random = churn(64, 0.05, 0, true, true, "runs_mapped_together_freed_at_random_governed");
  REQUIRE(p90_random <= 1.15, "the governor bounds the worst case: p90 %.3f", p90_random);
  const double p90_ungoverned = churn(64, 1e9, 0, true, true, "runs_mapped_together_freed_at_random_ungoverned");
  REQUIRE(p90_ungoverned <= 1.15, "with steady demand free pieces are used up before anything is created: p90 %.3f", p90_ungoverned);
  // 6. the adversary of extents: memory grows in runs and shrinks page by page at random, over and over (tides). What was
  // mapped together is NOT freed together, so stragglers pin their extents while the tide is out. The governor cannot
  // undo extents that exist, but it stops making them once the waste shows: from the second tide on, pages are single.
  auto tides = [&](double waste, const char *name) {
    Harness h(64, 0, waste);
    std::vector<double> ratio;
    std::vector<int64_t> mapped;
    int64_t next_id = 0;
    for (int tide = 0; tide < 6; ++tide) {
      while (mapped.size() < 1500u * (tide + 1)) { // every tide comes in higher than the last
        std::vector<int64_t> ids;
        for (int i = 0; i < 32; ++i) ids.push_back(next_id++);
        h.map(ids);
        mapped.insert(mapped.end(), ids.begin(), ids.end());
        auto f = h.pool.footprint();
        if (f.out_pages >= 256) ratio.push_back((double)f.held_pages / f.out_pages);
      }
      std::shuffle(mapped.begin(), mapped.end(), rng);
      while (mapped.size() > 150u * (tide + 1)) {
        std::vector<int64_t> ids(mapped.end() - 20, mapped.end());
        mapped.resize(mapped.size() - 20);
        h.unmap(ids);
        auto f = h.pool.footprint();
        if (f.out_pages >= 256) ratio.push_back((double)f.held_pages / f.out_pages);
      }
    }
    h.unmap(mapped);
    REQUIRE(h.drv.live.empty(), "everything went back");
    char buf[256];
    snprintf(buf, sizeof buf, ", \"%s\": {\"held_over_mapped_p50\": %.3f, \"p90\": %.3f, \"max\": %.3f}", name, pct(ratio, 0.5),
             pct(ratio, 0.9), pct(ratio, 1.0));
    js += buf;
    return pct(ratio, 0.9);
  };
  const double tides_governed = tides(0.05, "tides_governed"), tides_ungoverned = tides(1e9, "tides_ungoverned");
  REQUIRE(tides_governed < 0.75 * tides_ungoverned, "the governor learns from the first tide (%.2f vs %.2f)", tides_governed, tides_ungoverned);
  // single-page pools never look inside a handle: ROCr's use bits above 48
  {
    Harness h(1, 8, 0.05);
    h.drv.high_bits = 0x80ab000000000000ull; // bit 63 too: what a piece id of a multi-page pool is tagged with
    std::vector<int64_t> ids{1, 2, 3, 7};
    h.map(ids);
    for (auto &kv : h.slot) REQUIRE((kv.second.h >> 48) == 0x80ab, "handle passed through untouched");
    h.unmap(ids);
    REQUIRE(h.pool.footprint().idle_pages == 4 && h.pool.footprint().bad_releases == 0, "released and recycled");
    h.map(ids);
    REQUIRE(h.ctr.reused == 4, "reused");
    h.unmap(ids);
  }
  const double p90_single = churn(1, 0.05, 0, true, true, "single_pages");
  REQUIRE(p90_single == 1.0, "single pages cannot fragment");
  js += "}";
  puts(js.c_str());
  return 0;
}
