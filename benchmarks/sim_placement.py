"""CPU simulation of physical-extent placement policies: what does a policy cost in memory held
from the driver per mapped page, and how many page-table ioctls does it need?

The workload is real: a live KVCacheManager on the library's `cpu` device (bookkeeping only) is driven by the op mix of
benchmarks/soak_manager.py (or by a Poisson request trace at the Llama-3-8B geometry), and the map / unmap calls it
issues - the page ids, in the order and grouping of the product's batched calls - are replayed into a Python model of
the extent pool. One region is modelled (slot = page id): every layer's K and V region sees the same sequence.

    python benchmarks/sim_placement.py [--workload soak|poisson] [--ops 20000]
"""
import argparse
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
PAGE = 2 << 20


def record_soak(ops: int, seed: int, page_ids: int = 1024, prealloc: bool = False):
    """The op mix of soak_manager.py; returns [(kind, [page ids])...] with kind 0 = map call, 1 = unmap call,
    2 = sample point."""
    os.environ["KVCACHED_PAGE_PREALLOC_ENABLED"] = "true" if prealloc else "false"
    os.environ.setdefault("KVCACHED_IPC_NAME", f"kvc_sim_{os.getpid()}")
    os.environ.setdefault("KVCACHED_LOG_LEVEL", "ERROR")
    import kvc_testlib as T
    layers, block_tokens, cell = 4, 16, 2048
    num_blocks = page_ids * (PAGE // (block_tokens * cell))
    ad = T.ProductAdapter(num_blocks, block_tokens, cell, layers, device="cpu", batch_page_alloc=True)
    rng = np.random.default_rng(seed)
    live, next_rid, ev = {}, 1, []

    def drain():
        for kind, offs in ad.drain_events():
            ev.append((kind, sorted(o // PAGE for o in offs)))

    def free_one(rid):
        ad.free(live.pop(rid))
        drain()

    for i in range(ops):
        if i % 16 == 0:
            ev.append((2, []))
        r = rng.random()
        held = sum(len(v) for v in live.values())
        if r < 0.50 or not live:
            n = int(rng.integers(1, 4000)) if rng.random() < 0.8 else int(rng.integers(1, 40))
            ids = ad.alloc(n)
            drain()
            if ids is None:
                if live:
                    free_one(next(iter(live)))
                continue
            live[next_rid] = ids
            next_rid += 1
        elif r < 0.93:
            keys = list(live)
            free_one(keys[int(rng.integers(len(keys)))])
        elif r < 0.96:
            ad.trim()
            drain()
        if held > num_blocks * 0.6:
            for rid in list(live)[:3]:
                free_one(rid)
    for rid in list(live):
        free_one(rid)
    ad.trim()
    drain()
    ad.close()
    return ev


def record_poisson(seconds: float, lam: float, seed: int, page_ids: int = 4096):
    """Llama-3-8B geometry request trace (SURVEY §8d cfg 3): Poisson arrivals, log-normal prompts, geometric outputs; a
    request allocates ceil(len/16) blocks at arrival, one block every 16 decode steps (one step = 25 ms), frees all at
    completion."""
    os.environ["KVCACHED_PAGE_PREALLOC_ENABLED"] = "false"
    os.environ.setdefault("KVCACHED_IPC_NAME", f"kvc_sim_{os.getpid()}")
    os.environ.setdefault("KVCACHED_LOG_LEVEL", "ERROR")
    import heapq

    import kvc_testlib as T
    block_tokens, cell, layers = 16, 2048, 2
    num_blocks = page_ids * 64
    ad = T.ProductAdapter(num_blocks, block_tokens, cell, layers, device="cpu", batch_page_alloc=True)
    rng = np.random.default_rng(seed)
    ev, t, step = [], 0.0, 0.025
    active = {}   # rid -> [tokens so far, tokens at end, block ids]
    next_arrival, rid = rng.exponential(1 / lam), 0

    def drain():
        for kind, offs in ad.drain_events():
            ev.append((kind, sorted(o // PAGE for o in offs)))

    while t < seconds:
        t += step
        while next_arrival <= t:
            prompt = int(np.clip(rng.lognormal(6.5, 1.0), 16, 8192))
            out = int(np.clip(rng.geometric(1 / 256), 1, 2048))
            ids = ad.alloc(-(-prompt // block_tokens))
            drain()
            if ids is not None:
                active[rid] = [prompt, prompt + out, ids]
                rid += 1
            next_arrival += rng.exponential(1 / lam)
        for r in list(active):
            a = active[r]
            a[0] += 1
            if a[0] % block_tokens == 1:
                more = ad.alloc(1)
                drain()
                if more:
                    a[2].extend(more)
            if a[0] >= a[1]:
                ad.free(a[2])
                drain()
                del active[r]
        ev.append((2, []))
    for r in list(active):
        ad.free(active[r][2])
    drain()
    ad.trim()
    drain()
    ad.close()
    return ev


class ExtentPool:
    """Model of the physical pool. An extent = one buffer object of `n` pages; piece i of it can back any slot; adjacent
    slots on adjacent pieces of one extent are one map ioctl. A whole-free extent goes back to the driver (pool off:
    fragmentation is what is being measured)."""

    def __init__(self, policy: str, kmax: int, alpha: float = 0.05, fixed: bool = False):
        self.policy, self.kmax_cfg, self.alpha, self.fixed = policy, kmax, alpha, fixed
        self.kmax = kmax
        self.ext = {}          # id -> [n, free bitmask]
        self.next_id = 1
        self.slot = {}         # page id -> (extent id, piece)
        self.held = 0          # pages held from the driver
        self.mapped = 0
        self.free_pieces = 0   # W
        self.map_ioctls = self.unmap_ioctls = self.creates = self.pages_mapped = 0

    # ---- helpers
    @staticmethod
    def _runs(mask, n):
        """[(start, len)] of the free runs of an n-piece mask."""
        out, i = [], 0
        while i < n:
            if mask >> i & 1:
                j = i
                while j < n and mask >> j & 1:
                    j += 1
                out.append((i, j - i))
                i = j
            else:
                i += 1
        return out

    def _take_free_run(self, want):
        """Best free run among partly used extents: exact fit, else the shortest run >= want, else the longest run."""
        best = None
        for eid, (n, mask) in self.ext.items():
            if not mask:
                continue
            for start, ln in self._runs(mask, n):
                key = (0, ln) if ln >= want else (1, -ln)
                if best is None or key < best[0]:
                    best = (key, eid, start, ln)
                    if ln == want:
                        break
        if best is None:
            return None
        _, eid, start, ln = best
        k = min(ln, want)
        e = self.ext[eid]
        e[1] &= ~(((1 << k) - 1) << start)
        self.free_pieces -= k
        return eid, start, k

    def _create(self, n):
        eid = self.next_id
        self.next_id += 1
        self.ext[eid] = [n, (1 << n) - 1]
        self.held += n
        self.free_pieces += n
        self.creates += 1
        return eid

    def map(self, ids):
        i = 0
        while i < len(ids):
            j = i + 1
            while j < len(ids) and ids[j] == ids[j - 1] + 1:
                j += 1
            while i < j:
                want = j - i
                got = self._take_free_run(want) if self.free_pieces else None
                if got is None:
                    if self.fixed:
                        n = self.kmax_cfg
                    else:
                        n = min(want, self.kmax)
                    self._create(n)
                    got = self._take_free_run(want)
                eid, start, k = got
                for t in range(k):
                    self.slot[ids[i + t]] = (eid, start + t)
                self.map_ioctls += 1
                self.mapped += k
                self.pages_mapped += k
                i += k
        self._govern()

    def unmap(self, ids):
        # one CLEAR per run of adjacent slots (any extents), as the product does (runs capped at 16 there for single pages)
        i = 0
        while i < len(ids):
            j = i + 1
            while j < len(ids) and ids[j] == ids[j - 1] + 1:
                j += 1
            self.unmap_ioctls += 1
            i = j
        touched = set()
        for p in ids:
            eid, piece = self.slot.pop(p)
            e = self.ext[eid]
            e[1] |= 1 << piece
            self.free_pieces += 1
            self.mapped -= 1
            touched.add(eid)
        for eid in touched:
            n, mask = self.ext[eid]
            if mask == (1 << n) - 1:
                del self.ext[eid]
                self.held -= n
                self.free_pieces -= n
            elif n > 1:
                self.unmap_ioctls += 2   # refresh_mappings_of: the remainder of a split mapping is rewritten
        self._govern()

    def _govern(self):
        if self.policy != "governed" or self.fixed:
            return
        w = self.free_pieces
        if w > self.alpha * max(self.mapped, 64):
            self.kmax = max(1, self.kmax // 2)
        elif w < 0.25 * self.alpha * max(self.mapped, 64):
            self.kmax = min(self.kmax_cfg, self.kmax * 2)


def replay(ev, pool):
    fp = []
    for kind, ids in ev:
        if kind == 0:
            pool.map(ids)
        elif kind == 1:
            pool.unmap(ids)
        elif pool.mapped >= 256:
            fp.append(pool.held / pool.mapped)
    fp.sort()
    q = lambda f: round(fp[min(len(fp) - 1, int(len(fp) * f))], 3) if fp else None  # noqa: E731
    return {"p50": q(0.5), "p90": q(0.9), "p99": q(0.99), "max": q(1.0), "samples": len(fp),
            "map_ioctls_per_page": round(pool.map_ioctls / max(1, pool.pages_mapped), 3),
            "unmap_ioctls_per_page": round(pool.unmap_ioctls / max(1, pool.pages_mapped), 3),
            "creates_per_page": round(pool.creates / max(1, pool.pages_mapped), 3), "pages_mapped": pool.pages_mapped,
            "leak": pool.held}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="soak")
    ap.add_argument("--ops", type=int, default=20000)
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--lam", type=float, default=16)
    ap.add_argument("--seconds", type=float, default=120)
    ap.add_argument("--prealloc", action="store_true")
    args = ap.parse_args()
    ev = record_soak(args.ops, args.seed, prealloc=args.prealloc) if args.workload == "soak" else record_poisson(args.seconds, args.lam, args.seed)
    n_map = sum(len(i) for k, i in ev if k == 0)
    runs = []
    for k, ids in ev:
        if k == 0:
            i = 0
            while i < len(ids):
                j = i + 1
                while j < len(ids) and ids[j] == ids[j - 1] + 1:
                    j += 1
                runs.append(j - i)
                i = j
    print(json.dumps({"workload": args.workload, "map_calls": sum(1 for k, _ in ev if k == 0), "pages_mapped": n_map,
                      "mean_run": round(float(np.mean(runs)), 2), "p50_run": int(np.median(runs)), "max_run": int(max(runs))}))
    for name, pool in (("one buffer per page", ExtentPool("plain", 1)),
                       ("fixed 16-page chunks (round 1)", ExtentPool("plain", 16, fixed=True)),
                       ("fixed 4-page chunks", ExtentPool("plain", 4, fixed=True)),
                       ("run-sized extents <= 16", ExtentPool("plain", 16)),
                       ("run-sized extents <= 64", ExtentPool("plain", 64)),
                       ("run-sized <= 64, governed 5 %", ExtentPool("governed", 64, 0.05)),
                       ("run-sized <= 64, governed 3 %", ExtentPool("governed", 64, 0.03))):
        print(json.dumps({"policy": name, **replay(ev, pool)}))


if __name__ == "__main__":
    main()
