#!/usr/bin/env python3
"""bench_engine_geometry — the map/unmap hot path in the geometry engines actually use on ROCm.

The reference forces the NON-contiguous layout on ROCm (kvcached/utils.py:150-171): one region per layer, K half and V
half, so ONE page id is 64 single 2 MiB slots in 64 places of the address space for Llama-3-8B (32 layers x K/V;
csrc/allocator.cpp:189-206) and n consecutive page ids are 64 runs of n adjacent slots. This script drives exactly that
through the C ABI (no socket, no Python manager): n page ids per call, consecutive or scattered, pool warm, and prints
per-call p50 latencies, GB/s backed and where the host time of a call goes (segments 130..161 of the library).

    python benchmarks/bench_engine_geometry.py [--mode compat|lazy] [--iters 40] [--ids 1,8,64] [--layers 32]
One JSON line per (n, placement).
"""
from __future__ import annotations

import argparse
import json
import os
import statistics
import sys
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
PAGE = 2 << 20

SEG = {0: "map.offsets_to_slots", 1: "map.classify", 2: "map.runs", 3: "map.pool", 4: "map.ioctls", 5: "map.per_run_bookkeeping",
       6: "map.invalidation_owed(total)", 7: "map.wait_own_fill", 8: "map.wait_scrub", 10: "unmap.offsets_to_slots", 11: "unmap.runs",
       12: "unmap.ioctls", 13: "unmap.per_slot", 14: "unmap.remainders", 15: "unmap.epochs", 16: "unmap.invalidation",
       17: "unmap.scrub_launch", 18: "unmap.pool", 19: "both.kfd_remap_half", 20: "map.remainder_rewrite(share of invalidation_owed)"}
COUNTS = {21: "map.ioctls_issued", 22: "unmap.ioctls_issued", 23: "map.runs_found"}


def run(capi, n_ids, placement, iters, half_slots, layers, rng):
    """placement:
    'steady'     page ids b .. b+n-1 with b advancing by n and wrapping: after one sweep over the window every call finds the
                 address space as its previous visit left it (what an engine that has been up for a while sees);
    'straddling' the same runs shifted against the shape the previous unmap left: every call splits rest mappings;
    'scattered'  n page ids drawn at random (a churned free list), most of them touched for the first time."""
    def ids_for(it):
        if placement == "steady":
            b = (it * n_ids) % (half_slots - half_slots % n_ids)
            return list(range(b, b + n_ids))
        if placement == "straddling":
            b = (it * n_ids * 3 + 5) % (half_slots - n_ids)
            return list(range(b, b + n_ids))
        return sorted(int(x) for x in rng.choice(half_slots, size=n_ids, replace=False))

    # warm: one sweep over the window in the timed pattern (steady), a few cycles otherwise; the pool then holds buffers of
    # the shapes this pattern asks for
    warm = half_slots // n_ids if placement == "steady" else max(6, 2048 // (n_ids * layers * 2))
    for it in range(warm):
        offs = [p * PAGE for p in ids_for(it if placement == "steady" else 10_000 + it)]
        capi.map_to_kv_tensors(offs)
        capi.unmap_from_kv_tensors(offs)
    capi.flush_unmaps()
    capi.reset_stats()
    c0 = int(capi.get_option(115))
    tm, tu = [], []
    for it in range(iters):
        arr = capi.i64_array([p * PAGE for p in ids_for(it)])
        t0 = time.perf_counter()
        capi.map_to_kv_tensors(arr)
        t1 = time.perf_counter()
        capi.unmap_from_kv_tensors(arr)
        t2 = time.perf_counter()
        tm.append(t1 - t0)
        tu.append(t2 - t1)
    capi.flush_unmaps()
    st = capi.get_stats()
    slots = n_ids * layers * 2
    seg = {name: round(int(capi.get_option(130 + i)) / 1e3 / iters, 1) for i, name in SEG.items()}
    cnt = {name: round(int(capi.get_option(130 + i)) / iters, 1) for i, name in COUNTS.items()}
    p50m, p50u = statistics.median(tm), statistics.median(tu)
    return {"page_ids": n_ids, "placement": placement, "slots_2MiB": slots,
            "map_ms": {"p50": round(p50m * 1e3, 3), "p90": round(sorted(tm)[int(.9 * (len(tm) - 1))] * 1e3, 3), "max": round(max(tm) * 1e3, 3)},
            "unmap_ms": {"p50": round(p50u * 1e3, 3), "p90": round(sorted(tu)[int(.9 * (len(tu) - 1))] * 1e3, 3)},
            "map_GBps_backed_p50": round(slots * PAGE / p50m / 1e9, 1),
            "cycle_GBps": round(slots * PAGE * iters / (sum(tm) + sum(tu)) / 1e9, 1),
            "us_per_2MiB_map": round(p50m * 1e6 / slots, 2),
            "host_us_per_call": {k: v for k, v in seg.items() if v},
            "per_call": cnt, "tlb_shootdowns_per_cycle": round(st["tlb_shootdowns"] / iters, 2),
            "shootdown_us": round(st["shootdown_ns"] / 1e3 / max(1, st["tlb_shootdowns"]), 1),
            "driver_allocations_in_timed_region": int(capi.get_option(115)) - c0,
            "handles_created": st["handles_created"], "handles_reused": st["handles_reused"]}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--mode", choices=["compat", "lazy"], default="compat")
    ap.add_argument("--iters", type=int, default=40)
    ap.add_argument("--ids", default="1,8,64")
    ap.add_argument("--layers", type=int, default=32)
    ap.add_argument("--half-slots", type=int, default=512, help="page ids per region half (K or V of one layer)")
    ap.add_argument("--placements", default="steady,straddling,scattered")
    args = ap.parse_args()
    os.environ.setdefault("KVCACHED_IPC_NAME", f"kvc_engine_{os.getpid()}")
    os.environ.setdefault("KVCACHED_LOG_LEVEL", "ERROR")
    os.environ["KVCACHED_ZERO_BACKFILL"] = "true" if args.mode == "compat" else "false"
    import numpy as np
    import torch
    assert torch.cuda.is_available(), "needs the MI355X"
    torch.cuda.set_device(0)
    from kvcached_amd import capi
    t0 = time.perf_counter()
    capi.init("cuda:0", PAGE, False)
    capi.create_kv_tensors(2 * args.half_slots * PAGE, 1, "cuda:0", args.layers, 2, 0, False)
    print(json.dumps({"startup_s": round(time.perf_counter() - t0, 3), "mode": args.mode, "layers": args.layers,
                      "VA_GiB": 2 * args.half_slots * PAGE * args.layers / 2**30, "prt": int(capi.get_option(128)),
                      "max_extent_pages": int(capi.get_option(119))}), flush=True)
    rng = np.random.default_rng(0)
    try:
        for placement in args.placements.split(","):
            for n in [int(x) for x in args.ids.split(",")]:
                if n > args.half_slots // 2:
                    continue
                print(json.dumps(run(capi, n, placement, args.iters, args.half_slots, args.layers, rng)), flush=True)
    finally:
        capi.shutdown()


if __name__ == "__main__":
    main()
