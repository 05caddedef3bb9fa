#!/usr/bin/env python3
"""bench_vmm — BASELINE.json configs[1] in full (SURVEY §8d "bench_vmm (cfg 2)").

One MI355X, 64 GiB of reserved VA = 32 768 slots of 2 MiB, swept as 32 batches x 1024 pages:
  batch mode    every batch is mapped (+zeroed) and unmapped again through the C ABI
                (kvc_map_to_kv_tensors / kvc_unmap_from_kv_tensors); offsets inside a batch are
                (i) sequential, (ii) a seed-0 permutation. 3 warm-up sweeps + `--sweeps` timed sweeps.
                Reported per batch: p50/p90/p99 of map+zero and of unmap, aggregate GB/s backed
                (1024 x 2 MiB / t_batch), and the per-phase split the library measures around every
                driver call: pool-pop|create, map, set_access, (TLB shootdown), zero_fill, unmap, release.
  per-call mode batch = 1 page, the shape of the reference's own table
                (benchmarks/bench_vmm/README.md:49-58: create 193 / map 1.5 / set_access 36 / unmap 26 us on
                A100): latency of one kvc_map_to_kv_tensors([off]) and one kvc_unmap_from_kv_tensors([off]),
                avg/p50/p90/p99/max. Here one call = pool-pop + hipMemMap + hipMemSetAccess + TLB
                shootdown + a 2 MiB fill launch + stream sync — the shootdown (~0.2 ms) is the floor.

Prints one JSON line per measurement. The raw driver-call latencies (no library around them) are in
profiles/r01_vmm_probe.log (kvcached_amd/csrc/tools/vmm_probe.cpp).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)

MiB, GiB = 1 << 20, 1 << 30
PAGE = 2 * MiB
BATCH = 1024
WINDOW_BATCHES = 32


def pct(xs, q):
    xs = sorted(xs)
    return xs[min(len(xs) - 1, int(round(q * (len(xs) - 1))))]


def lat(xs, scale=1e6):
    return {"avg": round(sum(xs) / len(xs) * scale, 2), "p50": round(pct(xs, 0.5) * scale, 2),
            "p90": round(pct(xs, 0.9) * scale, 2), "p99": round(pct(xs, 0.99) * scale, 2),
            "max": round(max(xs) * scale, 2)}


def offsets(batch, order):
    import numpy as np
    idx = np.arange(BATCH) if order == "sequential" else np.random.default_rng(0).permutation(BATCH)
    return [int(batch * BATCH + i) * PAGE for i in idx]


def batch_mode(capi, device, order, sweeps, warm):
    capi.init(device, PAGE, False)
    t0 = time.perf_counter()
    capi.create_kv_tensors(WINDOW_BATCHES * BATCH * PAGE, 1, device, 1, 1, 0, True)
    reserve_s = time.perf_counter() - t0
    try:
        for _ in range(warm):
            for b in range(WINDOW_BATCHES):
                o = offsets(b, order)
                capi.map_to_kv_tensors(o)
                capi.unmap_from_kv_tensors(o)
        capi.set_option(capi.OPT_PROFILE, 1)
        capi.reset_stats()
        t_map, t_unmap = [], []
        t_all = time.perf_counter()
        for _ in range(sweeps):
            for b in range(WINDOW_BATCHES):
                o = offsets(b, order)
                ta = time.perf_counter()
                capi.map_to_kv_tensors(o)
                tb = time.perf_counter()
                capi.unmap_from_kv_tensors(o)
                tc = time.perf_counter()
                t_map.append(tb - ta)
                t_unmap.append(tc - tb)
        t_all = time.perf_counter() - t_all
        st = capi.get_stats()
        drv = capi.get_driver_breakdown()
        capi.set_option(capi.OPT_PROFILE, 0)
        n = len(t_map)
        pages = n * BATCH
        phases = {k: round(v / 1e3 / pages, 3) for k, v in drv.items() if v}
        phases["tlb_shootdown"] = round(st["shootdown_ns"] / 1e3 / pages, 3)
        phases["zero_fill_kernel"] = round(st["fill_ms"] * 1e3 / pages, 3)
        return {"mode": "batch", "order": order, "batches": n, "pages_per_batch": BATCH, "window_GiB": 64,
                "cycle_GBps": round(pages * PAGE / t_all / 1e9, 2),
                "map_zero_GBps": round(pages * PAGE / sum(t_map) / 1e9, 2),
                "unmap_GBps": round(pages * PAGE / sum(t_unmap) / 1e9, 2),
                "map_zero_batch_ms": lat(t_map, 1e3), "unmap_batch_ms": lat(t_unmap, 1e3),
                "phase_us_per_page": phases,
                "zero_fill_GBps": round(st["fill_bytes"] / (st["fill_ms"] * 1e-3) / 1e9, 1) if st["fill_ms"] else None,
                "handles_created": st["handles_created"], "handles_reused": st["handles_reused"],
                "va_reserve_s": round(reserve_s, 4)}
    finally:
        capi.shutdown()


def per_call_mode(capi, device, calls):
    import numpy as np
    capi.init(device, PAGE, False)
    capi.create_kv_tensors(WINDOW_BATCHES * BATCH * PAGE, 1, device, 1, 1, 0, True)
    try:
        slots = [int(i) * PAGE for i in np.random.default_rng(0).permutation(WINDOW_BATCHES * BATCH)[:calls]]
        for o in slots[:64]:
            capi.map_to_kv_tensors([o])
            capi.unmap_from_kv_tensors([o])
        out = {}
        for name, fill, shoot in (("default", 1, 1), ("no_zero_fill", 0, 1), ("no_fill_no_shootdown(unsafe, floor)", 0, 0)):
            capi.set_option(capi.OPT_ZERO_FILL, fill)
            capi.set_option(capi.OPT_TLB_SHOOTDOWN, shoot)
            capi.reset_stats()
            t_map, t_unmap = [], []
            for o in slots:
                ta = time.perf_counter()
                capi.map_to_kv_tensors([o])
                tb = time.perf_counter()
                capi.unmap_from_kv_tensors([o])
                tc = time.perf_counter()
                t_map.append(tb - ta)
                t_unmap.append(tc - tb)
            drv = capi.get_driver_breakdown()
            out[name] = {"map_call_us": lat(t_map), "unmap_call_us": lat(t_unmap),
                         "driver_us_per_call": {k: round(v / 1e3 / calls, 2) for k, v in drv.items() if v}}
        capi.set_option(capi.OPT_ZERO_FILL, 1)
        capi.set_option(capi.OPT_TLB_SHOOTDOWN, 1)
        return {"mode": "per_call", "calls": calls, "page_MiB": 2, **out}
    finally:
        capi.shutdown()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--sweeps", type=int, default=10)
    ap.add_argument("--warmup-sweeps", type=int, default=3)
    ap.add_argument("--calls", type=int, default=1024)
    args = ap.parse_args()
    import torch
    if not torch.cuda.is_available():
        raise SystemExit("bench_vmm needs a GPU: the HIP path has no CPU fallback")
    os.environ.setdefault("KVCACHED_IPC_NAME", f"kvc_bench_vmm_{os.getpid()}")
    os.environ.setdefault("KVCACHED_LOG_LEVEL", "ERROR")
    from kvcached_amd import capi
    for order in ("sequential", "shuffled"):
        print(json.dumps(batch_mode(capi, "cuda:0", order, args.sweeps, args.warmup_sweeps)), flush=True)
    print(json.dumps(per_call_mode(capi, "cuda:0", args.calls)), flush=True)


if __name__ == "__main__":
    main()
