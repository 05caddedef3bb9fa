#!/usr/bin/env python3
"""bench.py — GB/s of KV physically backed (map + zero) at 2 MiB granularity on MI355X.

Workload (BASELINE.json configs[1], "bench_vmm: reserve 64 GiB VA, map/unmap 2 MiB pages in 1024-page
batches"): one STEP = one elastic cycle on one batch of 1024 x 2 MiB pages, sweeping the 64 GiB window:
  kvc_map_to_kv_tensors(1024 offsets)   physical pages from the library's pool (run-sized extents) + one page-table ioctl per run
                                        of adjacent slots + the TLB invalidation the batch owes; the pages were zeroed when
                                        they came back (zero_fill_pages through their buffers' alias mappings, queued behind the
                                        previous unmap), so the call only waits for that scrub if it has not finished
  kvc_unmap_from_kv_tensors(same)       one ioctl per run back to the rest state + TLB invalidation + the zero fill of the
                                        pages queued on the library's scrub stream + pages back to the pool
  (the cycle ends with kvc_flush_unmaps inside the timed bracket: every scrub and invalidation in flight is waited for)
Set-up (untimed, like the VA reservation): every batch of the window is mapped and unmapped once — the warm-up sweep of
the bench_vmm protocol. It matters: the first ~30 batches a process pushes through the driver are ~20 % slower
(hipMemMap 3.5 instead of 2.3 us/page) whatever VA they touch; the variant `fresh_va_window_warm_process` shows that
a never-mapped VA window in an already warm process runs at full speed, so this is process warm-up, not page-table
first touch.
Both halves are inside the timed bracket; `value` = bytes backed / total wall time. `map_zero_GBps` and
`p50_map_batch_ms` isolate the map+zero half. Offsets inside a batch are a seeded permutation (SURVEY
§8d). The other natural reading — a growth burst of K batches with nothing unmapped — is reported as the
variant `growth_burst` (there hipMemCreate's O(live handles) cost dominates).

  python bench.py [--gpus N --steps K --warmup W]
N > 1: this script starts its own N ranks (spawn_ranks: one process per GPU, RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set as
torch.distributed.run would; under such a launcher - WORLD_SIZE already set - it is a rank). The path shards with no
data-path collective — every rank backs its own KV shard — and the only exchange is the one the
reference has too: rank 0's offset vector is broadcast to the TP group (CollectiveFanout: one RCCL
broadcast + one status all-reduce per step). Weak scaling: per-GPU work is fixed.

At N = 1 the line also carries the geometry engines really use on ROCm (engine_llama3_8b_noncontig_*: one region per layer,
K and V halves - one page id = 64 slots in 64 places; benchmarks/bench_engine_geometry.py) and, at N > 1, the shared-pool
leg (rank 0 backs and exports page ids, the other ranks import and map them and read rank 0's signature through the mapping).

Prints ONE JSON line: metric/value/unit/... + "roofline" (zero_fill_pages, HBM-bound; kernel time
from HIP events recorded inside the library on the stream the kernel runs on) + "cpu_baseline"
(the CPU oracle timed on the host cores, rank 0 at N = 1) + extras (p50 batch latency, per-mode
variants, and — when oracle/_ref exists — the REAL reference's HIP path timed on the same box).
"""
from __future__ import annotations

import argparse
import json
import os
import statistics
import subprocess
import sys
import time

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

MiB, GiB = 1 << 20, 1 << 30
PAGE = 2 * MiB
BATCH_PAGES = 1024
HBM_PEAK_GBPS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md)


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=24)
    ap.add_argument("--warmup", type=int, default=4)
    ap.add_argument("--mode", choices=["lazy", "compat"], default="compat",
                    help="compat (library default, the reference's semantics): unbacked VA reads as zeros (a PRT mapping), a map "
                         "batch invalidates the TLBs before its pages are used; lazy (KVCACHED_ZERO_BACKFILL=false): unbacked VA "
                         "stays unmapped, a stray access faults, and neither call waits for an invalidation")
    ap.add_argument("--pool-mb", type=int, default=None, help="idle physical-handle pool cap (default: library default)")
    ap.add_argument("--no-variants", action="store_true", help="skip the extra per-mode runs at N=1")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--backend", choices=["hip", "hybrid", "drm"], default=os.environ.get("KVCACHED_VMM_BACKEND", "drm"),
                    help="VMM backend requested for the main measurement (library default: drm; DESIGN.md §4.6/§4.7); the line "
                         "reports the one in effect after the library's self tests")
    ap.add_argument("--shared-pool-leg", action="store_true",
                    help="internal: this process is one rank of the shared-pool leg (a child of a rank of the N > 1 run, so that a "
                         "failure in the cross-GPU import cannot take the bench line with it)")
    ap.add_argument("--growth-burst-only", action="store_true",
                    help="internal: run the growth burst (24 x 2 GiB backed, nothing unmapped) as the first GPU work of a "
                         "fresh process and print its summary (the N=1 line's growth_burst_first_touch leg)")
    return ap.parse_args()


# host time of the two calls by segment (read-only options 130 + i of the library)
HOST_SEGMENTS = {0: "map: offsets to slots", 1: "map: classify slots", 2: "map: sort into runs", 3: "map: pool", 4: "map: page-table ioctls",
                 5: "map: per-run bookkeeping", 6: "map: invalidation owed", 7: "map: wait for own fill", 8: "map: wait for the scrub of these pages", 9: "map: scrubs behind the newest (count, x1000)", 10: "unmap: offsets to slots", 11: "unmap: sort into runs",
                 12: "unmap: page-table ioctls", 13: "unmap: per-slot bookkeeping", 14: "unmap: remainders of split mappings",
                 15: "unmap: epochs", 16: "unmap: TLB invalidation", 17: "unmap: scrub launch", 18: "unmap: pool",
                 19: "both: re-MAP half of the KFD pairs (the UNMAP half is the flush itself)"}


def batch_offsets(batch_index: int, seed: int = 0, slot: int = PAGE):
    """Offsets of one 2 GiB batch: 1024 shuffled 2 MiB slots (or 2 GiB / slot compound slots)."""
    import numpy as np
    n = BATCH_PAGES * PAGE // slot
    perm = np.random.default_rng(seed + batch_index).permutation(n)
    base = batch_index * n
    return [int(base + p) * slot for p in perm]


class Pool:
    """One region of `window` batches, driven through the C ABI."""

    def __init__(self, capi, device: str, window_batches: int, mode: str, pool_mb, page=PAGE, group_id=0,
                 compound_layers: int = 0, backend: str = "drm"):
        self.capi, self.device, self.window = capi, device, window_batches
        os.environ["KVCACHED_VMM_BACKEND"] = backend
        os.environ["KVCACHED_ZERO_BACKFILL"] = "true" if mode == "compat" else "false"
        if pool_mb is not None:
            os.environ["KVCACHED_PHYS_POOL_MB"] = str(pool_mb)
        else:
            os.environ.pop("KVCACHED_PHYS_POOL_MB", None)
        self.size = window_batches * BATCH_PAGES * PAGE       # 64 GiB of VA whatever the slot size
        t0 = time.perf_counter()
        if compound_layers:
            # contiguous layout: one region, slot = compound page (2 MiB x layers x K/V), e.g. 128 MiB for Llama-3-8B
            capi.init(device, page, True)
            self.slot = page * compound_layers * 2
            self.tensors = capi.create_kv_tensors(self.size // compound_layers, 1, device, compound_layers, 2, group_id, False)
        else:
            # one layer, one buffer, unified pool: one 2 MiB slot per offset (the bench_vmm shape)
            capi.init(device, page, False)
            self.slot = page
            self.tensors = capi.create_kv_tensors(self.size, 1, device, 1, 1, group_id, True)
        self.reserve_s = time.perf_counter() - t0

    def close(self):
        self.capi.shutdown()
        os.environ.pop("KVCACHED_VMM_BACKEND", None)


def run_steps(capi, mapper, first_batch: int, n: int, slot: int = PAGE):
    """n timed map+zero steps; returns per-step seconds."""
    per = []
    for i in range(n):
        offs = batch_offsets(first_batch + i, slot=slot)
        t0 = time.perf_counter()
        mapper(offs)
        per.append(time.perf_counter() - t0)
    return per


def measure(capi, device, steps, warmup, mode, pool_mb, fanout=None, barrier=None, sync=None, compound_layers=0,
            burst=False, prefault=True, backend="drm", page=PAGE):
    """cycle (default): every step maps+zeroes one batch and unmaps it again. burst=True: `steps` batches are
    backed one after the other and only unmapped after the timed region.
    prefault: as in the bench_vmm protocol (warm-up sweeps over the whole window before the timed sweeps), every batch
    of the window is mapped and unmapped once during set-up - the state of an engine that has been running for a
    second. prefault=False skips the sweep (the timed steps then touch VA that was never mapped)."""
    window = max(32, steps + warmup) if burst else 32  # 64 GiB of VA
    pool = Pool(capi, device, window, mode, pool_mb, page=page, compound_layers=compound_layers, backend=backend)
    slot = pool.slot
    try:
        if fanout is not None:
            mapper = lambda offs: fanout.map_to_kv_tensors(offs)     # noqa: E731  rank 0's offsets win
            unmapper = lambda offs: fanout.unmap_from_kv_tensors(offs)  # noqa: E731
        else:
            mapper, unmapper = capi.map_to_kv_tensors, capi.unmap_from_kv_tensors
        if prefault:
            for b in range(window * int(os.environ.get("KVC_BENCH_SETUP_SWEEPS", "1"))):
                b %= window
                offs = batch_offsets(b, slot=slot)
                mapper(offs)
                unmapper(offs)
        # warm-up: W full cycles (code paths warm, the idle-handle pool holds what its cap allows)
        for b in range(warmup):
            offs = batch_offsets(b % window, slot=slot)
            mapper(offs)
            unmapper(offs)
        capi.set_option(capi.OPT_PROFILE, 1)
        capi.reset_stats()
        c0 = [int(capi.get_option(k)) for k in (112, 113, 114, 115)]   # KFD alloc / export / DRM import ns, creations
        if barrier:
            barrier()
        if sync:
            sync()
        per_map, per_unmap = [], []
        # the inputs of the timed steps are ready before the clock starts: the offsets of every batch, already in the form
        # the boundary takes them in (an int64 array for the C ABI; the TP fan-out pickles Python lists)
        ready = [batch_offsets((warmup + i) % window, slot=slot) for i in range(steps)]
        if fanout is None:
            ready = [capi.i64_array(o) for o in ready]
        else:
            import numpy as np
            ready = [np.asarray(o, dtype=np.int64) for o in ready]   # packed into the broadcast message with one copy
        t0 = time.perf_counter()
        for i in range(steps):
            offs = ready[i]
            ta = time.perf_counter()
            mapper(offs)
            tb = time.perf_counter()
            per_map.append(tb - ta)
            if not burst:
                unmapper(offs)
                per_unmap.append(time.perf_counter() - tb)
        capi.flush_unmaps()   # nothing of the timed work is left owed or in flight on the library's own thread
        if fanout is not None:
            fanout.finish()   # ... nor unagreed between the ranks
        if sync:
            sync()
        if barrier:
            barrier()
        elapsed = time.perf_counter() - t0
        st = capi.get_stats()
        c1 = [int(capi.get_option(k)) for k in (112, 113, 114, 115)]
        n_created = c1[3] - c0[3]
        create_split = ({"driver_allocations": n_created, "kfd_alloc_us": round((c1[0] - c0[0]) / n_created / 1e3, 2),
                         "kfd_export_us": round((c1[1] - c0[1]) / n_created / 1e3, 2),
                         "drm_import_us": round((c1[2] - c0[2]) / n_created / 1e3, 2)} if n_created else None)
        st["driver_ns"] = capi.get_driver_breakdown()
        st["host_segments_ns"] = {name: int(capi.get_option(130 + i)) for i, name in HOST_SEGMENTS.items()}
        capi.set_option(capi.OPT_PROFILE, 0)
        if burst:  # give everything back, outside the timed region
            for i in range(steps):
                tb = time.perf_counter()
                unmapper(batch_offsets((warmup + i) % window, slot=slot))
                per_unmap.append(time.perf_counter() - tb)
        return {"elapsed": elapsed, "per_step": per_map, "per_unmap": per_unmap, "stats": st, "reserve_s": pool.reserve_s,
                "window_GiB": pool.size / GiB, "burst": burst,
                "backend_in_effect": {0: "hip", 2: "hybrid", 3: "drm"}.get(int(capi.get_option(108)), "?"),
                "kfd_create": int(capi.get_option(110)), "kfd_tlb_flush": int(capi.get_option(118)), "prt": int(capi.get_option(128)),
                "max_extent_pages": int(capi.get_option(119)), "create_split": create_split}
    finally:
        pool.close()


def summarize(res, steps, n_gpus=1):
    bytes_per_step = BATCH_PAGES * PAGE
    st = res["stats"]
    t_map, t_unmap = sum(res["per_step"]), sum(res["per_unmap"])
    out = {
        "GBps": n_gpus * steps * bytes_per_step / res["elapsed"] / 1e9,
        "ms_per_step": res["elapsed"] / steps * 1e3,
        "map_zero_GBps": n_gpus * steps * bytes_per_step / t_map / 1e9,
        "p50_map_batch_ms": statistics.median(res["per_step"]) * 1e3,
        "p90_map_batch_ms": sorted(res["per_step"])[int(0.9 * (len(res["per_step"]) - 1))] * 1e3,
        "map_us_per_page": t_map / steps / BATCH_PAGES * 1e6,
        "unmap_us_per_page": t_unmap / max(1, len(res["per_unmap"])) / BATCH_PAGES * 1e6,
        "unmap_GBps": len(res["per_unmap"]) * bytes_per_step / max(t_unmap, 1e-9) / 1e9,
        "handles_created": st["handles_created"], "handles_reused": st["handles_reused"],
        "va_reserve_and_backfill_s": res["reserve_s"],
        "driver_us_per_page": {k: round(v / 1e3 / (steps * BATCH_PAGES), 2) for k, v in st.get("driver_ns", {}).items() if v},
        "tlb_shootdown_us": round(st["shootdown_ns"] / 1e3 / max(1, st["tlb_shootdowns"]), 1),
        "host_us_per_call": {k: round(v / 1e3 / steps, 1) for k, v in st.get("host_segments_ns", {}).items() if v},
    }
    return out


def roofline_from(st):
    launches, ms, nbytes = st["fill_launches"], st["fill_ms"], st["fill_bytes"]
    if not launches or ms <= 0:
        return None
    achieved = nbytes / (ms * 1e-3) / 1e9
    traffic = None
    tpath = os.path.join(REPO, "profiles", "zero_fill_traffic.json")
    if os.path.exists(tpath):
        try:
            with open(tpath) as f:
                traffic = json.load(f).get("hbm_bytes_per_launch")
        except Exception:
            traffic = None
    return {"kernel": "zero_fill_pages", "bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBPS,
            "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBPS, 4), "traffic": traffic,
            "traffic_source": "profiles/zero_fill_traffic.json: rocprofv3 --pmc WRITE_SIZE / FETCH_SIZE in separate passes over this "
                              "same command (tools_gpu_round.sh), per launch, gfx950 unit corrections applied; not collected "
                              "inside this run" if traffic is not None else None,
            "launches": launches, "bytes_per_launch": nbytes // launches,
            "avg_launch_us": round(ms / launches * 1e3, 2)}


def compaction_roofline(capi, device):
    """Secondary kernel: compact_blocks on the Llama-3-8B geometry (64 regions = 32 layers x K/V, 32 KiB blocks,
    2048 disjoint moves = 4 GiB read + 4 GiB written). Event-timed like zero_fill_pages. Next to it, in the same session, a
    plain copy on this box: the same kernel on ONE region with 2 MiB blocks i -> i + 1024, i.e. a contiguous 2 GiB -> 2 GiB copy
    (rounds 1-3 called it the ceiling; since every XCD works inside its own eighth of the regions the scattered blocks move
    FASTER than that stream - DESIGN.md §5), and torch's own D2D copy."""
    import numpy as np
    import torch

    def timed(bases, src, dst, block, reps=5):
        for _ in range(2):
            capi.compact_blocks(bases, src, dst, block)
        capi.set_option(capi.OPT_PROFILE, 1)
        capi.reset_stats()
        for _ in range(reps):
            capi.compact_blocks(bases, src, dst, block, sync=False)
        capi.compact_blocks(bases[:1], src[:1], dst[:1], block, sync=True)
        st = capi.get_stats()
        capi.set_option(capi.OPT_PROFILE, 0)
        return st

    block, regions, n_blocks, moves = 32 * 1024, 64, 4096, 2048
    capi.init(device, PAGE, False)
    if os.environ.get("KVC_BENCH_COMPACT_VARIANT"):                 # A/B runs of the kernel's placement variants; default: the library's
        capi.set_option(capi.OPT_COMPACT_VARIANT, int(os.environ["KVC_BENCH_COMPACT_VARIANT"]))
    try:
        # The data where it lives in the product: 64 regions (32 layers x K/V) reserved and backed BY THE LIBRARY - 64 page ids,
        # every 2 MiB slot its own page-table entry (8 GiB) - with random contents (what is moved is bytes, not zeros).
        ids = np.random.default_rng(0).permutation(n_blocks)[:2 * moves]
        src, dst = [int(x) for x in ids[:moves]], [int(x) for x in ids[moves:]]
        regs = capi.create_kv_tensors(2 * (n_blocks * block), 1, device, regions // 2, 2, 0, False)
        page_ids = n_blocks * block // PAGE
        capi.map_to_kv_tensors([p * PAGE for p in range(page_ids)])
        bases = capi.get_region_bases(0)
        assert len(bases) == regions, (len(bases), regions)
        import ctypes
        hip = ctypes.CDLL("libamdhip64.so")                         # (the runtime torch has loaded already)
        noise = torch.randint(0, 127, (n_blocks * block,), dtype=torch.int8, device=device)
        torch.cuda.synchronize()
        for b in bases:                                             # device-to-device (kind 3) into the mapped range
            rc = hip.hipMemcpy(ctypes.c_void_p(b), ctypes.c_void_p(noise.data_ptr()), ctypes.c_size_t(n_blocks * block), 3)
            assert rc == 0, f"hipMemcpy into a KV region failed: {rc}"
        del noise
        torch.cuda.synchronize()
        st = timed(bases, src, dst, block)
        achieved = st["compact_bytes"] / (st["compact_ms"] * 1e-3) / 1e9
        out = {"kernel": "compact_blocks (LDS-staged, a contiguous eighth of the pairs per XCD, non-temporal, 32 KiB tiles)", "bound": "hbm", "achieved": round(achieved, 1),
               "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBPS, 4),
               "bytes": "read + written = 2 x block_bytes x regions per moved block (4 MiB per Llama-3-8B block)",
               "where": f"{regions} KV regions reserved and backed by the library ({page_ids} page ids = {regions * page_ids} slots of 2 MiB), random pairing of {moves} moves",
               "launches": st["compact_launches"], "bytes_per_launch": st["compact_bytes"] // st["compact_launches"],
               "avg_launch_us": round(st["compact_ms"] / st["compact_launches"] * 1e3, 2)}
        # the same regions, moves as KVCacheManager.plan_compaction makes them on pages that are 30 % full at random:
        # donors' live blocks ascending into receivers' free blocks ascending, instead of a random pairing
        sys.path.insert(0, os.path.join(REPO, "benchmarks"))
        from bench_compact import planned_moves
        psrc, pdst = planned_moves(n_blocks, PAGE // block)
        stp = timed(bases, psrc, pdst, block)
        out["planner_moves_GBps"] = round(stp["compact_bytes"] / (stp["compact_ms"] * 1e-3) / 1e9, 1)
        out["planner_moves"] = f"{len(psrc)} moves per region as KVCacheManager.plan_compaction orders them on 30 %-occupied pages (seed 2)"
        capi.unmap_from_kv_tensors([p * PAGE for p in range(page_ids)])
        # the same moves on 64 torch buffers (what rounds 1-3 measured; physically contiguous 128 MiB each)
        bufs = [torch.randint(0, 127, (n_blocks * block,), dtype=torch.int8, device=device) for _ in range(regions)]
        torch.cuda.synchronize()
        stt = timed([b.data_ptr() for b in bufs], src, dst, block)
        out["on_torch_buffers_GBps"] = round(stt["compact_bytes"] / (stt["compact_ms"] * 1e-3) / 1e9, 1)
        del bufs
        big = torch.randint(0, 127, (4 * GiB,), dtype=torch.int8, device=device)
        torch.cuda.synchronize()
        stc = timed([big.data_ptr()], list(range(1024)), list(range(1024, 2048)), 2 * MiB)
        ceiling = stc["compact_bytes"] / (stc["compact_ms"] * 1e-3) / 1e9
        a, b = big[:2 * GiB], big[2 * GiB:]
        b.copy_(a)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            b.copy_(a)
        e1.record()
        torch.cuda.synchronize()
        out["copy_ceiling_GBps"] = round(ceiling, 1)
        out["copy_ceiling"] = ("the same kernel, same session, on ONE region with 2 MiB blocks i -> i + 1024: a contiguous 2 GiB -> 2 GiB copy, "
                               "read + written bytes over event time (a reference point, not a bound: two plain streams 2 GiB apart)")
        out["frac_of_copy_ceiling"] = round(achieved / ceiling, 4)
        out["torch_d2d_copy_GBps"] = round(5 * 2 * 2 * GiB / (e0.elapsed_time(e1) * 1e-3) / 1e9, 1)
        return out
    finally:
        capi.shutdown()


def engine_geometry(capi, device, mode):
    """The non-contiguous Llama-3-8B geometry (the reference forces it on ROCm: kvcached/utils.py:150-171; one page id = 32 layers x
    K/V = 64 slots of 2 MiB in 64 places, csrc/allocator.cpp:189-206) through the same C ABI: 1 / 8 / 64 consecutive page ids
    per call, pool warm. 'steady': every call finds the address space as its previous visit left it; 'straddling': the runs are
    shifted against that shape (rest mappings are split, their remainders rewritten); 'scattered': single page ids of a
    churned free list. Per-call p50, GB/s backed by the map call, and where the host time goes."""
    import importlib.util
    import numpy as np
    spec = importlib.util.spec_from_file_location("bench_engine_geometry", os.path.join(REPO, "benchmarks", "bench_engine_geometry.py"))
    eg = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(eg)
    os.environ["KVCACHED_ZERO_BACKFILL"] = "true" if mode == "compat" else "false"
    half = 512                                            # page ids per K (or V) half of a layer: 64 GiB of VA in all
    capi.init(device, PAGE, False)
    out = {}
    try:
        capi.create_kv_tensors(2 * half * PAGE, 1, device, 32, 2, 0, False)
        rng = np.random.default_rng(0)
        for placement, ids in (("steady", (1, 8, 64)), ("straddling", (1, 8)), ("scattered", (8,))):
            for n in ids:
                r = eg.run(capi, n, placement, 30, half, 32, rng)
                name = f"engine_llama3_8b_noncontig_{n}_page_id{'s' if n > 1 else ''}" + ("" if placement == "steady" else f"_{placement}")
                out[name] = {"p50_map_ms": r["map_ms"]["p50"], "max_map_ms": r["map_ms"]["max"], "p50_unmap_ms": r["unmap_ms"]["p50"],
                             "map_GBps": r["map_GBps_backed_p50"], "cycle_GBps": r["cycle_GBps"], "us_per_2MiB_map": r["us_per_2MiB_map"],
                             "slots_2MiB": r["slots_2MiB"], "ioctls_per_map_call": r["per_call"]["map.ioctls_issued"],
                             "ioctls_per_unmap_call": r["per_call"]["unmap.ioctls_issued"], "tlb_shootdown_us": r["shootdown_us"],
                             "driver_allocations": r["driver_allocations_in_timed_region"], "host_us_per_call": r["host_us_per_call"]}
        out["lanes_per_buffer"] = int(capi.get_option(129))
    finally:
        capi.shutdown()
        os.environ.pop("KVCACHED_ZERO_BACKFILL", None)
    return out


def cpu_baseline():
    """The reference's CPU path restated (oracle/, kind "port"), 1 thread, bounded sample: for each
    batch the page-id bookkeeping of the reference allocator (PageAllocator state machine ->
    offsets) and the CPU statement of map+zero on host memory (anonymous mmap like the reference's
    cpu device, csrc/ftensor.cpp:40-44; CPUPage::map is a no-op, so "zero" = memset)."""
    import ctypes
    import mmap
    sys.path.insert(0, os.path.join(REPO, "tests"))
    import kvc_testlib as T
    lib = T.load_oracle()
    batches = 24                                                       # ~10 s of single-core work on the GPU box
    size = batches * BATCH_PAGES * PAGE
    buf = mmap.mmap(-1, size, flags=mmap.MAP_PRIVATE | mmap.MAP_ANONYMOUS)   # reserved, committed on first touch
    anchor = ctypes.c_char.from_buffer(buf)
    base = ctypes.addressof(anchor)
    pa = T.OraclePA.create(lib, 1, size, PAGE, num_kv_buffers=1, max_res=0, min_res=0)
    dt = 0.0
    for b in range(batches):
        t0 = time.perf_counter()
        pids = [pa.alloc_page() for _ in range(BATCH_PAGES)]           # bookkeeping
        ev = pa.drain_events()
        offs = [o for _, os_ in ev for o in os_]
        assert len(offs) == BATCH_PAGES and len(pids) == BATCH_PAGES
        ptrs = (ctypes.c_void_p * BATCH_PAGES)(*[base + o for o in offs])
        lib.okvc_zero_fill_pages(ptrs, BATCH_PAGES, PAGE)               # "map" is a no-op on the cpu device
        dt += time.perf_counter() - t0
        buf.madvise(mmap.MADV_DONTNEED, min(offs), BATCH_PAGES * PAGE)  # untimed: keep the resident set at one batch
        done = b + 1
        if dt > 12.0:                                                   # bounded sample on a slow host
            break
    batches = done
    pa.close()
    del ptrs, anchor
    buf.close()
    return {"value": round(batches * BATCH_PAGES * PAGE / dt / 1e9, 3), "unit": "GB/s", "cores": 1, "kind": "port",
            "sample": f"{batches} batches x {BATCH_PAGES} x 2 MiB: oracle PageAllocator bookkeeping + memset of "
                      f"first-touched anonymous host memory ({dt:.1f} s)",
            "host_cores_available": os.cpu_count()}


REF_SNIPPET = r"""
import importlib.machinery, importlib.util, json, sys, time, torch
so, steps, warmup, burst = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), sys.argv[4] == "burst"
loader = importlib.machinery.ExtensionFileLoader("vmm_ops", so)
spec = importlib.util.spec_from_loader("vmm_ops", loader)
ref = importlib.util.module_from_spec(spec); loader.exec_module(ref)
PAGE, N = 2 << 20, 1024
import numpy as np
window = max(32, steps + warmup)
offs = lambda b: [int((b % window) * N + p) * PAGE for p in np.random.default_rng(b % window).permutation(N)]
ref.init_kvcached("cuda:0", PAGE, False)
t0 = time.perf_counter()
ref.create_kv_tensors(window * N * PAGE, 1, "cuda:0", 1, 1, 0, True)
t_create = time.perf_counter() - t0
for b in range(warmup):
    assert ref.map_to_kv_tensors(offs(b)); assert ref.unmap_from_kv_tensors(offs(b))
torch.cuda.synchronize()
pm, pu = [], []
t_all = time.perf_counter()
for i in range(steps):
    o = offs(warmup + i)
    ta = time.perf_counter(); assert ref.map_to_kv_tensors(o); tb = time.perf_counter(); pm.append(tb - ta)
    if not burst:
        assert ref.unmap_from_kv_tensors(o); pu.append(time.perf_counter() - tb)
torch.cuda.synchronize()
t_all = time.perf_counter() - t_all
if burst:
    for i in range(steps):
        tb = time.perf_counter(); assert ref.unmap_from_kv_tensors(offs(warmup + i)); pu.append(time.perf_counter() - tb)
print(json.dumps({"GBps": steps * N * PAGE / t_all / 1e9, "map_GBps": steps * N * PAGE / sum(pm) / 1e9,
                  "p50_map_batch_ms": sorted(pm)[len(pm)//2] * 1e3, "map_us_per_page": sum(pm) / steps / N * 1e6,
                  "unmap_us_per_page": sum(pu) / steps / N * 1e6, "va_reserve_and_backfill_s": t_create,
                  "steps": steps, "warmup": warmup, "window_GiB": window * 2, "workload": "burst" if burst else "cycle"}))
ref.shutdown_kvcached()
"""


def reference_on_box(steps, warmup, burst=False):
    """The REAL reference (oracle/_ref/vmm_ops.so, compiled from its own sources in the build
    container) running its own HIP path on this GPU on the SAME workload (same window, warm-up, steps and
    offsets): its FTensor::map per page = unmap zero alias + hipMemCreate + hipMemMap + hipMemSetAccess,
    no zero fill, no TLB invalidation (so its pages are not even reliably private).
    Context only."""
    so = os.path.join(REPO, "oracle", "_ref", "vmm_ops.so")
    if not os.path.exists(so):
        return None
    try:
        out = subprocess.run([sys.executable, "-c", REF_SNIPPET, so, str(steps), str(warmup), "burst" if burst else "cycle"],
                             capture_output=True, text=True, timeout=600)
        line = [l for l in out.stdout.splitlines() if l.startswith("{")]
        if out.returncode != 0 or not line:
            return {"error": (out.stderr or out.stdout)[-300:]}
        d = json.loads(line[-1])
        d["what"] = "reference csrc (KVCACHED_USE_HIP) map_to_kv_tensors on the same workload, no zero fill"
        return {k: (round(v, 3) if isinstance(v, float) else v) for k, v in d.items()}
    except Exception as e:  # never let the context number break the bench line
        return {"error": str(e)[:200]}


# ------------------------------------------------------------------ shared pool (BASELINE config 4)
def shared_pool_leg(rank, world, device, backend, rehearsal):
    """BASELINE.json configs[3]: "TP=8 shared KV pool: rank-0 create + IPC export, ranks 1-7 map over xGMI" (the reference's own
    harness only fans offsets out: benchmarks/bench_tp_ipc/kvcached_tp_ipc_benchmark.py:117-212, kvcached/tp_ipc_util.py:173-192).
    Llama-3-8B geometry on every rank (32 layers x K/V, same VA layout). Per round: rank 0 backs k page ids and writes a
    signature into each of them; SharedPoolChannel.share(): offsets over the group's collective (RCCL broadcast), descriptors over
    SCM_RIGHTS, every other rank imports and maps the SAME physical pages at the same offsets; each of them reads the signature
    through its own mapping with a kernel on ITS GPU (rank 0's memory: the read crosses xGMI) and the ranks agree on the result;
    everything is unmapped again. Twice: with page ids as units (ONE dmabuf per page id: the buffer its lane lives in, DESIGN.md
    §4.11) and slot by slot (one per 2 MiB slot, exportable single pages: round 2's form). Reports ms per page id and which
    import path the peers took (straight into KFD + DRM, or the runtime's import - the fallback for a buffer the direct path
    refuses)."""
    import torch
    import torch.distributed as dist
    from kvcached_amd import capi, vmm_ops
    from kvcached_amd.tp_ipc_util import CollectiveFanout, SharedPoolChannel
    L, half = 32, 64
    cdev = device if backend == "nccl" else "cpu"
    fan = CollectiveFanout(device=cdev)
    red = lambda v, op: (lambda t: (dist.all_reduce(t, op=op), float(t.item()))[1])(torch.tensor([v], dtype=torch.float64, device=cdev))   # noqa: E731
    med = lambda v: round(statistics.median(v) * 1e3, 3)   # noqa: E731
    out = {"ranks": world, "layers": L, "transport": f"{backend} broadcast of the offsets + all-reduce(min) of the status; dmabuf fds over SCM_RIGHTS"}
    for units in ("page_ids", "slots"):
        exporter = importer = None
        seen = []
        if rehearsal:   # the library's cpu device maps nothing and exports nothing: stand-ins that prove the control flow and that fds travel
            per = 1 if units == "page_ids" else 2 * L      # (the stand-in exports one descriptor per page id: nothing here knows about buffers)
            exporter = lambda offs, gid, per=per: [os.memfd_create(f"kvc_rehearsal_{o}") for o in offs for _ in range(per)]   # noqa: E731
            importer = lambda offs, fds, gid, meta: seen.append((len(offs), sum(1 for fd in fds if os.fstat(fd).st_size == 0)))   # noqa: E731
        if units == "slots":
            os.environ["KVCACHED_EXPORTABLE_HANDLES"] = "1"
        else:
            os.environ.pop("KVCACHED_EXPORTABLE_HANDLES", None)
        res = {}
        try:
            vmm_ops.init_kvcached(device, PAGE, False)
            ts = vmm_ops.create_kv_tensors(2 * half * PAGE, 8, device, L, 2, 0, False)     # int64 elements
            imp0 = (capi.get_option(163), capi.get_option(164))
            chan = SharedPoolChannel(fan, exporter=exporter, importer=importer, units=units)
            epp = PAGE // 8
            for k in (1, 8):
                back, share, verify, ok_all = [], [], [], True
                for it in range(6):
                    ids = [(it * k + j) % half for j in range(k)]
                    offs = [p * PAGE for p in ids]
                    sig = 0x5EED0000 + 977 * it + k
                    dist.barrier()
                    t0 = time.perf_counter()
                    if rank == 0:
                        capi.map_to_kv_tensors(offs)
                        if not rehearsal:
                            for t in (ts[0], ts[L - 1]):                      # first K row and last V row of every page id
                                for p in ids:
                                    t[p * epp] = sig + p
                                    t[(half + p) * epp + 5] = sig - p
                            torch.cuda.synchronize()
                    t1 = time.perf_counter()
                    chan.share(offs)
                    t2 = time.perf_counter()
                    good = 1.0
                    if rank != 0 and not rehearsal:
                        for t in (ts[0], ts[L - 1]):
                            idx = torch.tensor([p * epp for p in ids] + [(half + p) * epp + 5 for p in ids], device=device)
                            got = (t[idx] + 0).cpu().tolist()              # a gather kernel on THIS rank's GPU reads rank 0's pages
                            good = good if got == [sig + p for p in ids] + [sig - p for p in ids] else 0.0
                    elif rank != 0:
                        good = 1.0 if seen and seen[-1] == (k, k * (1 if units == "page_ids" else 2 * L)) else 0.0
                    t3 = time.perf_counter()
                    ok_all = ok_all and red(good, dist.ReduceOp.MIN) == 1.0
                    if rank != 0:
                        capi.unmap_from_kv_tensors(offs)                      # the imports are dropped first ...
                    dist.barrier()
                    if rank == 0:
                        capi.unmap_from_kv_tensors(offs)                      # ... then the owner gives the pages back
                    if it:                                                    # (the first round warms sockets and pools)
                        back.append(t1 - t0)
                        share.append(red(t2 - t1, dist.ReduceOp.MAX))
                        verify.append(red(t3 - t2, dist.ReduceOp.MAX))
                res[f"{k}_page_ids"] = {"slots_2MiB": k * 2 * L, "descriptors": (("one per buffer: at most " if not rehearsal else "") + str(k)) if units == "page_ids" else k * 2 * L,
                                        "rank0_back_ms_p50": med(back), "export_ship_import_map_ms_p50_slowest_rank": med(share),
                                        "ms_per_page_id": round(statistics.median(share) * 1e3 / k, 3),
                                        "us_per_slot_per_peer": round(statistics.median(share) * 1e6 / (k * 2 * L) / max(1, world - 1), 2),
                                        "peers_read_rank0_signature_ms_p50": med(verify), "signature_seen_by_every_peer": ok_all}
            # which way the peers' imports went (each rank reports its own counters; rank 0 imports nothing)
            mine = torch.tensor([float(capi.get_option(163) - imp0[0]), float(capi.get_option(164) - imp0[1])], dtype=torch.float64, device=cdev)
            every = [torch.zeros_like(mine) for _ in range(world)]
            dist.all_gather(every, mine)
            res["imports_by_rank"] = [{"rank": r, "straight_into_KFD_and_DRM": int(e[0]), "through_the_runtime_ROCr": int(e[1])} for r, e in enumerate(every)]
            chan.close()
        except Exception as e:   # (a rank that fails here takes the others with it through the collectives' timeouts: the mode is reported as failed)
            res["error"] = f"{type(e).__name__}: {str(e)[:300]}"
        finally:
            vmm_ops.shutdown_kvcached()
            os.environ.pop("KVCACHED_EXPORTABLE_HANDLES", None)
        out["page_ids_as_units" if units == "page_ids" else "slot_by_slot"] = res
    gpus = torch.tensor([float(torch.cuda.current_device()) if not rehearsal else -1.0], dtype=torch.float64, device=cdev)
    seen_gpus = [torch.zeros_like(gpus) for _ in range(world)]
    dist.all_gather(seen_gpus, gpus)
    distinct = len({int(g.item()) for g in seen_gpus})
    out["gpus_used"] = distinct if not rehearsal else 0
    out["what"] = ("cross-GPU: rank 0's pages are read by the peers over xGMI" if distinct == world and not rehearsal else
                   "REHEARSAL on the library's cpu device: offsets and file descriptors travel, nothing is mapped" if rehearsal else
                   f"REHEARSAL: {world} ranks share {distinct} GPU(s) - the export/ship/import/map machinery runs end to end, but nothing crosses xGMI")
    return out


def run_shared_pool_child(args):
    """One rank of the leg, in a process of its own: RANK / WORLD_SIZE / LOCAL_RANK / MASTER_* come from the parent rank."""
    import torch
    import torch.distributed as dist
    rank, world, local_rank = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"]), int(os.environ.get("LOCAL_RANK", "0"))
    rehearsal = os.environ.get("KVC_BENCH_REHEARSAL") == "cpu"
    backend = "gloo" if rehearsal else os.environ.get("KVC_BENCH_BACKEND", "nccl")
    os.environ.setdefault("KVCACHED_LOG_LEVEL", "ERROR")
    if rehearsal:
        device = "cpu"
    else:
        local_rank %= max(1, torch.cuda.device_count())
        torch.cuda.set_device(local_rank)
        device = f"cuda:{local_rank}"
    if backend == "nccl":
        dist.init_process_group("nccl", device_id=torch.device(device))
    else:
        dist.init_process_group(backend)
    try:
        out = shared_pool_leg(rank, world, device, backend, rehearsal)
        out["rccl_world_size"] = dist.get_world_size()
        if rank == 0:
            print(json.dumps(out), flush=True)
        dist.barrier()
    finally:
        dist.destroy_process_group()


def shared_pool_rehearsal_on_one_gpu():
    """N = 1 has no peers: the leg is REHEARSED with two child ranks that share this GPU (gloo for the collective - RCCL refuses two
    ranks on one device), so that the driver's single-GPU record shows the export / SCM_RIGHTS / import / map machinery running and the
    peers reading rank 0's signature on this very box. Labelled as a rehearsal in the result; nothing crosses xGMI."""
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK="0", WORLD_SIZE="2", LOCAL_WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   KVC_BENCH_BACKEND="gloo", KVC_BENCH_SHARE_GPUS="1", KVCACHED_IPC_NAME=f"kvc_bench_share_{port}")
        env.pop("KVC_BENCH_FORCE_DIST", None)
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__), "--shared-pool-leg"], env=env, text=True,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL, stderr=subprocess.PIPE if r == 0 else subprocess.DEVNULL))
    try:
        out, err = procs[0].communicate(timeout=180)
        for p in procs[1:]:
            p.wait(timeout=60)
    except subprocess.TimeoutExpired:
        for p in procs:
            if p.poll() is None:
                p.kill()
        return {"error": "the one-GPU rehearsal of the shared-pool leg did not finish within 180 s"}
    js = [l for l in out.splitlines() if l.startswith("{")]
    if procs[0].returncode != 0 or not js:
        return {"error": f"rank 0 of the rehearsal exited with status {procs[0].returncode}", "stderr_tail": (err or out)[-400:]}
    return json.loads(js[-1])


def shared_pool_in_children(rank, world):
    """Every rank of the N > 1 run starts ONE child that takes its place in the leg (own rendezvous port, own IPC name, the
    same GPU): whatever happens there - the cross-GPU import has never run on this code before an 8-GPU node sees it - the
    parent's measurement and its JSON line are safe. Rank 0 returns the leg's summary (or the error)."""
    import socket
    import torch.distributed as dist
    env = dict(os.environ)
    box = [None]
    if rank == 0:                                                # a port that is free right now, agreed on through the parents' own group
        with socket.socket() as s:
            s.bind(("127.0.0.1", 0))
            box[0] = s.getsockname()[1]
    dist.broadcast_object_list(box, src=0)
    port = int(box[0])
    env["MASTER_PORT"] = str(port)
    env["KVCACHED_IPC_NAME"] = f"kvc_bench_share_{port}"      # the same on every rank: the fd sockets live under one directory
    env.pop("KVC_BENCH_TEST_FAIL_RANK", None)
    try:
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "--shared-pool-leg"], env=env, capture_output=True, text=True, timeout=150)
    except subprocess.TimeoutExpired:
        return {"error": "the shared-pool leg did not finish within 150 s"}
    if rank != 0:
        return None
    js = [l for l in r.stdout.splitlines() if l.startswith("{")]
    if r.returncode != 0 or not js:
        return {"error": f"rank 0 of the leg exited with status {r.returncode}", "stderr_tail": (r.stderr or r.stdout)[-600:]}
    return json.loads(js[-1])



def ensure_built(local_rank: int) -> None:
    """The in-tree .so files normally travel with the repo; if they are missing, local rank 0 compiles them (hipcc,
    gfx950) while the other ranks wait. This only builds the product - there is no fallback to fall back to."""
    import glob
    pkg = os.path.join(REPO, "kvcached_amd")
    have = lambda: os.path.exists(os.path.join(pkg, "libkvcached_amd.so")) and glob.glob(os.path.join(pkg, "vmm_ops.*.so"))  # noqa: E731
    if have():
        return
    if local_rank == 0:
        subprocess.check_call([sys.executable, os.path.join(REPO, "kvcached_amd", "build.py")], cwd=REPO, stdout=sys.stderr)
        return
    t0 = time.time()
    while not have():
        if time.time() - t0 > 600:
            raise SystemExit("bench.py: the native extension was not built by local rank 0 within 10 minutes")
        time.sleep(1.0)
    time.sleep(2.0)  # let the linker finish writing


def spawn_ranks(n: int) -> int:
    """`python bench.py --gpus N` without a launcher: this process becomes the launcher. It starts N copies of this
    script, one rank per GPU (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* in their environment, as torch.distributed.run
    would set them - the reference's harness spawns its ranks itself too,
    benchmarks/bench_tp_ipc/kvcached_tp_ipc_benchmark.py:131,212), relays rank 0's JSON line and returns the worst
    exit status. The launcher never imports torch and never touches HIP (nothing that has initialised the GPU is
    replaced or forked), and the children are plain child processes, not an exec of this one."""
    import socket
    ensure_built(0)   # once, before the ranks start (a subprocess; no GPU involved)
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    import tempfile
    procs = []
    with tempfile.TemporaryFile(mode="w+") as rank0_out:
        for r in range(n):
            env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                       MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
            env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
            env.setdefault("KVCACHED_IPC_NAME", f"kvc_bench_{os.getpid()}_r{r}")
            procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                          stdout=rank0_out if r == 0 else sys.stderr))
        failed = None
        while any(p.poll() is None for p in procs):
            failed = next((p for p in procs if p.poll() not in (None, 0)), None)
            if failed is not None:   # the others would wait for it in a collective for ever: end exactly those processes
                for p in procs:
                    if p.poll() is None:
                        p.kill()
                break
            time.sleep(0.1)
        rcs = [p.wait() for p in procs]
        if failed is not None:
            print(f"bench.py: rank {procs.index(failed)} exited with status {failed.returncode}; the other ranks were stopped",
                  file=sys.stderr)
            return failed.returncode if failed.returncode > 0 else 1
        rank0_out.seek(0)
        sys.stdout.write(rank0_out.read())
        sys.stdout.flush()
    bad = [rc for rc in rcs if rc != 0]
    return (bad[0] if bad[0] > 0 else 1) if bad else 0


def use_dist_requested() -> bool:
    return int(os.environ.get("WORLD_SIZE", "1")) > 1 or os.environ.get("KVC_BENCH_FORCE_DIST") == "1"


def main():
    args = parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(spawn_ranks(args.gpus))
    ensure_built(int(os.environ.get("LOCAL_RANK", "0")))
    if args.shared_pool_leg:
        return run_shared_pool_child(args)
    import torch
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and world > 1:
        args.gpus = world
    # stdout carries ONE line, the JSON: whatever the libraries below print there (gloo's "[Gloo] Rank 0 is connected to ..." of
    # the fan-out's wake group, a runtime's banner) is sent to stderr from here on, at the level of the file descriptor
    sys.stdout.flush()
    json_out = os.fdopen(os.dup(1), "w")
    os.dup2(2, 1)
    # Rehearsal of the launcher / fan-out / JSON contract without a GPU (tests only, KVC_BENCH_REHEARSAL=cpu): the
    # library's "cpu" device keeps the books and maps nothing, so the line it prints is labelled as no measurement.
    rehearsal = os.environ.get("KVC_BENCH_REHEARSAL") == "cpu"
    if os.environ.get("KVC_BENCH_TEST_FAIL_RANK") == str(rank):   # tests: a rank that dies before the rendezvous
        sys.exit(7)
    if not rehearsal and not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the HIP path has no CPU fallback")
    os.environ.setdefault("KVCACHED_IPC_NAME", f"kvc_bench_{os.getpid()}")
    os.environ.setdefault("KVCACHED_LOG_LEVEL", "ERROR")
    if rehearsal:
        device = "cpu"
    else:
        n_dev = torch.cuda.device_count()
        if world > n_dev and os.environ.get("KVC_BENCH_SHARE_GPUS") != "1":
            raise SystemExit(f"bench.py: {world} ranks but only {n_dev} GPUs visible (KVC_BENCH_SHARE_GPUS=1 puts several "
                             "ranks on one GPU for a rehearsal; its numbers are not a scaling measurement)")
        local_rank %= max(1, n_dev)
        torch.cuda.set_device(local_rank)
        device = f"cuda:{local_rank}"
    from kvcached_amd import capi

    if args.growth_burst_only:   # child of the N=1 run: the first GPU work of a fresh process
        os.environ.setdefault("KVCACHED_IPC_NAME", f"kvc_bench_gb_{os.getpid()}")
        r1 = measure(capi, device, 24, 0, args.mode, None, burst=True, prefault=False, backend=args.backend)
        s = summarize(r1, 24)
        out = {k: (round(s[k], 3) if isinstance(s[k], float) else s[k]) for k in
               ("GBps", "p50_map_batch_ms", "map_us_per_page", "handles_created", "driver_us_per_page")}
        out["create_split"] = r1.get("create_split")
        out["per_batch_ms"] = [round(x * 1e3, 1) for x in r1["per_step"]]
        print(json.dumps(out), file=json_out, flush=True)
        return

    # N = 1: before this process maps anything (it has only asked torch for the device), a child - a fresh process - runs the
    # growth burst as the first allocation work on the box -
    # VRAM the kernel has not handed out since boot is cleared inside the allocation (~80 us per 2 MiB), memory that was
    # wiped on release is not: the same burst later in this run (variants) shows the other side. DESIGN.md §4.5.
    first_touch = None
    if world == 1 and not rehearsal and not args.no_variants and not use_dist_requested():
        try:
            out = subprocess.run([sys.executable, os.path.abspath(__file__), "--growth-burst-only", "--backend", args.backend, "--mode", args.mode],
                                 capture_output=True, text=True, timeout=300)
            js = [l for l in out.stdout.splitlines() if l.startswith("{")]
            first_touch = json.loads(js[-1]) if out.returncode == 0 and js else {"error": (out.stderr or out.stdout)[-300:]}
            if "error" not in first_touch:
                first_touch["what"] = ("24 x 1024 x 2 MiB backed with nothing unmapped, as the first GPU work of a fresh process on "
                                       "this box: physical pages are allocated inside the timed region (create_split)")
        except Exception as e:
            first_touch = {"error": str(e)[:200]}

    fanout = barrier = None
    sync = None if rehearsal else torch.cuda.synchronize
    backend = "gloo" if rehearsal else os.environ.get("KVC_BENCH_BACKEND", "nccl")   # "gloo" only to rehearse the N>1 code path
    ranks_seen = [rank]
    use_dist = world > 1 or os.environ.get("KVC_BENCH_FORCE_DIST") == "1"   # forced: rehearse the N>1 path on one GPU
    if use_dist:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device(device))
        else:
            dist.init_process_group(backend)
        from kvcached_amd.tp_ipc_util import CollectiveFanout
        # the ranks' agreement on call i travels while call i+1 is being broadcast (deferred_status); measure() ends the timed
        # region with fanout.finish(): nothing is left unchecked inside the bracket
        fanout = CollectiveFanout(device=device if backend == "nccl" else "cpu", deferred_status=True)
        barrier = dist.barrier

    res = measure(capi, device, args.steps, args.warmup, args.mode, args.pool_mb, fanout, barrier, sync, backend=args.backend)
    elapsed = res["elapsed"]
    if use_dist:
        import torch.distributed as dist
        t = torch.tensor([elapsed], dtype=torch.float64, device=device if backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        res["elapsed"] = elapsed
        # who took part, and how many pages each rank really backed inside the timed region (value = their sum / max time)
        mine = torch.tensor([rank, res["stats"]["pages_mapped"]], dtype=torch.int64, device=device if backend == "nccl" else "cpu")
        everyone = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(everyone, mine)
        ranks_seen = sorted(int(e[0]) for e in everyone)
        pages_by_rank = [int(e[1]) for e in sorted(everyone, key=lambda e: int(e[0]))]
        assert all(p == pages_by_rank[0] for p in pages_by_rank), f"ranks backed different amounts: {pages_by_rank}"
    main_sum = summarize(res, args.steps, world)
    shared = None
    if use_dist and world > 1 and os.environ.get("KVC_BENCH_SHARED_POOL", "1") != "0":
        import torch.distributed as dist
        dist.barrier()                       # every rank is past its timed region and has shut its allocator down
        shared = shared_pool_in_children(rank, world)

    if rank == 0:
        line = {
            "metric": "GB/s KV backed (map+zero), 2 MiB pages",
            "value": round(main_sum["GBps"], 2),
            "unit": "GB/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(main_sum["ms_per_step"], 3),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "u8",
            "data": "synthetic" if not rehearsal else "REHEARSAL on the library's cpu device: bookkeeping only, nothing is mapped - not a measurement",
            "ranks": ranks_seen,
            "config": {"workload": "bench_vmm: 64 GiB VA window (one untimed warm-up sweep over the window during set-up, as in "
                                   "the protocol); step = map+zero then unmap one batch of 1024 x 2 MiB "
                                   "pages (shuffled offsets), both halves timed; the offset arrays of the timed steps are made before "
                                   "the clock starts. Physical pages are recycled through the "
                                   "library's pool: the timed steps create none (handles_created) - allocation from the driver "
                                   "is what growth_burst_* measure",
                       "kfd_create": res["kfd_create"], "kfd_tlb_flush": res["kfd_tlb_flush"], "max_extent_pages": res["max_extent_pages"],
                       "unbacked_va": "PRT mapping (reads 0, writes dropped, no fault)" if res["prt"] else ("zero aliases" if args.mode == "compat" else "unmapped"),
                       "mode": args.mode, "vmm_backend": res["backend_in_effect"] if not rehearsal else "none (cpu device)", "vmm_backend_requested": args.backend, "page_MiB": 2, "batch_pages": BATCH_PAGES,
                       "window_GiB": res["window_GiB"], "per_gpu_bytes_per_step": BATCH_PAGES * PAGE,
                       "fanout": f"{backend} broadcast + all-reduce(min)" if use_dist else "local"},
            "map_zero_GBps": round(main_sum["map_zero_GBps"], 2),
            "p50_map_batch_ms": round(main_sum["p50_map_batch_ms"], 3),
            "p90_map_batch_ms": round(main_sum["p90_map_batch_ms"], 3),
            "map_us_per_page": round(main_sum["map_us_per_page"], 3),
            "unmap_us_per_page": round(main_sum["unmap_us_per_page"], 3),
            "unmap_GBps": round(main_sum["unmap_GBps"], 2),
            "handles_created": main_sum["handles_created"], "handles_reused": main_sum["handles_reused"],
            "va_reserve_and_backfill_s": round(main_sum["va_reserve_and_backfill_s"], 3),
            "driver_us_per_page": main_sum["driver_us_per_page"], "tlb_shootdown_us": main_sum["tlb_shootdown_us"],
            "host_us_per_call": main_sum["host_us_per_call"],
            "roofline": roofline_from(res["stats"]),
        }
        if use_dist:
            import torch.distributed as dist
            line["rccl_world_size"] = dist.get_world_size()
        if shared is not None:
            line["shared_pool"] = shared
            for k in (1, 8):
                leg = (shared.get("page_ids_as_units") or {}).get(f"{k}_page_ids")
                if isinstance(leg, dict):
                    line["config"][f"shared_pool_{k}_page_ids_ms"] = leg["export_ship_import_map_ms_p50_slowest_rank"]
        if world == 1 and not rehearsal:
            if first_touch is not None:
                line["growth_burst_first_touch"] = first_touch
                line["config"]["growth_burst_first_touch_GBps"] = first_touch.get("GBps")
            if not args.no_variants:
                variants = {}
                # name -> (mode, pool cap MB, compound layers, burst, prefault, backend, page, extra environment, steps)
                V = lambda mode=args.mode, pool=None, comp=0, burst=False, pre=True, be=None, page=PAGE, env=None, n=24: \
                    (mode, pool, comp, burst, pre, be or args.backend, page, env or {}, n)   # noqa: E731
                table = {
                    "one_buffer_per_page_round1_default": V(env={"KVCACHED_PHYS_CHUNK_PAGES": "1"}),
                    "extents_up_to_16_pages": V(env={"KVCACHED_PHYS_CHUNK_PAGES": "16"}),
                    "extents_up_to_64_pages": V(env={"KVCACHED_PHYS_CHUNK_PAGES": "64"}),
                    "unmap_waits_for_its_own_tlb_invalidation": V(env={"KVCACHED_ASYNC_SHOOTDOWN": "false"}),
                    # compat, relaxed: the unmap's invalidation trails the call (<= 300 us) and is absorbed by the next map batch's own;
                    # the pages wait for it un-scrubbed and un-offered (one invalidation per cycle)
                    "compat_unmap_invalidation_trails_300us": V(env={"KVCACHED_UNMAP_INVALIDATION_US": "300"}),
                    "tlb_flush_through_hipMalloc_instead_of_kfd": V(env={"KVCACHED_KFD_TLB_FLUSH": "false"}),
                    "zero_extent_instead_of_prt": V(env={"KVCACHED_PRT": "false"}),
                    "lazy_mode_opt_in": V(mode="lazy"),
                    "lazy_mode_with_prt_behind_unbacked_va": V(mode="lazy", env={"KVCACHED_PRT": "true"}),
                    "lazy_mode_map_waits_for_all_invalidations": V(mode="lazy", env={"KVCACHED_MAP_WAITS_FOR_ALL_FLUSHES": "true"}),
                    "lazy_mode_fill_in_the_map_call": V(mode="lazy", env={"KVCACHED_SCRUB_ON_RELEASE": "false"}),
                    "compat_sharded_zero_pages_through_rocr_round1": V(mode="compat", n=8, env={"KVCACHED_ZERO_EXTENT": "false", "KVCACHED_PRT": "false"}),
                    "hybrid_backend_same_cycle": V(be="hybrid"),
                    "hip_backend_same_cycle": V(be="hip"),
                    "hip_backend_lazy": V(mode="lazy", be="hip"),
                    "fresh_va_window_warm_process": V(pre=False),
                    "growth_burst_24x2GiB_nothing_unmapped": V(burst=True, pre=False),   # growth = fresh VA, fresh handles
                    "growth_burst_one_buffer_per_page": V(burst=True, pre=False, env={"KVCACHED_PHYS_CHUNK_PAGES": "1"}),
                    "no_pool_every_handle_created_and_released": V(pool=0, n=8),
                    "page_size_8MiB_instead_of_2MiB": V(page=8 * MiB, n=8),
                    "contiguous_layout_128MiB_compound_pages": V(comp=32, n=8),
                }
                for name, (mode, pool, comp, burst, pre, be, page, env, nsteps) in table.items():
                    saved = {k: os.environ.get(k) for k in env}
                    try:
                        os.environ.update(env)
                        r1 = measure(capi, device, nsteps, 4, mode, pool, compound_layers=comp, burst=burst, prefault=pre,
                                     backend=be, page=page)
                        s = summarize(r1, nsteps)
                        variants[name] = {k: (round(s[k], 3) if isinstance(s[k], float) else s[k])
                                          for k in ("GBps", "map_zero_GBps", "p50_map_batch_ms", "map_us_per_page",
                                                    "unmap_us_per_page", "handles_created", "handles_reused",
                                                    "driver_us_per_page")}
                        rf = roofline_from(r1["stats"])
                        variants[name]["fill_GBps"] = rf["achieved"] if rf else None
                        if r1.get("create_split"):
                            variants[name]["create_split"] = r1["create_split"]
                    except Exception as e:
                        variants[name] = {"error": str(e)[:200]}
                    finally:
                        for k, v in saved.items():
                            if v is None:
                                os.environ.pop(k, None)
                            else:
                                os.environ[k] = v
                try:   # the geometry engines use on ROCm, first-class: compat (the default) and lazy
                    eg = engine_geometry(capi, device, args.mode)
                    variants.update({k: v for k, v in eg.items() if isinstance(v, dict)})
                    line["config"]["engine_lanes_per_buffer"] = eg.get("lanes_per_buffer")
                    for n, key in ((1, "engine_llama3_8b_noncontig_1_page_id"), (8, "engine_llama3_8b_noncontig_8_page_ids"),
                                   (64, "engine_llama3_8b_noncontig_64_page_ids")):
                        line["config"][f"engine_{n}_page_ids_map_GBps"] = eg[key]["map_GBps"]
                        line["config"][f"engine_{n}_page_ids_p50_map_ms"] = eg[key]["p50_map_ms"]
                    other = "lazy" if args.mode == "compat" else "compat"
                    for k, v in engine_geometry(capi, device, other).items():
                        if isinstance(v, dict) and "straddling" not in k and "scattered" not in k:
                            variants[k.replace("noncontig", f"noncontig_{other}")] = {kk: v[kk] for kk in ("p50_map_ms", "p50_unmap_ms", "map_GBps", "cycle_GBps", "ioctls_per_map_call")}
                except Exception as e:
                    variants["engine_llama3_8b_noncontig"] = {"error": str(e)[:300]}
                line["variants"] = variants
                # the reference's semantics (unbacked VA reads as zeros) next to the headline, not only among the variants
                line["config"]["compat_relaxed_unmap_invalidation_GBps"] = variants.get("compat_unmap_invalidation_trails_300us", {}).get("GBps")
                line["lazy_mode_GBps"] = variants.get("lazy_mode_opt_in", {}).get("GBps")
                line["lazy_mode_p50_map_batch_ms"] = variants.get("lazy_mode_opt_in", {}).get("p50_map_batch_ms")
                try:
                    line["roofline_compact_blocks"] = compaction_roofline(capi, device)
                except Exception as e:
                    line["roofline_compact_blocks"] = {"error": str(e)[:200]}
                try:   # BASELINE config 4 on one GPU: a rehearsal (two ranks share the device), so that the machinery has run on this box
                    line["shared_pool"] = shared_pool_rehearsal_on_one_gpu()
                except Exception as e:
                    line["shared_pool"] = {"error": str(e)[:200]}
            if not args.no_cpu_baseline:
                line["cpu_baseline"] = cpu_baseline()
                line["reference_hip_path_on_this_box"] = reference_on_box(args.steps, args.warmup)
                line["reference_hip_path_growth_burst"] = reference_on_box(24, 4, burst=True)
        print(json.dumps(line), file=json_out, flush=True)
    if use_dist:
        import torch.distributed as dist
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
