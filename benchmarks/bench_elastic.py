#!/usr/bin/env python3
"""bench_elastic — BASELINE.json config 3: Llama-3-8B GQA KV (32 layers, 8 KV heads, d=128, bf16)
growing and shrinking under a Poisson request trace on one MI355X.

Pass 1 (parity): the exact trace of tests/golden/manager_large.json case
"cfg3_llama3_8b_poisson_l4_60s" (8678 ops, recorded from the REAL reference) is replayed through the
real integration API with every page physically backed on the GPU; the hash chain over block ids,
page offsets and counters must equal the golden one (prealloc off, like the golden).
Pass 2..: lambda in {4, 16, 64} req/s, 120 s of virtual time replayed as fast as possible with the
prealloc thread ON (the production configuration): alloc()/free() latency percentiles, pages
mapped/unmapped, GB/s backed inside map calls, peak mapped bytes.

    python benchmarks/bench_elastic.py [--rates 4,16,64] [--duration 120] [--no-parity]
Prints one JSON object per pass.
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
sys.path.insert(0, os.path.join(REPO, "tests"))

os.environ.setdefault("KVCACHED_IPC_NAME", f"kvc_elastic_{os.getpid()}")
os.environ.setdefault("KVCACHED_LOG_LEVEL", "ERROR")
os.environ["KVCACHED_CONTIGUOUS_LAYOUT"] = os.environ.get("KVCACHED_CONTIGUOUS_LAYOUT", "false")

import torch  # noqa: E402

L, BLOCK, HEADS, DIM = 32, 16, 8, 128
CELL = HEADS * DIM * 2            # bytes per token per layer per K|V (bf16)
PAGE = 2 << 20
DEV = "cuda:0"


def pct(v, p):
    v = sorted(v)
    return v[min(len(v) - 1, int(p * len(v)))]


def setup(prealloc: bool):
    import kvcached_amd.kv_cache_manager as kcm
    import kvcached_amd.integration.vllm.interfaces as vi
    kcm.PAGE_PREALLOC_ENABLED = prealloc
    torch.cuda.set_device(0)
    t0 = time.perf_counter()
    vi.init_kvcached(tp_rank=0, world_size=1, is_worker=True)
    kv = vi.alloc_kv_cache((2, 1 << 20, BLOCK, HEADS, DIM), BLOCK, torch.bfloat16, DEV, L)
    startup = time.perf_counter() - t0
    num_blocks = kv[0].shape[1]
    return vi, kcm, kv, num_blocks, startup


def parity_pass():
    import kvc_testlib as T
    import kvcached_amd.kv_cache_manager as kcm
    from kvcached_amd import capi
    kcm.PAGE_PREALLOC_ENABLED = False      # the golden was recorded without the (timing-dependent) prealloc thread
    case = next(c for c in json.load(open(os.path.join(T.GOLDEN_DIR, "manager_large.json")))["cases"]
                if c["name"] == "cfg3_llama3_8b_poisson_l4_60s")
    cfg = case["config"]
    ad = T.ProductAdapter(cfg["num_blocks"], cfg["block_size"], cfg["cell_size"], cfg["num_layers"],
                          reserve_null_block=cfg["reserve_null_block"], num_kv_buffers=cfg["num_kv_buffers"],
                          contiguous=cfg["contiguous"], phys_pages=T.PHYS_CAP - 1, device=DEV, execute=True)
    try:
        init = {"s": ad.snapshot(), "e": ad.drain_events()}
        capi.reset_stats()
        t0 = time.perf_counter()
        recs = T.replay(ad, case["ops"], full=False)
        wall = time.perf_counter() - t0
        chain = T.chain_hash(recs)
        st = capi.get_stats()
        ok = init == case["init"] and chain["final"] == case["chain"]["final"] and \
            chain["checkpoints"] == case["chain"]["checkpoints"]
        return {"pass": "parity_cfg3_llama3_8b_poisson_l4_60s", "ops": len(case["ops"]), "bit_exact_vs_reference": ok,
                "chain_sha256": chain["final"], "golden_sha256": case["chain"]["final"], "wall_s": round(wall, 3),
                "pages_mapped_2MiB": st["pages_mapped"], "pages_unmapped_2MiB": st["pages_unmapped"],
                "GB_backed": round(st["pages_mapped"] * PAGE / 1e9, 2),
                "GBps_inside_map_calls": round(st["pages_mapped"] * PAGE / max(st["map_ns"], 1), 2)}
    finally:
        ad.close()


# host time of the map / unmap calls by segment (read-only options 130 + i of the library; bench.py HOST_SEGMENTS)
SEGMENTS = {1: "map: classify", 2: "map: runs", 3: "map: pool (incl. creating buffers)", 4: "map: page-table ioctls", 5: "map: bookkeeping",
            6: "map: remainder rewrite + TLB invalidation", 7: "map: own fill", 8: "map: wait for scrub", 11: "unmap: runs", 12: "unmap: page-table ioctls",
            14: "unmap: remainders", 16: "unmap: TLB invalidation", 17: "unmap: scrub launch", 18: "unmap: pool", 20: "map: remainder rewrite alone"}


def elastic_pass(rate: float, duration: float, seed: int = 1, passes: int = 2):
    """`passes` replays of the same trace in one process: the first grows into whatever VRAM the box hands out (memory the
    kernel has not wiped since boot is cleared inside the allocation, ~80 us per 2 MiB: DESIGN.md §4.5), the later ones find
    what the earlier ones gave back - the steady state of an engine that has been up for a minute."""
    from kvcached_amd import capi
    from kvcached_amd.traces import poisson_trace, trace_stats
    ops = poisson_trace(rate, duration, seed=seed)
    vi, kcm, kv, num_blocks, startup = setup(prealloc=True)
    try:
        m = kcm.KVCacheManager(num_blocks, BLOCK, CELL, L, world_size=1)
        assert m._post_init_done.wait(30)
        outs = []
        for k in range(passes):
            out = _replay(capi, m, ops, rate, duration, num_blocks, startup)
            out["replay"] = k + 1
            out.update(trace_stats(ops))
            outs.append(out)
        del m
        return outs
    finally:
        vi.shutdown_kvcached()


def _replay(capi, m, ops, rate, duration, num_blocks, startup):
    time.sleep(0.3)                  # let the prealloc thread fill the reserved pool
    capi.reset_stats()
    c0 = [int(capi.get_option(k)) for k in (112, 113, 114, 115)]
    live, lat_a, lat_f, none_count = {}, [], [], 0
    peak_pages, sum_pages = 0, 0
    pa = m.page_allocator
    t_all = time.perf_counter()
    for op in ops:
        if op[0] == "a":
            t0 = time.perf_counter()
            got = m.alloc(op[2])
            lat_a.append(time.perf_counter() - t0)
            if got is None:
                none_count += 1
            else:
                live[op[1]] = got
        else:
            blocks = live.pop(op[1], None)
            if blocks:
                t0 = time.perf_counter()
                m.free(blocks)
                lat_f.append(time.perf_counter() - t0)
        used = pa.get_num_inuse_pages()
        peak_pages = max(peak_pages, used)
        sum_pages += used
    wall = time.perf_counter() - t_all
    for blocks in live.values():
        m.free(blocks)
    m.trim()
    capi.flush_unmaps()
    st = capi.get_stats()
    c1 = [int(capi.get_option(k)) for k in (112, 113, 114, 115)]
    n_created = max(1, c1[3] - c0[3])
    mapped = max(st["pages_mapped"], 1)
    unit = PAGE * L * 2
    out = {"pass": f"elastic_poisson_lambda{rate:g}", "virtual_seconds": duration,
           "wall_s": round(wall, 3), "speedup_vs_realtime": round(duration / wall, 1),
           "alloc_us": {"p50": round(pct(lat_a, .5) * 1e6, 1), "p90": round(pct(lat_a, .9) * 1e6, 1),
                        "p99": round(pct(lat_a, .99) * 1e6, 1), "max": round(max(lat_a) * 1e6, 1),
                        "mean": round(statistics.mean(lat_a) * 1e6, 1)},
           "free_us": {"p50": round(pct(lat_f, .5) * 1e6, 1), "p99": round(pct(lat_f, .99) * 1e6, 1),
                       "max": round(max(lat_f) * 1e6, 1)} if lat_f else None,
           "alloc_returned_none": none_count,
           "pages_mapped_2MiB": st["pages_mapped"], "pages_unmapped_2MiB": st["pages_unmapped"],
           "map_calls": st["map_calls"], "GB_backed": round(st["pages_mapped"] * PAGE / 1e9, 2),
           "GBps_inside_map_calls": round(st["pages_mapped"] * PAGE / max(st["map_ns"], 1), 2),
           "us_per_2MiB_inside_map_calls": round(st["map_ns"] / 1e3 / max(st["pages_mapped"], 1), 2),
           "handles_created": st["handles_created"], "handles_reused": st["handles_reused"],
           "peak_mapped_GiB": round(peak_pages * unit / 2**30, 2),
           "mean_mapped_GiB": round(sum_pages / max(len(ops), 1) * unit / 2**30, 2),
           "pool_GiB_virtual": round(num_blocks * BLOCK * CELL * L * 2 / 2**30, 1), "startup_s": round(startup, 3),
           "tlb_shootdowns": st["tlb_shootdowns"],
           "page_ids_per_map_call": round(st["pages_mapped"] / (2 * L) / max(st["map_calls"], 1), 2),
           "lanes_per_buffer": int(capi.get_option(129)),
           "host_us_per_2MiB_mapped": {name: round(int(capi.get_option(130 + i)) / 1e3 / mapped, 3) for i, name in SEGMENTS.items()
                                       if int(capi.get_option(130 + i))},
           "buffers_created": c1[3] - c0[3],
           "create_us_per_buffer": {"kfd_alloc": round((c1[0] - c0[0]) / n_created / 1e3, 1), "export": round((c1[1] - c0[1]) / n_created / 1e3, 1),
                                    "drm_import": round((c1[2] - c0[2]) / n_created / 1e3, 1)} if c1[3] > c0[3] else None}
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rates", default="4,16,64")
    ap.add_argument("--duration", type=float, default=120.0)
    ap.add_argument("--no-parity", action="store_true")
    args = ap.parse_args()
    assert torch.cuda.is_available(), "needs the MI355X"
    if not args.no_parity:
        print(json.dumps(parity_pass()), flush=True)
    for r in [float(x) for x in args.rates.split(",") if x]:
        for out in elastic_pass(r, args.duration):
            print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
