#!/usr/bin/env python3
"""KVCacheManager alloc/free micro-benchmark on one MI355X - the protocol of the reference's benchmarks/bench_alloc
(README.md:22-65; its numbers, on a GB10: BASELINE.md §2 rows "Slow-path alloc(k)", "available_size()",
"group_indices_by_page", "alloc(16)+free throughput"), same geometry (16 layers, block 16 tokens, 65536 blocks,
cell 1024 B = bench_alloc.py:16-22 there), through the same public API (`integration.vllm.interfaces`, `KVCacheManager`):

  1. available_size()                          us per call
  2. group_indices_by_page, N = 64/1024/16384  us per call
  3. fast path: alloc(k)+free, k = 1..256      us per pair (reserved pages absorb it: no driver call)
  4. slow path: KVCACHED_MIN/MAX_RESERVED_PAGES=0, every alloc backs fresh page ids and every free gives them back:
     alloc(k)+free for k = 128 / 1024 / 4096 blocks (1 / 8 / 32 page ids of 32 slots)
  5. alloc(16)+free from 1 / 4 / 8 Python threads, aggregate Kops/s

Sections 1-3 and 5 and section 4 run in two child processes (the reserved-page knobs are read when the library loads).

    python benchmarks/bench_alloc.py [--section fast|slow]      # no argument: both, as children
"""
from __future__ import annotations

import argparse
import json
import os
import subprocess
import sys
import threading
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
NUM_LAYERS, BLOCK_SIZE, NUM_BLOCKS, CELL = 16, 16, 65536, 1024
DEVICE = "cuda:0"


def setup():
    import torch
    from kvcached_amd.integration.vllm.interfaces import alloc_kv_cache, init_kvcached
    from kvcached_amd.kv_cache_manager import KVCacheManager
    from kvcached_amd.vmm_ops import kv_tensors_created
    torch.cuda.set_device(0)
    init_kvcached(tp_rank=0, world_size=1, is_worker=True, async_sched=False)
    alloc_kv_cache(kvcache_shape=(2, NUM_BLOCKS, BLOCK_SIZE, 8, 64), block_size=BLOCK_SIZE, dtype=torch.float16, device=DEVICE,
                   num_layers=NUM_LAYERS)
    t0 = time.time()
    while not kv_tensors_created():
        assert time.time() - t0 < 20, "KV tensors not created within 20 s"
        time.sleep(0.05)
    m = KVCacheManager(num_blocks=NUM_BLOCKS, block_size=BLOCK_SIZE, cell_size=CELL, num_layers=NUM_LAYERS, world_size=1)
    assert m._post_init_done.wait(30)
    return m


def pair_us(m, k, iters, warm):
    for _ in range(warm):
        m.free(m.alloc(k))
    t0 = time.perf_counter()
    for _ in range(iters):
        m.free(m.alloc(k))
    return (time.perf_counter() - t0) / iters * 1e6


def per_call_us(fn, iters):
    fn()
    t0 = time.perf_counter()
    for _ in range(iters):
        fn()
    return (time.perf_counter() - t0) / iters * 1e6


def section_fast():
    import numpy as np
    m = setup()
    out = {"section": "fast path (defaults: 5-10 page ids kept reserved, prealloc thread on)"}
    out["available_size_us"] = round(per_call_us(m.available_size, 50000), 3)
    rng = np.random.default_rng(0)
    for n in (64, 1024, 16384):
        idx = [int(x) for x in rng.choice(NUM_BLOCKS, n, replace=False)]
        out[f"group_indices_by_page_N{n}_us"] = round(per_call_us(lambda: m.page_allocator.group_indices_by_page(idx, m.block_mem_size), 2000 if n < 16384 else 300), 2)
    for k, iters in ((1, 50000), (4, 50000), (16, 50000), (64, 20000), (256, 10000)):
        out[f"alloc({k})+free_us"] = round(pair_us(m, k, iters, 100), 2)
    for threads in (1, 4, 8):
        per = 20000 // threads
        def worker():
            for _ in range(per):
                h = m.alloc(16)
                if h is not None:
                    m.free(h)
        for _ in range(200):
            m.free(m.alloc(16))
        ts = [threading.Thread(target=worker) for _ in range(threads)]
        t0 = time.perf_counter()
        for t in ts:
            t.start()
        for t in ts:
            t.join()
        out[f"alloc(16)+free_{threads}_threads_Kops"] = round(per * threads / (time.perf_counter() - t0) / 1e3, 1)
    print(json.dumps(out), flush=True)
    finish(m)


def section_slow():
    from kvcached_amd import capi
    m = setup()
    per_page = (2 << 20) // (BLOCK_SIZE * CELL)          # 128 blocks per page id; a page id = 16 layers x K/V = 32 slots of 2 MiB
    out = {"section": "slow path (KVCACHED_MIN/MAX_RESERVED_PAGES=0: every alloc backs fresh page ids, every free gives them back)",
           "reserved_pages": m.page_allocator.get_num_reserved_pages(), "slots_2MiB_per_page_id": NUM_LAYERS * 2}
    for k, iters in ((128, 300), (1024, 100), (4096, 40)):
        st0 = capi.get_stats()
        us = pair_us(m, k, iters, 5)
        st1 = capi.get_stats()
        pages = (st1["pages_mapped"] - st0["pages_mapped"]) / (iters + 5)
        out[f"alloc({k})+free_us"] = round(us, 1)
        out[f"alloc({k})+free_slots_mapped_per_pair"] = round(pages, 1)
        out[f"alloc({k})_us_per_2MiB_slot_(map+unmap)"] = round(us / max(pages, 1), 2)
        assert pages >= k // per_page * NUM_LAYERS * 2 * 0.99, (k, pages)          # it really is the slow path
    print(json.dumps(out), flush=True)
    finish(m)


def finish(m):
    from kvcached_amd.integration.vllm.interfaces import shutdown_kvcached
    del m
    shutdown_kvcached()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--section", choices=["fast", "slow"])
    args = ap.parse_args()
    os.environ.setdefault("KVCACHED_LOG_LEVEL", "ERROR")
    if args.section == "fast":
        return section_fast()
    if args.section == "slow":
        return section_slow()
    for sec, env in (("fast", {}), ("slow", {"KVCACHED_MIN_RESERVED_PAGES": "0", "KVCACHED_MAX_RESERVED_PAGES": "0"})):
        e = dict(os.environ, KVCACHED_IPC_NAME=f"kvc_bench_alloc_{os.getpid()}_{sec}", **env)
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "--section", sec], env=e, capture_output=True, text=True, timeout=900)
        lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
        print(lines[-1] if lines and r.returncode == 0 else json.dumps({"section": sec, "error": (r.stderr or r.stdout)[-400:]}), flush=True)


if __name__ == "__main__":
    main()
