#!/usr/bin/env python3
"""Slow-path alloc() latency on one MI355X: an alloc that needs k new page ids (Llama-3-8B geometry: one page id =
64 slots of 2 MiB), page by page like the reference (one map call + one TLB invalidation per page id) vs batched
(KVCACHED_BATCH_PAGE_ALLOC, the default here: one map call for all k). The reserved-page pool is switched off so
that every alloc really takes the slow path. One JSON line per k and mode."""
import json
import os
import statistics
import sys
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
os.environ.setdefault("KVCACHED_LOG_LEVEL", "ERROR")
os.environ.setdefault("KVCACHED_IPC_NAME", f"kvc_probe_batch_{os.getpid()}")
os.environ["KVCACHED_PAGE_PREALLOC_ENABLED"] = "false"
os.environ["KVCACHED_MIN_RESERVED_PAGES"] = "0"
os.environ["KVCACHED_MAX_RESERVED_PAGES"] = "0"


def main():
    import torch
    import kvcached_amd.kv_cache_manager as kcm
    from kvcached_amd import vmm_ops
    PAGE, L = 2 << 20, 32
    kcm.CONTIGUOUS_LAYOUT = False
    vmm_ops.init_kvcached("cuda:0", PAGE, False)
    nblocks = 256 * 64
    vmm_ops.create_kv_tensors(nblocks * 16 * 2048 * 2, 1, "cuda:0", L, 2, 0, False)
    try:
        for batch in (False, True):
            kcm.BATCH_PAGE_ALLOC = batch
            m = kcm.KVCacheManager(num_blocks=nblocks, block_size=16, cell_size=2048, num_layers=L)
            assert m._post_init_done.wait(10)
            for k in (1, 2, 4, 8, 16):
                ta, tf = [], []
                for it in range(12):
                    t0 = time.perf_counter()
                    ids = m.alloc(k * 64)
                    t1 = time.perf_counter()
                    m.free(ids)
                    t2 = time.perf_counter()
                    if it >= 2:
                        ta.append(t1 - t0)
                        tf.append(t2 - t1)
                print(json.dumps({"batched_page_alloc": batch, "new_page_ids": k, "slots_2MiB": k * 64,
                                  "alloc_ms_p50": round(statistics.median(ta) * 1e3, 3),
                                  "free_ms_p50": round(statistics.median(tf) * 1e3, 3)}), flush=True)
            del m
    finally:
        vmm_ops.shutdown_kvcached()


if __name__ == "__main__":
    main()
