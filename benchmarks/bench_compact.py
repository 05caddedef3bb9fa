#!/usr/bin/env python3
"""compact_blocks micro-benchmark on one MI355X: kernel variants x geometries, timed with HIP events inside the
library. `--profile-shape` runs ONE geometry/variant a few times and nothing else, for rocprofv3 passes
(kernel trace, and FETCH_SIZE / WRITE_SIZE each in its own --pmc pass) whose per-launch HBM traffic is compared
with the algorithmic bytes: read = written = block_bytes x regions x moves per launch."""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
KiB = 1 << 10
VARIANTS = ["default (= lds+nt, 32 KiB tiles, an eighth of the pairs per XCD)", "reg+xcd+nt", "lds", "reg", "lds+xcd", "reg+xcd", "lds+xcd+nt, 32 KiB tiles", "reg+xcd+nt, 32 KiB tiles", "lds+xcd+nt, 16 KiB tiles (default until r03)", "lds+nt, 32 KiB tiles, an eighth of the pairs per XCD", "reg+nt, 32 KiB tiles, an eighth of the pairs per XCD", "lds+nt, 32 KiB tiles, an eighth of the pairs per XCD, XCDs out of step"]
GEOMETRIES = ((2048 * KiB, 1, 2048, 1024),    # the ceiling: ONE region, 2 MiB blocks i -> i + 1024, i.e. a contiguous 2 GiB -> 2 GiB copy through the same kernel
              (32 * KiB, 64, 4096, 2048),     # Llama-3-8B: 32 layers x K/V regions, 32 KiB blocks
              (32 * KiB, 64, 4096, 256),
              (32 * KiB, 64, 16384, 0),       # the same geometry, moves as KVCacheManager.plan_compaction makes them on 30 %-occupied pages (256 pages per region)
              (16 * KiB, 32, 8192, 2048),     # cfg 1 of the reference's tests: 16 layers, 16 KiB blocks
              (18432, 54, 4096, 1024))        # MLA-like: 16 tokens x 1152 B, 27 layers x 2


def planned_moves(n_blocks, per_page, occupancy=0.3, seed=2):
    """What KVCacheManager.plan_compaction produces on pages that are `occupancy` full at random ('random
    30 %-occupied pages, seed 2'): the live blocks of the sparsest pages, ascending, go into the free blocks of the fullest
    pages, ascending - whole donor pages only, as long as the receivers can absorb them."""
    import numpy as np
    rng = np.random.default_rng(seed)
    live = rng.random(n_blocks) < occupancy
    pages = [(p, [b for b in range(p * per_page, (p + 1) * per_page) if live[b]], [b for b in range(p * per_page, (p + 1) * per_page) if not live[b]])
             for p in range(n_blocks // per_page)]
    donors = sorted(pages, key=lambda t: (len(t[1]), t[0]))
    receivers = sorted(pages, key=lambda t: (-len(t[1]), t[0]))
    moves, gone, taken, ri = [], set(), {}, 0
    for pid, used, _ in donors:
        if pid in taken:
            continue                                   # (already promised blocks as a receiver)
        plan, need = [], len(used)
        for rpid, _, rfree in receivers:
            if need == 0:
                break
            if rpid == pid or rpid in gone:
                continue
            room = len(rfree) - taken.get(rpid, 0)
            k = min(room, need)
            if k > 0:
                plan.append((rpid, rfree, k))
                need -= k
        if need:
            break
        it = iter(used)
        for rpid, rfree, k in plan:
            base = taken.get(rpid, 0)
            moves += [(next(it), d) for d in rfree[base:base + k]]
            taken[rpid] = base + k
        gone.add(pid)
    return [s for s, _ in moves], [d for _, d in moves]


def setup(block, regions, n_blocks, moves):
    import numpy as np
    import torch
    bufs = [torch.randint(0, 127, (n_blocks * block,), dtype=torch.int8, device="cuda:0") for _ in range(regions)]
    ids = np.random.default_rng(0).permutation(n_blocks)[:2 * moves]
    src, dst = [int(x) for x in ids[:moves]], [int(x) for x in ids[moves:]]
    if regions == 1:                             # the contiguous case
        src, dst = list(range(moves)), list(range(moves, 2 * moves))
    if moves == 0:                               # the planner's moves on 30 %-occupied pages
        src, dst = planned_moves(n_blocks, (2 << 20) // block)
    torch.cuda.synchronize()
    return bufs, [b.data_ptr() for b in bufs], src, dst


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--profile-shape", action="store_true", help="Llama-3-8B geometry, default variant, 6 calls, no sweep")
    ap.add_argument("--kv-regions", action="store_true",
                    help="the Llama-3-8B geometry on REAL KV regions: 64 page ids mapped through the library (64 regions of 2 MiB slots, "
                         "each slot its own page-table entry) instead of torch buffers")
    args = ap.parse_args()
    os.environ.setdefault("KVCACHED_LOG_LEVEL", "ERROR")
    import torch
    if not torch.cuda.is_available():
        raise SystemExit("needs a GPU")
    from kvcached_amd import capi
    capi.init("cuda:0", 2 << 20, False)
    try:
        if args.kv_regions:
            import numpy as np
            page, block, n_blocks = 2 << 20, 32 * KiB, 4096
            capi.create_kv_tensors(2 * 64 * page, 1, "cuda:0", 32, 2, 0, False)       # 32 layers x K/V, 64 page ids each
            capi.map_to_kv_tensors([p * page for p in range(64)])
            torch.cuda.synchronize()
            bases = capi.get_region_bases(0)
            ids = np.random.default_rng(0).permutation(n_blocks)[:4096]
            cases = {"random pairing, 2048 moves": ([int(x) for x in ids[:2048]], [int(x) for x in ids[2048:]]),
                     "planner-ordered moves on 30 %-occupied pages": planned_moves(n_blocks, page // block)}
            for label, (src, dst) in cases.items():
                for variant, vname in ((0, VARIANTS[0]), (6, VARIANTS[6]), (10, VARIANTS[10]), (11, VARIANTS[11]), (0, VARIANTS[0]), (11, VARIANTS[11])):
                    capi.set_option(capi.OPT_COMPACT_VARIANT, variant)
                    for _ in range(2):
                        capi.compact_blocks(bases, src, dst, block)
                    capi.set_option(capi.OPT_PROFILE, 1)
                    capi.reset_stats()
                    for _ in range(5):
                        capi.compact_blocks(bases, src, dst, block, sync=False)
                    capi.compact_blocks(bases[:1], src[:1], dst[:1], block, sync=True)
                    st = capi.get_stats()
                    capi.set_option(capi.OPT_PROFILE, 0)
                    print(json.dumps(dict(where="KV regions mapped by the library (2 MiB slots)", regions=len(bases), block=block, moves=len(src), what=label,
                                          variant=vname, event_GBps=round(st["compact_bytes"] / st["compact_ms"] / 1e6))), flush=True)
            capi.set_option(capi.OPT_COMPACT_VARIANT, 0)
            capi.unmap_from_kv_tensors([p * page for p in range(64)])
            return
        if args.profile_shape:
            block, regions, n_blocks, moves = GEOMETRIES[1]
            bufs, bases, src, dst = setup(block, regions, n_blocks, moves)
            for _ in range(6):
                capi.compact_blocks(bases, src, dst, block)
            print(json.dumps({"block": block, "regions": regions, "moves": moves,
                              "algorithmic_read_bytes_per_call": block * regions * moves,
                              "algorithmic_write_bytes_per_call": block * regions * moves}))
            return
        for block, regions, n_blocks, moves in GEOMETRIES:
            bufs, bases, src, dst = setup(block, regions, n_blocks, moves)
            for variant in range(len(VARIANTS)):
                capi.set_option(capi.OPT_COMPACT_VARIANT, variant)
                for _ in range(2):
                    capi.compact_blocks(bases, src, dst, block)
                capi.set_option(capi.OPT_PROFILE, 1)
                capi.reset_stats()
                t0 = time.perf_counter()
                for _ in range(5):
                    capi.compact_blocks(bases, src, dst, block, sync=False)
                capi.compact_blocks(bases[:1], src[:1], dst[:1], block, sync=True)
                wall = time.perf_counter() - t0
                st = capi.get_stats()
                capi.set_option(capi.OPT_PROFILE, 0)
                print(json.dumps(dict(block=block, regions=regions, moves=len(src), planned=moves == 0, variant=VARIANTS[variant],
                                      launches=st["compact_launches"], event_GBps=round(st["compact_bytes"] / st["compact_ms"] / 1e6),
                                      wall_GBps=round(st["compact_bytes"] / wall / 1e9))), flush=True)
            capi.set_option(capi.OPT_COMPACT_VARIANT, 0)
            del bufs
    finally:
        capi.shutdown()


if __name__ == "__main__":
    main()
