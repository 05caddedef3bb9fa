"""Soak of the whole hot path with the data checked: random alloc / free / trim / resize against a live
KVCacheManager on one MI355X, every block signed in every layer's K and V when it is handed out and verified just
before it is given back. A stale translation, a page mapped under the wrong slot or a handle recycled while still
visible shows up as a wrong signature; a leak shows up in the handle ledger at the end.

    python benchmarks/soak_manager.py --seconds 60 [--backend drm|hybrid|hip] [--async-unmap] [--compat] [--prealloc]

Prints one JSON line. Exit code 1 on any mismatch."""
import argparse
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))

PAGE = 2 << 20
LAYERS, BLOCK_TOKENS, CELL = 4, 16, 2048                       # 32 KiB blocks, 64 per page; a page id = 8 slots
BLOCK_BYTES = BLOCK_TOKENS * CELL


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=60)
    ap.add_argument("--backend", default=None)
    ap.add_argument("--async-unmap", action="store_true")
    ap.add_argument("--compat", action="store_true", help="KVCACHED_ZERO_BACKFILL=true (the reference's aliasing)")
    ap.add_argument("--prealloc", action="store_true", help="prealloc + watcher threads on")
    ap.add_argument("--page-ids", type=int, default=1024, help="virtual pool size in page ids (16 MiB each)")
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--pool-mb", type=int, default=None, help="KVCACHED_PHYS_POOL_MB (0: every whole extent goes straight back "
                    "to the driver, so that held/mapped shows fragmentation alone)")
    ap.add_argument("--extent-pages", type=int, default=None, help="KVCACHED_PHYS_CHUNK_PAGES")
    ap.add_argument("--touch-unbacked", action="store_true",
                    help="every 32 ops a kernel reads one word of EVERY slot of every region, backed or not (needs a mode in which "
                         "unbacked VA does not fault: PRT or zero aliases): if a PRT 'miss' were ever cached by the GPU, a slot "
                         "backed afterwards would be shadowed by it and its signatures would come back wrong")
    ap.add_argument("--unmap-invalidation-us", type=int, default=None,
                    help="KVCACHED_UNMAP_INVALIDATION_US: compat mode's relaxed form - the invalidation an unmap owes trails the call by at "
                         "most this long, the pages wait for it un-scrubbed and un-offered")
    ap.add_argument("--no-prt", action="store_true", help="compat mode with the zero extent behind unbacked VA instead of PRT (KVCACHED_PRT=false)")
    args = ap.parse_args()
    if args.no_prt:
        os.environ["KVCACHED_PRT"] = "false"
    if args.unmap_invalidation_us is not None:
        os.environ["KVCACHED_UNMAP_INVALIDATION_US"] = str(args.unmap_invalidation_us)
    if args.pool_mb is not None:
        os.environ["KVCACHED_PHYS_POOL_MB"] = str(args.pool_mb)
    if args.extent_pages is not None:
        os.environ["KVCACHED_PHYS_CHUNK_PAGES"] = str(args.extent_pages)
    if args.touch_unbacked and not args.compat:
        os.environ["KVCACHED_PRT"] = "true"        # lazy mode leaves unbacked VA unmapped by default: touching it would fault
    # A slot IN TRANSITION is not at rest: inside the one ioctl that replaces PRT by a page (or back) the kernel first clears
    # the range and then writes the new entries, and a GPU access that lands in that window faults (found the hard way:
    # profiles/r02_soak_touch_unbacked_fault.log; there is no sequence of DRM operations without the window:
    # tools/engine_ioctl_probe.cpp part H). Nothing legitimate touches a slot that is being backed or given up - but these
    # sweeps would, as soon as another thread of the library maps or unmaps in the background (the prealloc thread, the
    # reclaimer). So they run under capi.quiesced() - kvc_quiesce_begin/_end, the library's own fence for code that reads
    # whole KV tensors - which holds every page-table update for as long as a sweep's kernels run.
    if args.backend:
        os.environ["KVCACHED_VMM_BACKEND"] = args.backend
    os.environ["KVCACHED_ASYNC_UNMAP"] = "true" if args.async_unmap else "false"
    os.environ["KVCACHED_ZERO_BACKFILL"] = "true" if args.compat else "false"
    os.environ["KVCACHED_PAGE_PREALLOC_ENABLED"] = "true" if args.prealloc else "false"
    os.environ.setdefault("KVCACHED_IPC_NAME", f"kvc_soak_{os.getpid()}")
    os.environ.setdefault("KVCACHED_LOG_LEVEL", "ERROR")
    import torch

    import kvcached_amd.kv_cache_manager as kcm
    from kvcached_amd import capi, vmm_ops
    from kvcached_amd.cli.utils import update_kv_cache_limit

    dev = "cuda:0"
    torch.cuda.set_device(0)
    vmm_ops.init_kvcached(dev, PAGE, False)
    backend = {0: "hip", 2: "hybrid", 3: "drm"}[capi.get_option(108)]
    max_extent_pages = int(capi.get_option(119)) if backend == "drm" and capi.get_option(110) else 1
    capi_prt = 0
    blocks_per_page = PAGE // BLOCK_BYTES
    num_blocks = args.page_ids * blocks_per_page
    per_layer = num_blocks * BLOCK_BYTES * 2                   # K half + V half
    ts = vmm_ops.create_kv_tensors(per_layer, 1, dev, LAYERS, 2, 0, False)
    words = [t.view(torch.int64) for t in ts]                  # signatures are int64 words
    v_off = per_layer // 2 // 8
    m = kcm.KVCacheManager(num_blocks=num_blocks, block_size=BLOCK_TOKENS, cell_size=CELL, num_layers=LAYERS)
    assert m._post_init_done.wait(30)
    capi_prt = capi.get_option(128)
    lanes = capi.get_option(129)
    ipc = m.page_allocator._ipc_name()
    full_limit = per_layer * LAYERS
    rng = np.random.default_rng(args.seed)
    live = {}                                                   # request id -> (block ids tensor on the GPU, n)
    next_rid = 1
    counts = dict(alloc=0, alloc_refused=0, free=0, trim=0, resize=0, blocks_signed=0, blocks_verified=0)
    bad = 0
    # each block carries its signature in its first AND last word (a block may straddle nothing, but a 2 MiB page
    # holds 64 of them: both ends of every page get covered)
    last = BLOCK_BYTES // 8 - 1

    def sign(ids_t, rid):
        sig = (rid << 24) + ids_t
        base = ids_t * (BLOCK_BYTES // 8)
        for layer, w in enumerate(words):
            val = sig + (layer << 56)
            w[base] = val
            w[base + last] = val
            w[v_off + base] = ~val
            w[v_off + base + last] = ~val

    def verify(ids_t, rid):
        sig = (rid << 24) + ids_t
        base = ids_t * (BLOCK_BYTES // 8)
        wrong = torch.zeros((), dtype=torch.int64, device=dev)
        for layer, w in enumerate(words):
            val = sig + (layer << 56)
            wrong += (w[base] != val).sum() + (w[base + last] != val).sum()
            wrong += (w[v_off + base] != ~val).sum() + (w[v_off + base + last] != ~val).sum()
        return int(wrong)

    def free_one(rid):
        nonlocal bad
        ids_t, ids = live.pop(rid)
        w = verify(ids_t, rid)
        if w:
            bad += w
            print(f"[soak] request {rid}: {w} signature words wrong over {len(ids)} blocks", file=sys.stderr)
            # what is there instead? (decoded: layer, request, block; ~x = a V-half signature; 0 = zero fill)
            sig = (rid << 24) + ids_t
            base = ids_t * (BLOCK_BYTES // 8)
            shown = 0
            for layer, wd in enumerate(words):
                val = sig + (layer << 56)
                for off, want, tag in ((0, val, "K.first"), (last, val, "K.last"), (v_off, ~val, "V.first"), (v_off + last, ~val, "V.last")):
                    got = wd[base + off]
                    idx = torch.nonzero(got != want).flatten()[:3].tolist()
                    for i in idx:
                        g, wv = int(got[i]), int(want[i])
                        dec = lambda x: (x >> 56 & 0xff, (x >> 24) & 0xffffffff, x & 0xffffff)
                        print(f"    layer {layer} {tag} block {int(ids_t[i])} (page {int(ids_t[i]) // 64}): got {g:#x} {dec(g)} / ~{dec(~g)}  want {wv:#x}", file=sys.stderr)
                        shown += 1
                if shown > 12:
                    break
        counts["blocks_verified"] += len(ids)
        m.free(ids)
        counts["free"] += 1

    t_end = time.time() + args.seconds
    shrunk = False
    slots_per_page_id = LAYERS * 2
    footprint = []                                              # (pages held from the driver) / (pages mapped), sampled
    n_ops = 0
    touched_sum = 0
    if args.touch_unbacked:
        assert capi.get_option(128) or args.compat, "unbacked VA would fault"
    t_progress = time.time()
    while time.time() < t_end and not bad:
        n_ops += 1
        if time.time() - t_progress > 30:                       # a line every 30 s: long runs must not look hung
            t_progress = time.time()
            print(f"[soak] {int(t_end - time.time())} s to go, {counts['alloc']} allocs, {counts['blocks_verified']} blocks verified, "
                  f"{bad} wrong words", file=sys.stderr, flush=True)
        if args.touch_unbacked and n_ops % 32 == 0:
            # every slot of every region, backed or not: alternately one word per slot (a one-workgroup kernel) and one word
            # per 4 KiB (chip-wide: every XCD's TLBs get to see the unbacked neighbours of whatever is backed - the shape that
            # tools/prt_tlb_probe.cpp needed to show what cached PRT entries do)
            stride = PAGE // 8 if (n_ops // 32) % 2 else 512
            with capi.quiesced():                               # (no slot is in transition while the kernels run)
                for w in words:
                    touched_sum += int(w[::stride].sum())
            counts["sweeps_over_every_slot"] = counts.get("sweeps_over_every_slot", 0) + 1
        if n_ops % 64 == 0:
            st_now = capi.get_stats()
            held_pages = st_now["handles_created"] - st_now["handles_released"]
            mapped_pages = (m.page_allocator.get_num_inuse_pages() + m.page_allocator.get_num_reserved_pages()) * slots_per_page_id
            if mapped_pages >= 256:
                footprint.append(held_pages / mapped_pages)
        r = rng.random()
        held = sum(len(v[1]) for v in live.values())
        if r < 0.50 or not live:
            n = int(rng.integers(1, 4000)) if rng.random() < 0.8 else int(rng.integers(1, 40))
            ids = m.alloc(n)
            if ids is None:
                counts["alloc_refused"] += 1
                if live:
                    free_one(next(iter(live)))
                continue
            ids_t = torch.tensor(ids, dtype=torch.int64, device=dev)
            if args.touch_unbacked and next_rid % 8 == 0:
                # the window that only a map opens (until the next unmap the kernel has the remainders of the PRT mappings this
                # alloc has split still queued): the unbacked neighbours are looked at chip-wide BEFORE the new
                # blocks are written, and the blocks are read back at once
                with capi.quiesced():
                    for w in words:
                        touched_sum += int(w[::512].sum())
                sign(ids_t, next_rid)
                with capi.quiesced():
                    for w in words:
                        touched_sum += int(w[::512].sum())
                wnow = verify(ids_t, next_rid)
                if wnow:
                    bad += wnow
                    print(f"[soak] request {next_rid}: {wnow} signature words wrong right after its alloc, over {len(ids)} blocks", file=sys.stderr)
            else:
                sign(ids_t, next_rid)
            live[next_rid] = (ids_t, ids)
            next_rid += 1
            counts["alloc"] += 1
            counts["blocks_signed"] += n
        elif r < 0.93:
            keys = list(live)
            free_one(keys[int(rng.integers(len(keys)))])
        elif r < 0.96:
            m.trim()
            counts["trim"] += 1
        elif args.prealloc:                                    # the watcher applies limits written to the shm record
            shrunk = not shrunk
            from contextlib import redirect_stdout
            with open(os.devnull, "w") as devnull, redirect_stdout(devnull):
                update_kv_cache_limit(ipc, full_limit // 2 if shrunk else full_limit)
            counts["resize"] += 1
        else:
            m.available_size()
        if held > num_blocks * 0.6:                            # keep the pool from saturating: drop a few requests
            for rid in list(live)[:3]:
                free_one(rid)
    for rid in list(live):
        free_one(rid)
    torch.cuda.synchronize()
    m.trim()
    capi.flush_unmaps()
    inuse = m.page_allocator.get_num_inuse_pages()
    st = capi.get_stats()
    del m
    vmm_ops.shutdown_kvcached()
    end = capi.get_stats()
    leak = end["handles_created"] - end["handles_released"]
    fp = sorted(footprint) or [0.0]
    out = dict(held_over_mapped={"p50": round(fp[len(fp) // 2], 3), "p90": round(fp[int(len(fp) * 0.9)], 3), "max": round(fp[-1], 3),
                                 "what": "physical pages held from the driver (mapped + pooled + free pieces of partly used chunks) per mapped page"},
               max_extent_pages=max_extent_pages, prt=bool(capi_prt), pool_mb=os.environ.get("KVCACHED_PHYS_POOL_MB", "default (16384)"),
               backend=backend, seconds=args.seconds, async_unmap=args.async_unmap, compat=args.compat,
               unmap_invalidation_us=int(os.environ.get("KVCACHED_UNMAP_INVALIDATION_US", "0")), lanes_per_buffer=int(lanes),
               tlb_shootdowns=st["tlb_shootdowns"],
               prealloc=args.prealloc, **counts, wrong_words=bad, inuse_pages_at_end=inuse,
               pages_mapped=st["pages_mapped"], pages_unmapped=st["pages_unmapped"],
               handles_created=end["handles_created"], handles_reused=end["handles_reused"], handle_leak=leak)
    print(json.dumps(out))
    return 1 if (bad or leak or inuse) else 0


if __name__ == "__main__":
    sys.exit(main())
